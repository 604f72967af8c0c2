// TEST INFRASTRUCTURE ONLY — a tiny SIMT emulator so that the device code of typesense_amd/csrc/*.hip.h can
// be executed on the build container's CPU (which has no GPU) by tests/test_emu_*.py.
//
// This directory shadows <hip/hip_runtime.h> for ONE translation unit (tests/hipemu/emu_harness.cpp), compiled
// with the host compiler. The kernel sources are included unmodified; nothing here is linked into libtsgpu.so
// and the product never falls back to it. It models exactly what the kernels use:
//   * a workgroup = N cooperative fibers (ucontext), run one block at a time;
//   * __syncthreads(): all live fibers of the block; wave collectives (__ballot, __shfl*, MFMA): all live
//     lanes of the 64-wide wave — a collective reached by only part of a wave is reported as a divergence
//     error instead of silently "working";
//   * every textual __syncthreads() carries an id: fibers of a block that meet at DIFFERENT barriers (a race on the way into barrier-carrying
//     code, which the cooperative schedule would otherwise hide) are reported, fatally under HIPEMU_STRICT_BARRIERS=1 (the test tier sets it);
//   * runnable fibers take turns in ascending order, or — HIPEMU_SCHEDULE=reverse | shuffle — in another one: results must not depend on it;
//   * __shared__ = function-local static (one block at a time); atomics = plain ops (cooperative scheduling);
//   * __builtin_amdgcn_mfma_f32_32x32x2f32 with the CDNA4 fragment layout and k-ordered fmaf chain
//     (cdna_hip_programming.md §3: bit-exact model of v_mfma_f32_32x32x2_f32).
#pragma once
#include <ucontext.h>
#include <time.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>
#include <mutex>

#define TSGPU_HIP_EMU 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __forceinline__ inline

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace hipemu {

enum WaitKind { RUNNING = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;          // from the stack pool below: fiber stacks are reused across blocks and launches (a fresh zero-filled
                                    // 256 KB vector per fiber per block was most of the emulator's run time: 128 MB of page faults per 512-thread block)
    dim3 tid;
    int wait = RUNNING;
    unsigned wave_gen = 0;
    int barrier_site = 0;           // which textual __syncthreads() the fiber waits at (0 = an internal barrier of a wrapper)
};

struct WaveState {
    unsigned gen = 0;
    uint64_t u64[64];
    float f32a[64], f32b[64];
    uint16_t bfa[64][8], bfb[64][8];
    int8_t i8a[64][16], i8b[64][16];
};

struct BlockState {
    int or_flag = 0;
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    ucontext_t sched;
    int cur = -1;
    dim3 block_idx, block_dim, grid_dim;
    std::function<void()> body;
};

inline BlockState*& g() { static BlockState* b = nullptr; return b; }

inline void yield_to_sched() { BlockState* b = g(); swapcontext(&b->fibers[b->cur].ctx, &b->sched); }

inline void syncthreads(int site = 0) {
    BlockState* b = g();
    b->fibers[b->cur].wait = AT_BARRIER;
    b->fibers[b->cur].barrier_site = site;
    yield_to_sched();
}

// all live lanes of the calling lane's wave rendezvous here
inline void wave_sync() {
    BlockState* b = g();
    b->fibers[b->cur].wait = AT_WAVE;
    yield_to_sched();
}

inline void fiber_entry() {
    BlockState* b = g();
    b->body();
    b->fibers[b->cur].wait = DONE;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

static const size_t FIBER_STACK = 256 * 1024;
inline char* pooled_stack(unsigned i) {           // (launches are serialised by launch_mutex: one pool)
    static std::vector<char*> pool;
    while (pool.size() <= i) pool.push_back((char*)malloc(FIBER_STACK));
    return pool[i];
}
inline void run_block(BlockState& B) {
    g() = &B;
    const unsigned n = B.block_dim.x;
    B.fibers.resize(n);
    B.waves.assign((n + 63) / 64, WaveState());
    for (unsigned i = 0; i < n; i++) {
        Fiber& f = B.fibers[i];
        f.stack = pooled_stack(i);
        f.tid = dim3(i, 0, 0);
        f.wait = RUNNING;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = FIBER_STACK;
        f.ctx.uc_link = &B.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    // the order in which runnable fibers get their turn. The hardware promises none: a kernel whose RESULT depends on it has a race. Default:
    // ascending; HIPEMU_SCHEDULE=reverse | shuffle (per-block pseudo-random, reseeded every pass) to run the same tests under other orders.
    static const int sched_mode = [] { const char* e = getenv("HIPEMU_SCHEDULE"); return !e ? 0 : (!strcmp(e, "reverse") ? 1 : (!strcmp(e, "shuffle") ? 2 : 0)); }();
    std::vector<unsigned> order(n);
    for (unsigned i = 0; i < n; i++) order[i] = sched_mode == 1 ? n - 1 - i : i;
    uint64_t rng = 0x9E3779B97F4A7C15ull * (B.block_idx.x + 1) + n;
    for (;;) {
        bool progressed = false, any_live = false;
        if (sched_mode == 2) for (unsigned i = n; i > 1; i--) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i - 1], order[(rng >> 33) % i]); }
        for (unsigned oi = 0; oi < n; oi++) {
            const unsigned i = order[oi];
            Fiber& f = B.fibers[i];
            if (f.wait == DONE) continue;
            any_live = true;
            if (f.wait == RUNNING) { B.cur = (int)i; swapcontext(&B.sched, &f.ctx); progressed = true; }
        }
        if (!any_live) break;
        // release wave collectives: every live lane of a wave is AT_WAVE
        for (unsigned w = 0; w < B.waves.size(); w++) {
            bool all = true, any = false;
            for (unsigned l = w * 64; l < n && l < (w + 1) * 64; l++) {
                if (B.fibers[l].wait == DONE) continue;
                any = true;
                if (B.fibers[l].wait != AT_WAVE) all = false;
            }
            if (any && all) {
                for (unsigned l = w * 64; l < n && l < (w + 1) * 64; l++) if (B.fibers[l].wait == AT_WAVE) B.fibers[l].wait = RUNNING;
                progressed = true;
            }
        }
        // release the block barrier: every live fiber is AT_BARRIER
        bool all = true, any = false;
        for (unsigned i = 0; i < n; i++) {
            if (B.fibers[i].wait == DONE) continue;
            any = true;
            if (B.fibers[i].wait != AT_BARRIER) all = false;
        }
        if (any && all) {
            // every live fiber is at A barrier — on the hardware that is enough (s_barrier counts arrivals), but fibers that meet at DIFFERENT textual
            // barriers took different paths through barrier-carrying code (e.g. part of a workgroup entered a compaction the rest skipped): a race in
            // the kernel that a cooperative schedule would otherwise hide. Reported; fatal with HIPEMU_STRICT_BARRIERS=1.
            int site = -1; bool mixed = false;
            for (unsigned i = 0; i < n; i++) if (B.fibers[i].wait == AT_BARRIER) { if (site < 0) site = B.fibers[i].barrier_site; else if (B.fibers[i].barrier_site != site) mixed = true; }
            if (mixed) {
                static int reported = 0;
                static const bool strict = getenv("HIPEMU_STRICT_BARRIERS") != nullptr;
                if (reported++ < 8 || strict) {
                    fprintf(stderr, "hipemu: block %u: fibers met at DIFFERENT __syncthreads() call sites:", B.block_idx.x);
                    int last = -2;
                    for (unsigned i = 0; i < n; i++) if (B.fibers[i].wait == AT_BARRIER && B.fibers[i].barrier_site != last) { last = B.fibers[i].barrier_site; fprintf(stderr, " [t%u: site %d]", i, last); }
                    fprintf(stderr, "\n");
                }
                if (strict) abort();
            }
            for (unsigned i = 0; i < n; i++) if (B.fibers[i].wait == AT_BARRIER) B.fibers[i].wait = RUNNING;
            progressed = true;
        }
        if (!progressed) {
            fprintf(stderr, "hipemu: DEADLOCK / divergent collective in block %u: lane states:", B.block_idx.x);
            for (unsigned i = 0; i < n; i++) fprintf(stderr, " %d", B.fibers[i].wait);
            fprintf(stderr, "\n");
            abort();
        }
    }
    g() = nullptr;
}

// the emulator keeps ONE block's state in globals (g(), function-local statics standing in for __shared__): launches from
// concurrent host threads (two lanes, micro-batcher tests) run one after the other
inline std::recursive_mutex& launch_mutex() { static std::recursive_mutex m; return m; }
template <class F>
inline void launch(dim3 grid, dim3 block, F body) {
    std::lock_guard<std::recursive_mutex> lk(launch_mutex());
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        BlockState B;
        B.block_idx = dim3(bx, by, 0);
        B.block_dim = block;
        B.grid_dim = grid;
        B.body = body;
        run_block(B);
    }
}

inline dim3 cur_tid() { BlockState* b = g(); return b->fibers[b->cur].tid; }
inline dim3 cur_bid() { return g()->block_idx; }
inline dim3 cur_bdim() { return g()->block_dim; }
inline dim3 cur_gdim() { return g()->grid_dim; }
inline WaveState& cur_wave() { BlockState* b = g(); return b->waves[b->cur / 64]; }
inline unsigned cur_lane() { return (unsigned)g()->cur & 63u; }
inline bool lane_live(unsigned lane) { BlockState* b = g(); unsigned i = (unsigned)(b->cur / 64) * 64 + lane; return i < b->fibers.size() && b->fibers[i].wait != DONE; }

inline unsigned long long ballot(int pred) {
    WaveState& W = cur_wave();
    W.u64[cur_lane()] = pred ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; l++) if (lane_live(l) && W.u64[l]) m |= (1ull << l);
    wave_sync();
    return m;
}

template <class T>
inline T shfl(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl width");
    WaveState& W = cur_wave();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    W.u64[cur_lane()] = bits;
    wave_sync();
    T out = v;
    unsigned s = (unsigned)src_lane & 63u;
    if (lane_live(s)) memcpy(&out, &W.u64[s], sizeof(T));
    wave_sync();
    return out;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    WaveState& W = cur_wave();
    const unsigned l = cur_lane();
    W.f32a[l] = a;
    W.f32b[l] = b;
    wave_sync();
    f32x16 d = c;
    const unsigned col = l & 31;
    for (int r = 0; r < 16; r++) {
        const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (unsigned k = 0; k < 2; k++) acc = fmaf(W.f32a[row + 32 * k], W.f32b[col + 32 * k], acc);
        d[r] = acc;
    }
    wave_sync();
    return d;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r
inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    WaveState& W = cur_wave();
    const unsigned l = cur_lane();
    W.f32a[l] = a;
    W.f32b[l] = b;
    wave_sync();
    f32x4 d = c;
    const unsigned col = l & 15;
    for (int r = 0; r < 4; r++) {
        const unsigned row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (unsigned k = 0; k < 4; k++) acc = fmaf(W.f32a[row + 16 * k], W.f32b[col + 16 * k], acc);
        d[r] = acc;
    }
    wave_sync();
    return d;
}

// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)..+7], B[k=8*(l>>5)..+7][j=l&31]; D as the other 32x32 forms.
// bf16 x bf16 products are exact in fp32; the hardware's internal accumulation order is unspecified, the model
// adds them in k order (callers must not depend on the exact rounding of this instruction).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
inline f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    WaveState& W = cur_wave();
    const unsigned l = cur_lane();
    uint16_t ha[8], hb[8];
    memcpy(ha, &a, 16);
    memcpy(hb, &b, 16);
    for (int i = 0; i < 8; i++) { W.bfa[l][i] = ha[i]; W.bfb[l][i] = hb[i]; }
    wave_sync();
    f32x16 d = c;
    const unsigned col = l & 31;
    for (int r = 0; r < 16; r++) {
        const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (unsigned k = 0; k < 16; k++) {
            uint32_t ua = (uint32_t)W.bfa[row + 32 * (k >> 3)][k & 7] << 16, ub = (uint32_t)W.bfb[col + 32 * (k >> 3)][k & 7] << 16;
            float fa, fb;
            memcpy(&fa, &ua, 4);
            memcpy(&fb, &ub, 4);
            acc = fmaf(fa, fb, acc);
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}

// v_mfma_i32_32x32x32_i8: A[i=l&31][k=16*(l>>5)..+15], B[k=16*(l>>5)..+15][j=l&31] (16 int8 per lane = four dwords); exact int32 accumulation;
// D as the other 32x32 forms.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
inline i32x16 mfma_32x32x32_i8(i32x4 a, i32x4 b, i32x16 c) {
    WaveState& W = cur_wave();
    const unsigned l = cur_lane();
    memcpy(W.i8a[l], &a, 16);
    memcpy(W.i8b[l], &b, 16);
    wave_sync();
    i32x16 d = c;
    const unsigned col = l & 31;
    for (int r = 0; r < 16; r++) {
        const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int acc = c[r];
        for (unsigned k = 0; k < 32; k++) acc += (int)W.i8a[row + 32 * (k >> 4)][k & 15] * (int)W.i8b[col + 32 * (k >> 4)][k & 15];
        d[r] = acc;
    }
    wave_sync();
    return d;
}

}  // namespace hipemu

#define threadIdx (hipemu::cur_tid())
#define blockIdx (hipemu::cur_bid())
#define blockDim (hipemu::cur_bdim())
#define gridDim (hipemu::cur_gdim())

// every textual barrier gets an id (__COUNTER__): the scheduler checks that the fibers of a block meet at the SAME one
#define __syncthreads() hipemu::syncthreads(__COUNTER__ + 1)
#define __syncthreads_or(pred) hipemu::syncthreads_or_at((pred), __COUNTER__ + 1)
#define __syncthreads_count(pred) hipemu::syncthreads_count_at((pred), __COUNTER__ + 1)
namespace hipemu {
inline int syncthreads_or_at(int pred, int site) {      // barrier + OR-reduction over the block (two more barriers: read, then reset)
    BlockState* b = g();
    if (pred) b->or_flag = 1;
    syncthreads(site);
    const int r = b->or_flag;
    syncthreads(site);
    if (cur_tid().x == 0) b->or_flag = 0;
    syncthreads(site);
    return r;
}
inline int syncthreads_count_at(int pred, int site) {   // barrier + number of threads of the block whose pred is non-zero
    BlockState* b = g();
    if (pred) b->or_flag += 1;                // fibers of a block run one at a time: plain increment
    syncthreads(site);
    const int r = b->or_flag;
    syncthreads(site);
    if (cur_tid().x == 0) b->or_flag = 0;
    syncthreads(site);
    return r;
}
}  // namespace hipemu
inline unsigned long long __ballot(int pred) { return hipemu::ballot(pred); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
template <class T> inline T __shfl(T v, int lane, int = 64) { return hipemu::shfl(v, lane); }
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl(v, (int)(hipemu::cur_lane() ^ (unsigned)mask)); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) { unsigned s = hipemu::cur_lane() + d; return hipemu::shfl(v, s < 64 ? (int)s : (int)hipemu::cur_lane()); }
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) { unsigned l = hipemu::cur_lane(); return hipemu::shfl(v, l >= d ? (int)(l - d) : (int)l); }

template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
inline float4 make_float4(float x, float y, float z, float w) { float4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
// round-to-nearest single ops that must not be contracted into FMAs (emu TU is built with -ffp-contract=off)
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }

#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, x, y, z) hipemu::mfma_32x32x32_i8((a), (b), (c))
// global_load_lds_dwordx4: LDS destination = wave-uniform base + lane * 16, per-lane global source
inline void hipemu_global_load_lds4(const void* gsrc, void* lds_wave_base) { memcpy((char*)lds_wave_base + 4 * hipemu::cur_lane(), gsrc, 4); }
inline void hipemu_global_load_lds16(const void* gsrc, void* lds_wave_base) { memcpy((char*)lds_wave_base + 16 * hipemu::cur_lane(), gsrc, 16); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu::shfl(v, lane); }
// v_mov_b32_dpp with a quad_perm control (dpp_ctrl < 0x100): lane l reads lane (l & ~3) | perm[l & 3] of its quad; full row / bank masks
inline int __builtin_amdgcn_mov_dpp(int v, int dpp_ctrl, int row_mask, int bank_mask, bool) {
    if (dpp_ctrl < 0 || dpp_ctrl > 0xFF || row_mask != 0xF || bank_mask != 0xF) { fprintf(stderr, "hipemu: unsupported DPP control 0x%x\n", dpp_ctrl); abort(); }
    const unsigned l = hipemu::cur_lane();
    return hipemu::shfl(v, (int)((l & ~3u) | (((unsigned)dpp_ctrl >> (2 * (l & 3u))) & 3u)));
}
// device wall clock at 1 tick per microsecond (hipDeviceAttributeWallClockRate = 1000 kHz below)
inline long long hipemu_wall_clock64() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000; }
inline void __builtin_amdgcn_s_setprio(int) {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_sched_barrier(int) {}

// ---------------------------------------------------------------- host runtime API subset (emulated)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 1000; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 1; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(sizeof(hipemu_event)); (*e)->t = 0; return hipSuccess; }
static const unsigned hipEventBlockingSync = 1, hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
// fault injection (tests/test_emu_incremental.py): the N-th hipMalloc / synchronous hipMemcpy from now on fails once (0 = off).
// One counter per library image; set through the exported hipemu_fail_nth() below.
namespace hipemu { inline long& fail_countdown() { static long n = 0; return n; }
inline bool inject_fault() { long& n = fail_countdown(); if (n > 0 && --n == 0) return true; return false; } }
extern "C" inline __attribute__((visibility("default"), used)) void hipemu_fail_nth(long n) { hipemu::fail_countdown() = n; }
inline hipError_t hipMalloc(void** p, size_t n) { if (hipemu::inject_fault()) { *p = nullptr; return hipErrorOutOfMemory; } *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (hipemu::inject_fault()) return hipErrorUnknown; if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
