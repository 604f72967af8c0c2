// TEST INFRASTRUCTURE ONLY — self-test of the SIMT emulator's barrier-site check (tests/hipemu/hip/hip_runtime.h): a workgroup whose threads reach
// DIFFERENT textual __syncthreads() must be reported (fatal with HIPEMU_STRICT_BARRIERS=1); one whose threads all take the same path must not.
// The racy kernel is the pattern round 5 found in a first form of gb_select_kernel: a shared count read next to other threads' updates decides
// whether a thread enters barrier-carrying code. argv[1]: "race" | "clean". Built and run by tests/test_hipemu_selftest.py.
#include <hip/hip_runtime.h>
#include <cstring>

__device__ inline void compaction(unsigned* s_cnt) {          // barrier-carrying code every thread of the workgroup must enter together
    __syncthreads();
    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();
}

__global__ void racy_kernel(unsigned* out) {
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int it = 0; it < 4; it++) {
        if (s_cnt + 64 > 100) compaction(&s_cnt);            // the count is read while earlier threads of the same iteration already add to it
        atomicAdd(&s_cnt, 1u);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_cnt;
}

__global__ void clean_kernel(unsigned* out) {
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int it = 0; it < 4; it++) {
        const unsigned held = s_cnt;                          // read BEFORE the counting barrier, decide with the count it returns
        const unsigned n = (unsigned)__syncthreads_count(1);
        if (held + n > 100) compaction(&s_cnt);
        atomicAdd(&s_cnt, 1u);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_cnt;
}

int main(int argc, char** argv) {
    unsigned out[2] = {0, 0};
    if (argc > 1 && !strcmp(argv[1], "race")) hipLaunchKernelGGL(racy_kernel, dim3(2), dim3(64), 0, nullptr, out);
    else hipLaunchKernelGGL(clean_kernel, dim3(2), dim3(64), 0, nullptr, out);
    printf("done %u %u\n", out[0], out[1]);
    return 0;
}
