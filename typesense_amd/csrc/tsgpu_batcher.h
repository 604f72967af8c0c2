// tsgpu_batcher.h — the in-library micro-batcher. The reference calls the scoring seam ONCE PER QUERY from many pool threads
// under a shared lock (Index::search -> search_across_fields, src/index.cpp:3488; one request thread per HTTP request,
// src/http_server.cpp:827-832). One query per launch leaves the GPU idle, so concurrent small calls are coalesced here:
//
//   * a caller parks its request; the first parked caller without a leader becomes the LEADER of the next round;
//   * the leader gathers until every thread that is inside the entry point (and not already being executed) has parked, or
//     `batch_window_us` passed (10 us: a free lane must not idle — with 80 us the lanes were busy 2.5 of 4 under 256 callers), then
//     takes an execution resource (a keyword lane / the vector executor). While every resource is busy that acquisition blocks and
//     callers keep parking: under load the rounds size themselves;
//   * the leader takes the round (FIFO, compatible requests only), promotes the next parked caller to leader — it gathers
//     and plans on another lane while this round runs on the GPU —, executes the round as ONE batch and hands every caller
//     its slice.
//
// Waiting and waking (hundreds of request threads wake up per millisecond here): a parked caller waits on its request's state
// word — a short spin, then a futex wait on one of 16 BANK words (consecutive arrivals share a bank, a round is a run of
// consecutive arrivals). The leader publishes the states, bumps the touched banks and wakes each with ONE futex call: a round of
// 64 callers costs ~5 wake syscalls instead of 64, the woken threads start in parallel and none of them needs the combiner's
// mutex to leave.
//
// Results are identical to separate calls: a batch entry never influences another (scores depend only on the document).
#pragma once
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#endif

namespace tsgpu {

inline void futex_wait_u32(std::atomic<uint32_t>* w, uint32_t expected, long timeout_us) {      // timeout_us <= 0: no timeout
#if defined(__linux__)
    timespec ts;
    ts.tv_sec = timeout_us / 1000000;
    ts.tv_nsec = (long)(timeout_us % 1000000) * 1000;
    (void)syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, expected, timeout_us > 0 ? &ts : nullptr, nullptr, 0);
#else
    (void)w; (void)expected; (void)timeout_us;
    std::this_thread::yield();
#endif
}
inline void futex_wake_all(std::atomic<uint32_t>* w) {
#if defined(__linux__)
    (void)syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
#else
    (void)w;
#endif
}

struct ParkedRequest {
    enum : uint32_t { PARKED = 0, LEADER = 1, DONE = 2 };
    uint32_t units = 0;                              // queries of the call
    std::atomic<uint32_t> state{PARKED};
    uint32_t bank = 0;                               // which wake word this request sleeps on
    int rc = 0;
    std::string err;                                 // message for the caller's thread-local error slot
};

template <class Req>                                 // Req derives from ParkedRequest
struct Combiner {
    static const int BANKS = 16, PER_BANK = 16;
    std::mutex m;
    std::vector<Req*> pending;                       // FIFO
    uint32_t pending_units = 0;
    bool collecting = false;                         // a leader exists (gathering / waiting for its resource)
    bool gathering = false;                          // ... and sleeps on gather_word until everyone has parked
    int gather_target = 0;                           // ... "everyone" = this many calls (parked + executing)
    std::atomic<uint32_t> gather_word{0};
    std::atomic<int> executing_calls{0};             // calls inside rounds that are being executed
    uint64_t rounds = 0, coalesced_calls = 0, arrivals = 0;
    struct alignas(64) Bank { std::atomic<uint32_t> gen{0}; } banks[BANKS];

    // publish `new_state` to the requests and wake their banks. A request lives on its caller's stack: its state is stored LAST —
    // from then on it is never touched again (a caller that sees DONE leaves at once); the bank words belong to the combiner.
    void wake(Req* const* reqs, size_t n, uint32_t new_state) {
        bool touched[BANKS] = {};
        for (size_t i = 0; i < n; i++) { touched[reqs[i]->bank] = true; reqs[i]->state.store(new_state, std::memory_order_release); }
        for (int b = 0; b < BANKS; b++) if (touched[b]) { banks[b].gen.fetch_add(1, std::memory_order_release); futex_wake_all(&banks[b].gen); }
    }

    // callers = threads currently inside the entry point. acquire() blocks until an execution resource is free and returns a
    // std::unique_ptr to its lock; pick(pending, round) moves the requests of the round out of `pending` (FIFO, compatible ones; it
    // MUST take the front request — the leader's own); exec(round, guard) runs without the combiner's mutex and fills rc / err /
    // outputs of every request.
    // post_window_us: a SECOND gather after the resource has been acquired, for rounds whose cost hardly depends on their size (the
    // vector scan streams the whole collection once per round: 3.3 ms for 64 queries, 4.6 ms for 256). The callers of the round that
    // just finished are OUTSIDE the entry point for a moment (returning their result, calling again), so "everyone inside has parked"
    // is true too early and the rounds ping-pong between two halves of the callers; here the leader waits until as many have parked as
    // were recently seen inside at once (peak, decaying), or the window passes.
    int peak_callers = 0;
    template <class Acquire, class Pick, class Exec>
    void run(Req& me, const std::atomic<int>& callers, uint32_t window_us, Acquire acquire, Pick pick, Exec exec, uint32_t post_window_us = 0) {
        {
            std::lock_guard<std::mutex> lk(m);
            me.bank = (uint32_t)((arrivals++ / PER_BANK) % BANKS);
            pending.push_back(&me);
            pending_units += me.units;
            peak_callers = std::max(peak_callers, callers.load());
            if (gathering && (int)pending.size() >= gather_target - executing_calls.load()) { gather_word.fetch_add(1); futex_wake_all(&gather_word); }
            // no leader: pending was empty (a leader that leaves requests behind always promotes the front one), so this request is the front
            if (!collecting) { collecting = true; me.state.store(ParkedRequest::LEADER, std::memory_order_relaxed); }
        }
        // ---- parked: spin briefly, then sleep on the bank word ----
        uint32_t st = me.state.load(std::memory_order_acquire);
        // (the spin only pays when a round completes within microseconds, i.e. with few callers; with many it is CPU the quota may not have)
        const int spins = callers.load(std::memory_order_relaxed) >= 48 ? 0 : 200;
        for (int spin = 0; st == ParkedRequest::PARKED && spin < spins; spin++) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            st = me.state.load(std::memory_order_acquire);
        }
        while (st == ParkedRequest::PARKED) {
            const uint32_t gen = banks[me.bank].gen.load(std::memory_order_acquire);
            st = me.state.load(std::memory_order_acquire);
            if (st != ParkedRequest::PARKED) break;
            futex_wait_u32(&banks[me.bank].gen, gen, 0);
            st = me.state.load(std::memory_order_acquire);
        }
        if (st == ParkedRequest::DONE) return;
        // ---- leader of the next round (this request is the front of `pending`) ----
        std::unique_lock<std::mutex> lk(m);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
        auto gather = [&](std::chrono::steady_clock::time_point until, bool use_peak) {
            gathering = true;
            for (;;) {
                gather_target = use_peak ? std::max(peak_callers, callers.load()) : callers.load();
                if ((int)pending.size() >= gather_target - executing_calls.load()) break;
                const auto now = std::chrono::steady_clock::now();
                if (now >= until) break;
                const uint32_t w = gather_word.load();
                lk.unlock();
                futex_wait_u32(&gather_word, w, (long)std::max<long long>(1, std::chrono::duration_cast<std::chrono::microseconds>(until - now).count()));
                lk.lock();
            }
            gathering = false;
        };
        gather(deadline, false);
        lk.unlock();
        auto guard = acquire();                      // natural batching: callers keep parking while every resource is busy
        lk.lock();
        if (post_window_us) {
            gather(std::chrono::steady_clock::now() + std::chrono::microseconds(post_window_us), true);
            peak_callers = std::max(callers.load(), peak_callers - std::max(1, peak_callers / 8));      // (decays when the load drops)
        }
        std::vector<Req*> round;
        pick(pending, round);
        uint32_t units = 0;
        for (Req* r : round) units += r->units;
        pending_units -= units;
        executing_calls.fetch_add((int)round.size());
        rounds++;
        coalesced_calls += round.size();
        // hand the leadership on before executing: the next round is gathered and planned while this one runs
        collecting = false;
        if (!pending.empty()) { collecting = true; Req* next = pending.front(); wake(&next, 1, ParkedRequest::LEADER); }
        lk.unlock();
        exec(round, guard);
        guard.reset();                               // (a std::unique_ptr to the resource's lock)
        executing_calls.fetch_sub((int)round.size());
        // results are in place: release the round's callers — no mutex on this path
        std::vector<Req*> others;
        for (Req* r : round) if (r != &me) others.push_back(r);
        wake(others.data(), others.size(), ParkedRequest::DONE);
        {
            std::lock_guard<std::mutex> lk2(m);
            if (gathering) { gather_word.fetch_add(1); futex_wake_all(&gather_word); }          // this round's threads left the executing set
        }
    }
};

}  // namespace tsgpu
