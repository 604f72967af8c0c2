// tsgpu_batcher.h — the in-library micro-batcher. The reference calls the scoring seam ONCE PER QUERY from many pool threads
// under a shared lock (Index::search -> search_across_fields, src/index.cpp:3488; one request thread per HTTP request,
// src/http_server.cpp:827-832). One query per launch leaves the GPU idle, so concurrent small calls are coalesced here:
//
//   * a caller pushes its request on a lock-free stack; an arrival that finds no leader makes itself the LEADER of the next round
//     (one exchange), everybody else parks;
//   * the leader gathers until every thread that is inside the entry point (and not already being executed) has parked, or
//     `batch_window_us` passed (10 us: a free lane must not idle — with 80 us the lanes were busy 2.5 of 4 under 256 callers), then
//     takes an execution resource (a keyword lane / the vector executor). While every resource is busy that acquisition blocks and
//     callers keep parking: under load the rounds size themselves;
//   * the leader takes the whole stack, picks its round (FIFO, compatible requests only; the rest goes back), steps down — an
//     arrival or, if requests are left, a promoted parked caller leads the next round, which is gathered and planned on another lane
//     while this one runs on the GPU —, executes the round as ONE batch and hands every caller its slice.
//
// Waiting and waking (hundreds of request threads wake up per millisecond here): a parked caller waits on its request's state
// word — a short spin, then a futex wait on one of 16 BANK words (consecutive arrivals share a bank, a round is a run of
// consecutive arrivals). The leader publishes the states, bumps the touched banks and wakes each with ONE futex call: a round of
// 64 callers costs ~5 wake syscalls instead of 64, the woken threads start in parallel; there is no mutex anywhere on the path.
//
// Results are identical to separate calls: a batch entry never influences another (scores depend only on the document).
#pragma once
#include <algorithm>
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
    enum : uint32_t { PARKED = 0, LEADER = 1, DONE = 2, INROUND = 3 };     // INROUND: taken off the stack into a round that is being executed
    uint32_t units = 0;                              // queries of the call
    std::atomic<uint32_t> state{PARKED};
    uint32_t bank = 0;                               // which wake word this request sleeps on
    int rc = 0;
    std::string err;                                 // message for the caller's thread-local error slot
    ParkedRequest* next = nullptr;                   // the arrival stack's link
    uint64_t seq = 0;                                // arrival number (FIFO order of a round)
};

// LOCK-FREE on the arrival path. As a std::mutex-protected queue the combiner CONVOYED under hundreds of request threads: every
// contended unlock is a futex wake, every waiter a context switch, the lock's throughput falls to one hand-over per context switch —
// 48 % of the process's CPU time was pthread_mutex_lock / _unlock, and with the request threads on a CPU quota that time IS the
// throughput (profiles/r03/sigprof_256threads_before.txt; a spinning lock is worse: a holder that the quota throttles stalls every
// spinner). Now: an arrival pushes its request on an intrusive stack (one CAS) and tries to become the leader (one exchange); the
// leader takes the whole stack with one exchange. No lock anywhere.
template <class Req>                                 // Req derives from ParkedRequest
struct Combiner {
    static const int BANKS = 16, PER_BANK = 16;
    std::atomic<ParkedRequest*> head{nullptr};       // arrivals, newest first
    std::atomic<int> n_pending{0};
    std::atomic<bool> collecting{false};             // a leader exists (gathering / waiting for its resource / taking the stack)
    std::atomic<bool> gathering{false};              // ... and sleeps on gather_word until everyone has parked
    std::atomic<int> gather_target{0};               // ... "everyone" = this many calls (parked + executing)
    std::atomic<uint32_t> gather_word{0};
    std::atomic<int> executing_calls{0};             // calls inside rounds that are being executed
    std::atomic<uint64_t> rounds{0}, coalesced_calls{0}, arrivals{0};
    std::atomic<int> peak_callers{0};
    struct alignas(64) Bank { std::atomic<uint32_t> gen{0}; } banks[BANKS];

    // publish `new_state` to the requests and wake their banks. A request lives on its caller's stack: its state is stored LAST —
    // from then on it is never touched again (a caller that sees DONE leaves at once); the bank words belong to the combiner.
    void wake(Req* const* reqs, size_t n, uint32_t new_state) {
        bool touched[BANKS] = {};
        for (size_t i = 0; i < n; i++) { touched[reqs[i]->bank] = true; reqs[i]->state.store(new_state, std::memory_order_release); }
        for (int b = 0; b < BANKS; b++) if (touched[b]) { banks[b].gen.fetch_add(1, std::memory_order_release); futex_wake_all(&banks[b].gen); }
    }
    void push(ParkedRequest* r) {
        ParkedRequest* h = head.load(std::memory_order_relaxed);
        do { r->next = h; } while (!head.compare_exchange_weak(h, r));          // (sequentially consistent: see the leader election in run())
        n_pending.fetch_add(1);
    }
    // the leadership passes to a parked request (any one: a leader takes the WHOLE stack, its own request included)
    void promote_if_pending() {
        if (head.load() == nullptr) return;
        if (collecting.exchange(true)) return;       // an arrival made itself the leader meanwhile
        // (only the flag's holder takes the stack: whatever is on top now stays parked until its state changes)
        Req* next = static_cast<Req*>(head.load(std::memory_order_acquire));
        if (!next) { collecting.store(false); if (head.load() != nullptr) promote_if_pending(); return; }
        wake(&next, 1, ParkedRequest::LEADER);
    }

    // callers = threads currently inside the entry point. acquire() blocks until an execution resource is free and returns a
    // std::unique_ptr to its lock; pick(pending, round) moves the requests of the round out of `pending` (FIFO, compatible ones; it
    // MUST take the front request — the leader's own); exec(round, guard) fills rc / err / outputs of every request.
    // post_window_us: a SECOND gather after the resource has been acquired, for rounds whose cost hardly depends on their size (the
    // vector scan streams the whole collection once per round: 3.3 ms for 64 queries, 4.6 ms for 256). The callers of the round that
    // just finished are OUTSIDE the entry point for a moment (returning their result, calling again), so "everyone inside has parked"
    // is true too early and the rounds ping-pong between two halves of the callers; here the leader waits until as many have parked as
    // were recently seen inside at once (peak, decaying), or the window passes.
    template <class Acquire, class Pick, class Exec>
    void run(Req& me, const std::atomic<int>& callers, uint32_t window_us, Acquire acquire, Pick pick, Exec exec, uint32_t post_window_us = 0) {
        me.seq = arrivals.fetch_add(1, std::memory_order_relaxed);
        me.bank = (uint32_t)((me.seq / PER_BANK) % BANKS);
        {
            const int c = callers.load(std::memory_order_relaxed);
            int p = peak_callers.load(std::memory_order_relaxed);
            while (c > p && !peak_callers.compare_exchange_weak(p, c, std::memory_order_relaxed)) {}
        }
        push(&me);
        if (gathering.load(std::memory_order_acquire) && n_pending.load() >= gather_target.load() - executing_calls.load()) { gather_word.fetch_add(1); futex_wake_all(&gather_word); }
        // no leader: this thread leads the next round (it is awake, nobody has to be woken for it). Arrival: push, THEN read the flag; a
        // leader that steps down: clear the flag, THEN read the stack (promote_if_pending) — all four sequentially consistent, so at
        // least one of the two sees the other (Dekker): a request is never left parked without a leader.
        bool lead = !collecting.load() && !collecting.exchange(true);
        if (lead && me.state.load() != ParkedRequest::PARKED) {
            // the previous leader took this request into ITS round between the push and the election (it marks its round INROUND before
            // it steps down, and the flag was taken after that; the round may even be DONE already): not a leader — hand the flag on
            // and wait for that round
            collecting.store(false);
            promote_if_pending();
            lead = false;
        }
        if (!lead) {
            // ---- parked: spin briefly, then sleep on the bank word ----
            uint32_t st = me.state.load(std::memory_order_acquire);
            // (the spin only pays when a round completes within microseconds, i.e. with few callers; with many it is CPU the quota may not have)
            const int spins = callers.load(std::memory_order_relaxed) >= 48 ? 0 : 200;
            auto waiting = [](uint32_t x) { return x == ParkedRequest::PARKED || x == ParkedRequest::INROUND; };
            for (int spin = 0; waiting(st) && spin < spins; spin++) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                st = me.state.load(std::memory_order_acquire);
            }
            while (waiting(st)) {
                const uint32_t gen = banks[me.bank].gen.load(std::memory_order_acquire);
                st = me.state.load(std::memory_order_acquire);
                if (!waiting(st)) break;
                futex_wait_u32(&banks[me.bank].gen, gen, 0);
                st = me.state.load(std::memory_order_acquire);
            }
            if (st == ParkedRequest::DONE) return;
        }
        // ---- leader of the next round (this request is on the stack; the leader takes all of it) ----
        auto gather = [&](std::chrono::steady_clock::time_point until, bool use_peak) {
            gathering.store(true, std::memory_order_release);
            for (;;) {
                const int target = use_peak ? std::max(peak_callers.load(), callers.load()) : callers.load();
                gather_target.store(target);
                if (n_pending.load(std::memory_order_acquire) >= target - executing_calls.load()) break;
                const auto now = std::chrono::steady_clock::now();
                if (now >= until) break;
                const uint32_t w = gather_word.load();
                if (n_pending.load(std::memory_order_acquire) >= target - executing_calls.load()) break;
                futex_wait_u32(&gather_word, w, (long)std::max<long long>(1, std::chrono::duration_cast<std::chrono::microseconds>(until - now).count()));
            }
            gathering.store(false, std::memory_order_release);
        };
        if (window_us) gather(std::chrono::steady_clock::now() + std::chrono::microseconds(window_us), false);
        auto guard = acquire();                      // natural batching: callers keep parking while every resource is busy
        if (post_window_us) {
            gather(std::chrono::steady_clock::now() + std::chrono::microseconds(post_window_us), true);
            const int c = callers.load(), p = peak_callers.load();
            peak_callers.store(std::max(c, p - std::max(1, p / 8)));                    // (decays when the load drops)
        }
        // take everything that has arrived, oldest first, the leader's own request in front
        std::vector<Req*> pending, round;
        for (ParkedRequest* r = head.exchange(nullptr, std::memory_order_acquire); r; r = r->next) pending.push_back(static_cast<Req*>(r));
        n_pending.fetch_sub((int)pending.size(), std::memory_order_relaxed);
        std::sort(pending.begin(), pending.end(), [&](const Req* a, const Req* b) { return (a == &me) != (b == &me) ? a == &me : a->seq < b->seq; });
        pick(pending, round);
        for (Req* r : round) if (r != &me) r->state.store(ParkedRequest::INROUND);       // (before the flag is released: see the election above)
        executing_calls.fetch_add((int)round.size());
        rounds.fetch_add(1, std::memory_order_relaxed);
        coalesced_calls.fetch_add(round.size(), std::memory_order_relaxed);
        // what the round could not take goes back (its links are rewritten: nobody else reads a request that is not on the stack)
        for (auto it = pending.rbegin(); it != pending.rend(); ++it) push(*it);
        // hand the leadership on before executing: the next round is gathered and planned while this one runs
        collecting.store(false);
        promote_if_pending();
        exec(round, guard);
        guard.reset();                               // (a std::unique_ptr to the resource's lock)
        executing_calls.fetch_sub((int)round.size());
        // results are in place: release the round's callers
        std::vector<Req*> others;
        for (Req* r : round) if (r != &me) others.push_back(r);
        wake(others.data(), others.size(), ParkedRequest::DONE);
        if (gathering.load(std::memory_order_acquire)) { gather_word.fetch_add(1); futex_wake_all(&gather_word); }          // this round's threads left the executing set
    }
};

}  // namespace tsgpu
