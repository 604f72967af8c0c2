// tsgpu_batcher.h — the in-library micro-batcher. The reference calls the scoring seam ONCE PER QUERY from many pool threads
// under a shared lock (Index::search -> search_across_fields, src/index.cpp:3488; one request thread per HTTP request,
// src/http_server.cpp:827-832). One query per launch leaves the GPU idle, so concurrent small calls are coalesced here:
//
//   * a caller parks its request; the first parked caller without a leader becomes the LEADER of the next round;
//   * the leader gathers until every thread that is inside the entry point (and not already being executed) has parked, or
//     `batch_window_us` passed, then takes an execution resource (a keyword lane / the vector executor). While every
//     resource is busy that acquisition blocks and callers keep parking: under load the rounds size themselves;
//   * the leader takes the round (FIFO, compatible requests only), promotes the next parked caller to leader — it gathers
//     and plans on the other lane while this round runs on the GPU —, executes the round as ONE batch, hands every caller
//     its slice and wakes exactly those callers (one condition variable per request: no thundering herd).
//
// Results are identical to separate calls: a batch entry never influences another (scores depend only on the document).
#pragma once
#include <condition_variable>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>
#include <atomic>

namespace tsgpu {

struct ParkedRequest {
    uint32_t units = 0;                              // queries of the call
    bool done = false, leader = false;
    int rc = 0;
    std::string err;                                 // message for the caller's thread-local error slot
    std::condition_variable cv;
};

template <class Req>                                 // Req derives from ParkedRequest
struct Combiner {
    std::mutex m;
    std::vector<Req*> pending;                       // FIFO
    uint32_t pending_units = 0;
    bool collecting = false;                         // a leader is gathering / waiting for its resource
    Req* gatherer = nullptr;                         // the leader while it waits for more callers (woken when everyone has parked)
    std::atomic<int> executing_calls{0};             // calls inside rounds that are being executed
    uint64_t rounds = 0, coalesced_calls = 0;

    // callers = threads currently inside the entry point. acquire() blocks until an execution resource is free and returns a
    // std::unique_ptr to its lock; pick(pending) moves the requests of the round out of `pending` (FIFO, compatible ones; it must take the
    // front request); exec(round, guard) runs without the combiner's mutex and fills rc / err / outputs of every request.
    template <class Acquire, class Pick, class Exec>
    void run(Req& me, const std::atomic<int>& callers, uint32_t window_us, Acquire acquire, Pick pick, Exec exec) {
        std::unique_lock<std::mutex> lk(m);
        pending.push_back(&me);
        pending_units += me.units;
        if (gatherer && (int)pending.size() >= callers.load() - executing_calls.load()) gatherer->cv.notify_one();
        if (!collecting) { collecting = true; me.leader = true; }
        for (;;) {
            if (me.done) return;
            if (!me.leader) { me.cv.wait(lk); continue; }
            // ---- leader of the next round ----
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
            gatherer = &me;
            while ((int)pending.size() < callers.load() - executing_calls.load()) {
                if (me.cv.wait_until(lk, deadline) == std::cv_status::timeout) break;
            }
            gatherer = nullptr;
            lk.unlock();
            auto guard = acquire();                  // natural batching: callers keep parking while every resource is busy
            lk.lock();
            std::vector<Req*> round;
            pick(pending, round);
            uint32_t units = 0;
            for (Req* r : round) units += r->units;
            pending_units -= units;
            executing_calls.fetch_add((int)round.size());
            rounds++;
            coalesced_calls += round.size();
            me.leader = false;
            bool mine = false;
            for (Req* r : round) mine = mine || r == &me;
            // hand the leadership on before executing: the next round is gathered and planned while this one runs
            collecting = false;
            Req* next = nullptr;
            for (Req* r : pending) if (r != &me) { next = r; break; }
            if (!mine) next = nullptr;               // (this thread is still parked: it stays in charge of the next round below)
            if (next) { collecting = true; next->leader = true; next->cv.notify_one(); }
            lk.unlock();
            exec(round, guard);
            guard.reset();                           // (a std::unique_ptr to the resource's lock)
            lk.lock();
            executing_calls.fetch_sub((int)round.size());
            for (Req* r : round) { r->done = true; if (r != &me) r->cv.notify_one(); }
            if (gatherer) gatherer->cv.notify_one();  // the threads of this round left the executing set
            // FIFO cut this thread's own request off the round: lead again, unless another leader took it over meanwhile
            if (!mine && !me.done && !collecting) { collecting = true; me.leader = true; }
        }
    }
};

}  // namespace tsgpu
