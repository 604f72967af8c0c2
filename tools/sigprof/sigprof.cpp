// sigprof.cpp — experiment tooling (NOT part of the product): a process-wide CPU-time sampling profiler for boxes without perf.
// ITIMER_PROF delivers SIGPROF to a thread that is burning CPU (user or kernel time); the handler stores its backtrace. The report
// lists the stacks as "library+offset" frames (symbolised afterwards with llvm-symbolizer against the same .so files) with counts.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <execinfo.h>
#include <map>
#include <signal.h>
#include <string>
#include <sys/time.h>
#include <vector>

namespace {
const int DEPTH = 14, CAP = 1 << 18;
struct Sample { void* pc[DEPTH]; int n; };
Sample* g_s = nullptr;
std::atomic<uint32_t> g_n{0};
void on_prof(int, siginfo_t*, void*) {
    const uint32_t i = g_n.fetch_add(1);
    if (i >= (uint32_t)CAP) return;
    g_s[i].n = backtrace(g_s[i].pc, DEPTH);
}
}
extern "C" {
int sigprof_start(int hz) {
    if (!g_s) g_s = new Sample[CAP];
    void* warm[4]; (void)backtrace(warm, 4);                 // (loads libgcc's unwinder outside the handler)
    g_n = 0;
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    if (sigaction(SIGPROF, &sa, nullptr)) return -1;
    itimerval tv; tv.it_interval.tv_sec = 0; tv.it_interval.tv_usec = 1000000 / hz; tv.it_value = tv.it_interval;
    return setitimer(ITIMER_PROF, &tv, nullptr);
}
int sigprof_stop(const char* path, int skip, int keep) {
    itimerval tv; memset(&tv, 0, sizeof tv);
    setitimer(ITIMER_PROF, &tv, nullptr);
    const uint32_t n = g_n.load() < (uint32_t)CAP ? g_n.load() : (uint32_t)CAP;
    std::map<std::string, uint32_t> agg;
    for (uint32_t i = 0; i < n; i++) {
        std::string key;
        for (int f = skip; f < g_s[i].n && f < skip + keep; f++) {
            Dl_info di;
            char buf[512];
            if (dladdr(g_s[i].pc[f], &di) && di.dli_fname) {
                const char* base = strrchr(di.dli_fname, '/');
                snprintf(buf, sizeof buf, "%s+0x%lx[%s]", base ? base + 1 : di.dli_fname, (unsigned long)((char*)g_s[i].pc[f] - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
            } else snprintf(buf, sizeof buf, "?+%p", g_s[i].pc[f]);
            if (!key.empty()) key += ";";
            key += buf;
        }
        agg[key]++;
    }
    FILE* fp = fopen(path, "w");
    if (!fp) return -1;
    fprintf(fp, "# %u samples\n", n);
    for (auto& kv : agg) fprintf(fp, "%u %s\n", kv.second, kv.first.c_str());
    fclose(fp);
    return (int)n;
}
}
