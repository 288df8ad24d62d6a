// host_utils.h -- multi-threaded host helpers shared by the geometry-building code.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <stdio.h>
#ifdef __linux__
#include <sched.h>
#endif
#include <algorithm>
#include <thread>
#include <vector>

namespace chroma_host {

// threads of the host-side builders: the machine's, at most 64, or CHROMA_HOST_THREADS (several
// processes building the same geometry on one node, one per GPU, share the cores)
// (the cores this process may really use: its affinity mask, capped by the cgroup's CPU quota -- a container that shows
//  256 logical CPUs with a quota of 16 must not get 64 threads per parallel loop)
inline unsigned usable_cores()
{
    static const unsigned cached = [] {
        unsigned n = std::thread::hardware_concurrency();
#ifdef __linux__
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { int c = CPU_COUNT(&set); if (c > 0) n = (unsigned)c; }
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[64]; double period = 0.0;
            if (fscanf(f, "%63s %lf", quota, &period) == 2 && quota[0] != 'm' && period > 0.0) {
                double q = atof(quota) / period;
                if (q >= 1.0) n = std::min<unsigned>(n, (unsigned)(q + 0.5));
            }
            fclose(f);
        }
#endif
        return std::max(1u, n);
    }();
    return cached;
}

inline unsigned hw_threads()
{
    unsigned n = std::max(1u, std::min(usable_cores(), 64u));
    if (const char *e = getenv("CHROMA_HOST_THREADS")) { int v = atoi(e); if (v > 0) n = std::min<unsigned>((unsigned)v, 64u); }
    return n;
}

template <class F>
void parallel_for(size_t n, F f, size_t grain = 1u << 15)
{
    unsigned nt = hw_threads();
    if (n < grain * 2) nt = 1;
    if (nt == 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        size_t lo = std::min(n, (size_t)t * chunk), hi = std::min(n, lo + chunk);
        if (lo < hi) th.emplace_back([=] { f(lo, hi); });
    }
    for (auto &t : th) t.join();
}

// One stable LSD radix pass (16-bit digit) over an array of records: sequential reads of `in`,
// scattered writes to `out`.  Returns true (and leaves `out` untouched) when every record has the
// same digit, so the caller can skip the swap.
template <class Rec, class Digit>
bool radix_pass16(size_t n, const Rec *in, Rec *out, Digit digit)
{
    unsigned nt = hw_threads();
    if (n < (1u << 16)) nt = 1;
    size_t chunk = (n + nt - 1) / nt;
    std::vector<std::vector<uint32_t>> hist(nt, std::vector<uint32_t>(65536, 0));
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                size_t lo = std::min(n, (size_t)t * chunk), hi = std::min(n, lo + chunk);
                uint32_t *h = hist[t].data();
                for (size_t i = lo; i < hi; i++) h[digit(in[i])]++;
            });
        for (auto &t : th) t.join();
    }
    {
        uint32_t d0 = digit(in[0]);
        size_t total = 0;
        for (unsigned t = 0; t < nt; t++) total += hist[t][d0];
        if (total == n) return true;
    }
    size_t run = 0;      // exclusive prefix over (digit major, thread minor): keeps the sort stable
    for (size_t d = 0; d < 65536; d++)
        for (unsigned t = 0; t < nt; t++) {
            uint32_t c = hist[t][d];
            hist[t][d] = (uint32_t)run;
            run += c;
        }
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                size_t lo = std::min(n, (size_t)t * chunk), hi = std::min(n, lo + chunk);
                uint32_t *h = hist[t].data();
                for (size_t i = lo; i < hi; i++) out[h[digit(in[i])]++] = in[i];
            });
        for (auto &t : th) t.join();
    }
    return false;
}

}  // namespace chroma_host
