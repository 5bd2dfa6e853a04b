// sa_parallel.h -- a fork-join loop over host threads for the ingest and upload stages.
#ifndef SA_PARALLEL_H
#define SA_PARALLEL_H

#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

// Number of host threads the library uses for one stage: the hardware's, at most 16.
inline unsigned sa_host_threads() {
    return std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}

// Calls f(lo, hi, t) for T contiguous slices of [0, n), T <= sa_host_threads(), each at
// least `grain` long (except when n itself is shorter); returns T.  f must not throw.
template <class F>
unsigned sa_parallel_for(uint64_t n, uint64_t grain, F f) {
    unsigned T = (unsigned)std::min<uint64_t>(sa_host_threads(), std::max<uint64_t>(1, n / std::max<uint64_t>(1, grain)));
    if (T <= 1) { f((uint64_t)0, n, 0u); return 1; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (unsigned t = 1; t < T; t++) th.emplace_back(f, n * t / T, n * (t + 1) / T, t);
    f((uint64_t)0, n / T, 0u);
    for (auto& x : th) x.join();
    return T;
}

#endif
