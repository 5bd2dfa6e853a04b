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

// out[0] = 0, out[s + 1] = out[s] + val(s) for s in [0, n), on the host threads: slice sums, then every slice from its
// start value (the 4 001 280-row data set: three such sums took 17 ms of its upload on one thread)
template <class F>
void prefix_sums(uint64_t n, uint64_t* out, F val) {
    std::vector<uint64_t> start(sa_host_threads() + 1, 0);
    sa_parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned t) {
        uint64_t a = 0;
        for (uint64_t s = lo; s < hi; s++) a += val(s);
        start[t + 1] = a;
    });
    for (size_t t = 1; t < start.size(); t++) start[t] += start[t - 1];
    out[0] = 0;
    sa_parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned t) {
        uint64_t a = start[t];
        for (uint64_t s = lo; s < hi; s++) { a += val(s); out[s + 1] = a; }
    });
}


#endif
