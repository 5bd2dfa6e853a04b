// write_bench.cpp -- 500 MB into one file from T threads: pwrite, mmap + memcpy, fallocate + pwrite (whole file and per
// 32 MiB chunk).  Build: g++ -O2 -pthread -o tools/write_bench tools/write_bench.cpp; run: tools/write_bench <path> <threads>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <chrono>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/wb.out";
    const size_t n = 500u << 20;
    const int T = argc > 2 ? atoi(argv[2]) : 16;
    char* buf = (char*)malloc(n);
    memset(buf, 'x', n);
    for (int method = 0; method < 4; method++) {
        unlink(path);
        double t0 = now();
        int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0644);
        if (method == 2) { if (posix_fallocate(fd, 0, n)) perror("fallocate"); }
        char* map = nullptr;
        if (method == 1) { if (ftruncate(fd, n)) perror("ftruncate"); map = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
        if (method == 3) {
            const size_t chunk = 32u << 20;
            for (size_t c0 = 0; c0 < n; c0 += chunk) {
                const size_t c1 = c0 + chunk < n ? c0 + chunk : n;
                if (posix_fallocate(fd, c0, c1 - c0)) perror("fallocate");
                std::vector<std::thread> th;
                for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
                    size_t lo = c0 + (c1 - c0) * t / T, hi = c0 + (c1 - c0) * (t + 1) / T;
                    for (size_t o = lo; o < hi; ) { ssize_t w = pwrite(fd, buf + o, hi - o, o); if (w <= 0) break; o += w; }
                });
                for (auto& x : th) x.join();
            }
            close(fd);
            double t1 = now();
            printf("%s method 3 (fallocate per 32 MiB chunk + pwrite) threads %d: %.3f s  %.2f GB/s\n", path, T, t1 - t0, n / (t1 - t0) * 1e-9);
            continue;
        }
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
            size_t lo = n * t / T, hi = n * (t + 1) / T;
            if (method == 1) memcpy(map + lo, buf + lo, hi - lo);
            else for (size_t o = lo; o < hi; ) { size_t c = hi - o < (4u << 20) ? hi - o : (4u << 20); ssize_t w = pwrite(fd, buf + o, c, o); if (w <= 0) { perror("pwrite"); break; } o += w; }
        });
        for (auto& x : th) x.join();
        if (map) munmap(map, n);
        close(fd);
        double t1 = now();
        printf("%s method %d (%s) threads %d: %.3f s  %.2f GB/s\n", path, method, method == 0 ? "pwrite" : method == 1 ? "mmap+memcpy" : "fallocate+pwrite", T, t1 - t0, n / (t1 - t0) * 1e-9);
    }
    unlink(path);
    return 0;
}
