// row_bench.hip -- issue cost of one values-only DP row (64 columns, two pairs per lane) as the checkpointing
// long-sequence kernel runs it, in isolation: no global memory, 1 / 2 / 4 waves per SIMD on every SIMD.
// Variants: table entries from LDS as in the kernel, or from registers (no LDS at all).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sequencealigner_amd/csrc tools/row_bench.hip -o tools/row_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define SA_LCK_ROW_BENCH 1
#include "sa_kernels_long_ckpt.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 20000;

__global__ void __launch_bounds__(64, 4) k_rows(unsigned* out, unsigned long long* cyc, unsigned seed) {
    __shared__ int32_t prof[21 * SA_LCK_PS + 16];
    for (int e = threadIdx.x; e < 21 * SA_LCK_PS; e += 64) prof[e] = 0x00030003 + (e % 7) * 0x00010001;
    __syncthreads();
    uint32_t R[SA_LP];
#pragma unroll
    for (int c = 0; c < SA_LP; c++) R[c] = seed + c;
    uint32_t b = (threadIdx.x * 7 + seed) % 20;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < ROWS; r += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int oe = 8;
            asm volatile("" : "+s"(oe));
            fill_row_score<true>(R, oe, R[63] & 0xff, R[0] & 0xff, prof + b * SA_LCK_PS);
            b = (b + 3) % 20;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < SA_LP; c++) s ^= R[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    unsigned* d_out; CHECK(hipMalloc(&d_out, 4 * ncu * 16 * 64));
    unsigned long long* d_cyc; CHECK(hipMalloc(&d_cyc, 8 * ncu * 16));
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = ncu * 4 * wps;
        k_rows<<<blocks, 64>>>(d_out, d_cyc, 1);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 3; r++) k_rows<<<blocks, 64>>>(d_out, d_cyc, r);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(blocks);
        CHECK(hipMemcpy(cyc.data(), d_cyc, 8 * blocks, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        const double med = (double)cyc[blocks / 2];
        const double us = ms * 1e3 / 3;
        // s_memtime counts at 100 MHz: wave lifetime in us = med / 100
        printf("waves/SIMD=%d  %.1f ns per row per wave  %.1f ns per row per SIMD  -> %.1f TCUPS chip-wide (wave lifetime %.0f us, launch %.0f us)\n",
               wps, med * 10.0 / ROWS, us * 1e3 / ROWS / wps, 8192.0 * ROWS * blocks / (us * 1e-6) / 1e12, med / 100.0, us);
    }
    return 0;
}
