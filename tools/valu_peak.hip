// valu_peak.hip -- measures the integer VALU issue rate of the device for the
// instructions the DP inner loop is made of, at 1, 2 and 4 waves per SIMD.
// The result calibrates the "peak" of the INT32 roofline used by bench.py.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(unsigned* out, unsigned seed) {
    unsigned a[CHAINS], b = seed + threadIdx.x, c = seed * 3 + 1;
    for (int k = 0; k < CHAINS; k++) a[k] = threadIdx.x * 7 + k;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < CHAINS; k++) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == 1) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == 2) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                        : "+v"(a[k]) : "v"(b) : "vcc");
            if (KIND == 3) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == 4) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == 5) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == 6) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                                        : "+v"(a[k]) : "v"(b));
            if (KIND == 7) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == 8) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
        }
    }
    unsigned s = 0;
    for (int k = 0; k < CHAINS; k++) s ^= a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int instr_per_step, unsigned* d_out, int ncu, double clk_ghz) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = ncu * wps;             // 256 threads = 4 waves = one wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_valu<KIND><<<blocks, 256>>>(d_out, 1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) k_valu<KIND><<<blocks, 256>>>(d_out, r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double lane_ops = 5.0 * blocks * 256.0 * ITERS * CHAINS * instr_per_step;
        double tops = lane_ops / (ms * 1e-3) / 1e12;
        double per_cu_clk = lane_ops / (ms * 1e-3) / ncu / (clk_ghz * 1e9);
        printf("%-28s waves/SIMD=%d  %.2f T lane-instr/s  (%.1f lanes/clk/CU at %.2f GHz)\n",
               name, wps, tops, per_cu_clk, clk_ghz);
    }
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    double clk = p.clockRate / 1e6;
    printf("device %s CUs=%d clock=%.3f GHz\n", p.name, ncu, clk);
    unsigned* d_out; CHECK(hipMalloc(&d_out, sizeof(unsigned) * ncu * 4 * 256 * 2));
    run<0>("v_add_u32", 1, d_out, ncu, clk);
    run<1>("v_max3_i32", 1, d_out, ncu, clk);
    run<2>("v_cmp_ne+v_addc (2 instr)", 2, d_out, ncu, clk);
    run<3>("v_pk_add_i16", 1, d_out, ncu, clk);
    run<4>("v_pk_max_i16", 1, d_out, ncu, clk);
    run<5>("v_pk_mad_u16", 1, d_out, ncu, clk);
    run<6>("v_add_u32_sdwa byte sel", 1, d_out, ncu, clk);
    run<7>("v_add3_u32", 1, d_out, ncu, clk);
    run<8>("v_lshl_or_b32", 1, d_out, ncu, clk);
    run<9>("v_cndmask_b32", 1, d_out, ncu, clk);
    CHECK(hipFree(d_out));
    return 0;
}
