// valu_peak.hip -- measures the VALU issue cost (shader cycles per wave64
// instruction per SIMD) of the instructions the DP inner loop is made of, at 1,
// 2 and 4 waves per SIMD, plus the wall-clock lane-instruction rate.  The
// results calibrate the INT32 roofline peak used by bench.py and guide the
// instruction selection of the fill loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 20000;
constexpr int CHAINS = 8;

#define K_ADD 0
#define K_MAX3 1
#define K_CMPADDC_VCC 2
#define K_PKADD 3
#define K_SDWA 4
#define K_MAX 5
#define K_CMP 6
#define K_ADDC 7
#define K_SUB 8
#define K_MIN 9
#define K_LSHLOR 10
#define K_BFE 11
#define K_AND 12
#define K_LSHR 13
#define K_CELL6 14
#define K_CELL_B 15
#define K_CELL_C 16
#define K_MOV 17
#define K_CMPADDC_SGPR 18
#define K_PKMAX 19
#define K_PKSUB 20
#define K_PKMIN 21
#define K_PKMAD 22
#define K_PERM 23
#define K_MADU24 24
#define K_CELL_D 25
#define K_CELL_SIGN 26
#define K_CELL_TAG 27
#define K_ALIGNBIT 28
#define K_ANDOR 29
#define K_PKMAX3F 30
#define K_BFI 31
#define K_CELL_PK 32
#define K_CELL_PK3 33
#define K_LSHR64 34
#define K_FFBL 35
#define K_ADDCLAMP 36
#define K_OR 37
#define K_XOR 38
#define K_LSHL 39
#define K_NOT 40
#define K_BCNT 41
#define K_SUBLIT 42
#define K_CMPSDWA 43
#define K_MIN3 44
#define K_MED3 45
#define K_ANDLIT 46
#define K_LSHRV 47
#define K_SCORE_CHAIN 48
#define K_SCORE_CHAIN_LDS 49

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(unsigned* out, unsigned long long* cyc, unsigned seed) {
    unsigned a[CHAINS], b = seed + threadIdx.x, c = seed * 3 + 1, d = seed ^ 0x55;
    unsigned f[CHAINS], g[CHAINS];
    for (int k = 0; k < CHAINS; k++) { a[k] = threadIdx.x * 7 + k; f[k] = k; g[k] = k * 3; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < CHAINS; k++) {
            if (KIND == K_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_MAX3) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_CMPADDC_VCC) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
            if (KIND == K_PKADD) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_SDWA) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[k]) : "v"(b));
            if (KIND == K_MAX) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_CMP) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
            if (KIND == K_ADDC) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[k]) : : "vcc");
            if (KIND == K_SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_BFE) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[k]));
            if (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_LSHR) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[k]));
            if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_PKMAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_PKSUB) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_PKMIN) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_PKMAD) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_CMPADDC_SGPR) {
                // compare result in its own SGPR pair per chain, carry-out discarded into another pair
                asm volatile("v_cmp_ne_u32_e64 s[20:21], %0, %1\n\tv_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]"
                             : "+v"(a[k]) : "v"(b) : "s20", "s21", "s22", "s23");
            }
            if (KIND == K_CELL6) {
                // add, max3, (cmp, addc) x2 : six instructions per DP cell
                unsigned m, h;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(h) : "v"(m), "v"(c), "v"(d));
                asm volatile("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(f[k]) : "v"(h), "v"(m) : "vcc");
                asm volatile("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(g[k]) : "v"(h), "v"(c) : "vcc");
                a[k] = h;
            }
            if (KIND == K_CELL_B) {
                // add, max, max, (sub, min, lshl_or) x2 : nine instructions, no SGPR traffic
                unsigned m, h, x, y;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(h) : "v"(m), "v"(c));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(h) : "v"(d));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(h), "v"(m));
                asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(1u));
                asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(f[k]) : "v"(x));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(y) : "v"(h), "v"(c));
                asm volatile("v_min_u32 %0, %0, %1" : "+v"(y) : "v"(1u));
                asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(g[k]) : "v"(y));
                a[k] = h;
            }
            if (KIND == K_CELL_C) {
                // add, max, max, cmp+addc (separate SGPR pairs) x2
                unsigned m, h;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(h) : "v"(m), "v"(c));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(h) : "v"(d));
                asm volatile("v_cmp_ne_u32_e64 s[20:21], %1, %2\n\tv_cmp_ne_u32_e64 s[24:25], %1, %3\n\t"
                             "v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]"
                             : "+v"(f[k]) : "v"(h), "v"(m), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25");
                asm volatile("v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[24:25]" : "+v"(g[k]) : : "s22", "s23", "s24", "s25");
                a[k] = h;
            }
            if (KIND == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[k]) : "v"(b));
            if (KIND == K_LSHR64) { unsigned long long q = ((unsigned long long)f[k] << 32) | a[k]; asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "v"(b)); a[k] = (unsigned)q; f[k] = (unsigned)(q >> 32); }
            if (KIND == K_FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[k]));
            if (KIND == K_ADDCLAMP) asm volatile("v_add_u32 %0, %0, 32 clamp" : "+v"(a[k]));
            if (KIND == K_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_LSHL) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[k]));
            if (KIND == K_NOT) asm volatile("v_not_b32 %0, %0" : "+v"(a[k]));
            if (KIND == K_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
            if (KIND == K_SUBLIT) asm volatile("v_subrev_u32 %0, 0x1010101, %0" : "+v"(a[k]));
            if (KIND == K_CMPSDWA) asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:DWORD src1_sel:BYTE_2" : : "v"(a[k]), "v"(b) : "vcc");
            if (KIND == K_MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_ANDLIT) asm volatile("v_and_b32 %0, 0x7f7f7f7f, %0" : "+v"(a[k]));
            if (KIND == K_LSHRV) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
            if (KIND == K_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, 1" : "+v"(a[k]) : "v"(b));
            if (KIND == K_PKMAX3F) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == K_CELL_PK) {
                // packed 16-bit cells, 2-bit tags: add, 2 x pk_max, and, lshl_or, and, or per two cells
                unsigned m, h, t;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(h) : "v"(m), "v"(c));
                asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(h) : "v"(d));
                asm volatile("v_and_b32 %0, 0x30003, %1" : "=v"(t) : "v"(h));
                asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(f[k]) : "v"(t));
                asm volatile("v_and_b32 %0, 0xfffcfffc, %1" : "=v"(d) : "v"(h));
                asm volatile("v_or_b32 %0, 0x20002, %1" : "=v"(a[k]) : "v"(d));
            }
            if (KIND == K_CELL_PK3) {
                // packed cells, 4-bit tag field shared by two columns, three-way half-precision maximum:
                // 11 instructions per four cells
                unsigned m, ha, hb, t, la;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(ha) : "v"(m), "v"(c), "v"(d));
                asm volatile("v_and_b32 %0, 0xfff0fff0, %1" : "=v"(la) : "v"(ha));
                asm volatile("v_or_b32 %0, 0x80008, %1" : "=v"(a[k]) : "v"(la));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(g[k]), "v"(b));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(hb) : "v"(m), "v"(c), "v"(la));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(g[k]) : "v"(hb), "s"(0xfff0fff0u), "v"(b));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(t) : "s"(0xc000cu), "v"(ha), "v"(hb));
                asm volatile("v_and_b32 %0, 0xf000f, %1" : "=v"(t) : "v"(t));
                asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(f[k]) : "v"(t));
                d = hb;
            }
            if (KIND == K_SCORE_CHAIN) {
                // values-only packed cells: add (independent), three-way maximum whose left operand is the column before
                unsigned m;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_pk_maximum3_f16 %0, %1, %0, %2" : "+v"(a[k]) : "v"(m), "v"(d));
                d = a[k];
            }
            if (KIND == K_CELL_SIGN) {
                // add, max3, (sub, alignbit) x2: direction bits = sign bits of (m - h) and (up - h)
                unsigned m, h, x, y;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(h) : "v"(m), "v"(c), "v"(d));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(m), "v"(h));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(f[k]) : "v"(x));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(y) : "v"(c), "v"(h));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(g[k]) : "v"(y));
                a[k] = h;
            }
            if (KIND == K_CELL_TAG) {
                // values scaled by 4 with a 2-bit origin tag: add, max3, alignbit (2 tag bits), and, or
                unsigned m, h, z;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(h) : "v"(m), "v"(c), "v"(d));
                asm volatile("v_alignbit_b32 %0, %1, %0, 2" : "+v"(f[k]) : "v"(h));
                asm volatile("v_and_b32 %0, -4, %1" : "=v"(z) : "v"(h));
                asm volatile("v_or_b32 %0, 1, %1" : "=v"(a[k]) : "v"(z));
                d = z;
            }
            if (KIND == K_CELL_D) {
                // 2-bit code per cell by arithmetic: add, max, max, sub, sub, add(+1), min3, lshl_or
                unsigned m, h, x, y, cde;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(m) : "v"(a[k]), "v"(b));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(h) : "v"(m), "v"(c));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(h) : "v"(d));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(h), "v"(m));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(y) : "v"(h), "v"(c));
                asm volatile("v_min_u32 %0, %0, %1" : "+v"(y) : "v"(1u));
                asm volatile("v_min3_u32 %0, %1, %2, 2" : "=v"(cde) : "v"(x), "v"(y));
                asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(f[k]) : "v"(cde));
                a[k] = h;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int k = 0; k < CHAINS; k++) s ^= a[k] ^ f[k] ^ g[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int instr_per_step, unsigned* d_out, unsigned long long* d_cyc, int ncu) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = ncu * wps;             // 256 threads = 4 waves = one wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_valu<KIND><<<blocks, 256>>>(d_out, d_cyc, 1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 3; r++) k_valu<KIND><<<blocks, 256>>>(d_out, d_cyc, r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(blocks * 4);
        CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        double med = (double)cyc[cyc.size() / 2];
        double n_instr = (double)ITERS * CHAINS * instr_per_step;
        // cycles each SIMD spends per wave-instruction = wave lifetime / instructions / waves sharing the SIMD
        double cyc_per_instr = med / n_instr / wps;
        double lane_ops = 3.0 * blocks * 256.0 * n_instr;
        double wave_us = ms * 1e3 / 3.0;
        printf("%-30s w/SIMD=%d  %6.2f cyc/instr/SIMD  %6.2f T lane-instr/s  clk~%.2f GHz\n", name, wps,
               cyc_per_instr, lane_ops / (ms * 1e-3) / 1e12, med / wave_us / 1e3);
    }
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    printf("device %s CUs=%d clock=%.3f GHz\n", p.name, ncu, p.clockRate / 1e6);
    unsigned* d_out; CHECK(hipMalloc(&d_out, sizeof(unsigned) * ncu * 4 * 256));
    unsigned long long* d_cyc; CHECK(hipMalloc(&d_cyc, 8 * ncu * 4 * 4));
    if (getenv("VALU_PEAK_WALK")) {
        run<K_LSHR64>("v_lshrrev_b64", 1, d_out, d_cyc, ncu);
        run<K_FFBL>("v_ffbl_b32", 1, d_out, d_cyc, ncu);
        run<K_ADDCLAMP>("v_add_u32 clamp", 1, d_out, d_cyc, ncu);
        run<K_OR>("v_or_b32", 1, d_out, d_cyc, ncu);
        run<K_XOR>("v_xor_b32", 1, d_out, d_cyc, ncu);
        run<K_LSHL>("v_lshlrev_b32", 1, d_out, d_cyc, ncu);
        run<K_LSHRV>("v_lshrrev_b32 (register amount)", 1, d_out, d_cyc, ncu);
        run<K_NOT>("v_not_b32", 1, d_out, d_cyc, ncu);
        run<K_BCNT>("v_bcnt_u32_b32", 1, d_out, d_cyc, ncu);
        run<K_SUBLIT>("v_subrev_u32 literal", 1, d_out, d_cyc, ncu);
        run<K_ANDLIT>("v_and_b32 literal", 1, d_out, d_cyc, ncu);
        run<K_CMPSDWA>("v_cmp_eq_u32_sdwa", 1, d_out, d_cyc, ncu);
        run<K_MIN3>("v_min3_u32", 1, d_out, d_cyc, ncu);
        run<K_MED3>("v_med3_i32", 1, d_out, d_cyc, ncu);
        return 0;
    }
    if (getenv("VALU_PEAK_SCORE")) {
        run<K_SCORE_CHAIN>("2 cells: add, pk_max3 chained through the columns (2)", 2, d_out, d_cyc, ncu);
        run<K_PKMAX3F>("v_pk_maximum3_f16 (8 chains)", 1, d_out, d_cyc, ncu);
        run<K_CELL_PK3>("4 cells: 2add,2pk_max3,and,or,and_or,bfi,and,lshl_or (10)", 10, d_out, d_cyc, ncu);
        return 0;
    }
    run<K_CELL_SIGN>("cell: add,max3,2x(sub,alignbit) (6)", 6, d_out, d_cyc, ncu);
    run<K_CELL_TAG>("cell: add,max3,alignbit,and,or (5)", 5, d_out, d_cyc, ncu);
    run<K_ALIGNBIT>("v_alignbit_b32", 1, d_out, d_cyc, ncu);
    run<K_ANDOR>("v_and_or_b32", 1, d_out, d_cyc, ncu);
    run<K_PKMAX3F>("v_pk_maximum3_f16", 1, d_out, d_cyc, ncu);
    run<K_BFI>("v_bfi_b32", 1, d_out, d_cyc, ncu);
    run<K_CELL_PK>("2 cells: add,2pk_max,and,lshl_or,and,or (7)", 7, d_out, d_cyc, ncu);
    run<K_CELL_PK3>("4 cells: 2add,2pk_max3,and,or,and_or,bfi,and,lshl_or (10)", 10, d_out, d_cyc, ncu);
    if (getenv("VALU_PEAK_SHORT")) return 0;
    run<K_ADD>("v_add_u32", 1, d_out, d_cyc, ncu);
    run<K_SUB>("v_sub_u32", 1, d_out, d_cyc, ncu);
    run<K_MAX>("v_max_i32", 1, d_out, d_cyc, ncu);
    run<K_MIN>("v_min_u32", 1, d_out, d_cyc, ncu);
    run<K_AND>("v_and_b32", 1, d_out, d_cyc, ncu);
    run<K_LSHR>("v_lshrrev_b32", 1, d_out, d_cyc, ncu);
    run<K_MOV>("v_mov_b32", 1, d_out, d_cyc, ncu);
    run<K_MAX3>("v_max3_i32", 1, d_out, d_cyc, ncu);
    run<K_LSHLOR>("v_lshl_or_b32", 1, d_out, d_cyc, ncu);
    run<K_BFE>("v_bfe_u32", 1, d_out, d_cyc, ncu);
    run<K_PERM>("v_perm_b32", 1, d_out, d_cyc, ncu);
    run<K_MADU24>("v_mad_u32_u24", 1, d_out, d_cyc, ncu);
    run<K_SDWA>("v_add_u32_sdwa", 1, d_out, d_cyc, ncu);
    run<K_CMP>("v_cmp_ne_u32 vcc", 1, d_out, d_cyc, ncu);
    run<K_ADDC>("v_addc_co_u32 vcc", 1, d_out, d_cyc, ncu);
    run<K_CMPADDC_VCC>("cmp+addc via vcc (2)", 2, d_out, d_cyc, ncu);
    run<K_CMPADDC_SGPR>("cmp+addc via sgpr pair (2)", 2, d_out, d_cyc, ncu);
    run<K_PKADD>("v_pk_add_i16", 1, d_out, d_cyc, ncu);
    run<K_PKSUB>("v_pk_sub_i16", 1, d_out, d_cyc, ncu);
    run<K_PKMAX>("v_pk_max_i16", 1, d_out, d_cyc, ncu);
    run<K_PKMIN>("v_pk_min_u16", 1, d_out, d_cyc, ncu);
    run<K_PKMAD>("v_pk_mad_u16", 1, d_out, d_cyc, ncu);
    run<K_CELL6>("cell: add,max3,2x(cmp,addc) (6)", 6, d_out, d_cyc, ncu);
    run<K_CELL_C>("cell: add,2max,2cmp,2addc (7)", 7, d_out, d_cyc, ncu);
    run<K_CELL_B>("cell: add,2max,2x(sub,min,lshlor) (9)", 9, d_out, d_cyc, ncu);
    run<K_CELL_D>("cell: add,2max,2sub,min,min3,lshlor (8)", 8, d_out, d_cyc, ncu);
    return 0;
}
