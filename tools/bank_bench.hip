// bank_bench.hip -- does the position of a three-way maximum's operands in the vector register file matter?
// One values-only DP row of 64 columns in fixed registers v64..v127, temporaries in chosen registers, four waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bank_bench.hip -o tools/bank_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ROWS = 20000;

// column c: m = R[c-1] + t (t: a register standing for the table entry), R[c] = max3(m_prev, R[c], R[c-1])
// TEMPS: the four registers the diagonal operands rotate through, TAB: the four "table entry" registers
template <int VARIANT>
__global__ void __launch_bounds__(64, 4) k_bank(unsigned* out, unsigned seed) {
    unsigned acc = 0;
    // registers v64..v127 hold the row; everything is written as one block per row so that the numbers are ours
    asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0" :: "v"(seed + threadIdx.x) : "v2", "v3", "v4", "v5");
#define INIT(c) asm volatile("v_mov_b32 v" #c ", %0" :: "v"(seed + c) : "v" #c);
    INIT(64) INIT(65) INIT(66) INIT(67) INIT(68) INIT(69) INIT(70) INIT(71) INIT(72) INIT(73) INIT(74) INIT(75) INIT(76) INIT(77) INIT(78) INIT(79)
    INIT(80) INIT(81) INIT(82) INIT(83) INIT(84) INIT(85) INIT(86) INIT(87) INIT(88) INIT(89) INIT(90) INIT(91) INIT(92) INIT(93) INIT(94) INIT(95)
    INIT(96) INIT(97) INIT(98) INIT(99) INIT(100) INIT(101) INIT(102) INIT(103) INIT(104) INIT(105) INIT(106) INIT(107) INIT(108) INIT(109) INIT(110) INIT(111)
    INIT(112) INIT(113) INIT(114) INIT(115) INIT(116) INIT(117) INIT(118) INIT(119) INIT(120) INIT(121) INIT(122) INIT(123) INIT(124) INIT(125) INIT(126) INIT(127)
    for (int r = 0; r < ROWS; r++) {
        // one group of four columns C..C+3 (registers v(64+C)..): temporaries T0..T3, table registers S0..S3, left operand L
#define GROUP(B0, B1, B2, B3, L, T0, T1, T2, T3, MA)                                    \
        "v_add_u32 v" #T0 ", v" #B0 ", v2\n\t"                                           \
        "v_add_u32 v" #T1 ", v" #B1 ", v3\n\t"                                           \
        "v_pk_maximum3_f16 v" #B0 ", v" #MA ", v" #B0 ", v" #L "\n\t"                    \
        "v_add_u32 v" #T2 ", v" #B2 ", v4\n\t"                                           \
        "v_pk_maximum3_f16 v" #B1 ", v" #T0 ", v" #B1 ", v" #B0 "\n\t"                   \
        "v_add_u32 v" #MA ", v" #B3 ", v5\n\t"                                           \
        "v_pk_maximum3_f16 v" #B2 ", v" #T1 ", v" #B2 ", v" #B1 "\n\t"                   \
        "v_pk_maximum3_f16 v" #B3 ", v" #T2 ", v" #B3 ", v" #B2 "\n\t"
        if (VARIANT == 0) {
            // temporaries v8, v9, v10 (banks 0, 1, 2), next diagonal v11: as the allocator might pick them
            asm volatile(
                GROUP(64, 65, 66, 67, 127, 8, 9, 10, 0, 11) GROUP(68, 69, 70, 71, 67, 8, 9, 10, 0, 11) GROUP(72, 73, 74, 75, 71, 8, 9, 10, 0, 11) GROUP(76, 77, 78, 79, 75, 8, 9, 10, 0, 11)
                GROUP(80, 81, 82, 83, 79, 8, 9, 10, 0, 11) GROUP(84, 85, 86, 87, 83, 8, 9, 10, 0, 11) GROUP(88, 89, 90, 91, 87, 8, 9, 10, 0, 11) GROUP(92, 93, 94, 95, 91, 8, 9, 10, 0, 11)
                GROUP(96, 97, 98, 99, 95, 8, 9, 10, 0, 11) GROUP(100, 101, 102, 103, 99, 8, 9, 10, 0, 11) GROUP(104, 105, 106, 107, 103, 8, 9, 10, 0, 11) GROUP(108, 109, 110, 111, 107, 8, 9, 10, 0, 11)
                GROUP(112, 113, 114, 115, 111, 8, 9, 10, 0, 11) GROUP(116, 117, 118, 119, 115, 8, 9, 10, 0, 11) GROUP(120, 121, 122, 123, 119, 8, 9, 10, 0, 11) GROUP(124, 125, 126, 127, 123, 8, 9, 10, 0, 11)
                ::: "v8", "v9", "v10", "v11", "memory");
        } else if (VARIANT == 1) {
            // conflict-free if bank = register number mod 4: column c = 4G+k uses R[c] (bank k) and R[c-1] (bank k-1);
            // its diagonal operand goes to bank k+1 or k+2.  MA feeds column 0 (banks 0, 3): bank 1 -> v13; T0 feeds
            // column 1 (banks 1, 0): bank 2 -> v14; T1 feeds column 2 (banks 2, 1): bank 3 -> v15; T2 feeds column 3
            // (banks 3, 2): bank 0 -> v16
            asm volatile(
                GROUP(64, 65, 66, 67, 127, 14, 15, 16, 0, 13) GROUP(68, 69, 70, 71, 67, 14, 15, 16, 0, 13) GROUP(72, 73, 74, 75, 71, 14, 15, 16, 0, 13) GROUP(76, 77, 78, 79, 75, 14, 15, 16, 0, 13)
                GROUP(80, 81, 82, 83, 79, 14, 15, 16, 0, 13) GROUP(84, 85, 86, 87, 83, 14, 15, 16, 0, 13) GROUP(88, 89, 90, 91, 87, 14, 15, 16, 0, 13) GROUP(92, 93, 94, 95, 91, 14, 15, 16, 0, 13)
                GROUP(96, 97, 98, 99, 95, 14, 15, 16, 0, 13) GROUP(100, 101, 102, 103, 99, 14, 15, 16, 0, 13) GROUP(104, 105, 106, 107, 103, 14, 15, 16, 0, 13) GROUP(108, 109, 110, 111, 107, 14, 15, 16, 0, 13)
                GROUP(112, 113, 114, 115, 111, 14, 15, 16, 0, 13) GROUP(116, 117, 118, 119, 115, 14, 15, 16, 0, 13) GROUP(120, 121, 122, 123, 119, 14, 15, 16, 0, 13) GROUP(124, 125, 126, 127, 123, 14, 15, 16, 0, 13)
                ::: "v13", "v14", "v15", "v16", "memory");
        } else {
            // the worst case under the same assumption: every diagonal operand in the bank of its column's register
            // MA -> bank 0 (v12), T0 -> bank 1 (v17), T1 -> bank 2 (v18), T2 -> bank 3 (v19)
            asm volatile(
                GROUP(64, 65, 66, 67, 127, 17, 18, 19, 0, 12) GROUP(68, 69, 70, 71, 67, 17, 18, 19, 0, 12) GROUP(72, 73, 74, 75, 71, 17, 18, 19, 0, 12) GROUP(76, 77, 78, 79, 75, 17, 18, 19, 0, 12)
                GROUP(80, 81, 82, 83, 79, 17, 18, 19, 0, 12) GROUP(84, 85, 86, 87, 83, 17, 18, 19, 0, 12) GROUP(88, 89, 90, 91, 87, 17, 18, 19, 0, 12) GROUP(92, 93, 94, 95, 91, 17, 18, 19, 0, 12)
                GROUP(96, 97, 98, 99, 95, 17, 18, 19, 0, 12) GROUP(100, 101, 102, 103, 99, 17, 18, 19, 0, 12) GROUP(104, 105, 106, 107, 103, 17, 18, 19, 0, 12) GROUP(108, 109, 110, 111, 107, 17, 18, 19, 0, 12)
                GROUP(112, 113, 114, 115, 111, 17, 18, 19, 0, 12) GROUP(116, 117, 118, 119, 115, 17, 18, 19, 0, 12) GROUP(120, 121, 122, 123, 119, 17, 18, 19, 0, 12) GROUP(124, 125, 126, 127, 123, 17, 18, 19, 0, 12)
                ::: "v12", "v17", "v18", "v19", "memory");
        }
    }
    asm volatile("v_xor_b32 %0, v64, v127" : "=v"(acc));
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int V> void run(const char* name, unsigned* d_out, int ncu) {
    const int blocks = ncu * 16;
    k_bank<V><<<blocks, 64>>>(d_out, 1);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) k_bank<V><<<blocks, 64>>>(d_out, r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 3;
    printf("%-46s %.1f ns per row per SIMD, %.2f T lane-instr/s, %.1f TCUPS\n", name, us * 1e3 / ROWS / 4, 128.0 * 64 * ROWS * blocks / (us * 1e-6) / 1e12,
           8192.0 * ROWS * blocks / (us * 1e-6) / 1e12);
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    unsigned* d_out; CHECK(hipMalloc(&d_out, 4 * p.multiProcessorCount * 16 * 64));
    run<0>("temporaries in v8..v11", d_out, p.multiProcessorCount);
    run<1>("temporaries placed off the operands' banks", d_out, p.multiProcessorCount);
    run<2>("temporaries placed on the operands' banks", d_out, p.multiProcessorCount);
    run<0>("temporaries in v8..v11", d_out, p.multiProcessorCount);
    return 0;
}
