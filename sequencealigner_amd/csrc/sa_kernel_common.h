// sa_kernel_common.h -- device code shared by the peptide kernel (sa_kernels.hip)
// and the long-sequence kernel (sa_kernels_long.hip): the DP cell, one row of a
// 64-column strip in both operand modes, and small wave helpers.
#ifndef SA_KERNEL_COMMON_H
#define SA_KERNEL_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sa_device.h"

#define SA_MODE_TRI 0
#define SA_MODE_ADJ 1

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
// Maximum of a value that is >= 0 in every lane, as a wave-uniform number, all 64 lanes active.  A max-scan along the
// rows of 16 lanes by DPP shifts, then lane 15 of rows 0 and 2 handed to rows 1 and 3 and lane 31 to the upper half:
// lane 63 ends with the maximum.  Lanes without a source take 0.  Ten vector instructions and no LDS traffic against
// six ds_bpermute exchanges with their address arithmetic.
__device__ __forceinline__ int wave_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- tagged cells --------------------------------------------------------------
// Values are kept as 4*D plus a 2-bit tag, and the tag of the winning operand
// of the max3 tells the traceback where the cell came from:
//   diagonal operand  4*(Dd + S') + 3     (bank value 4*Dd + 1 plus table entry 4*S' + 2)
//   upper operand     4*Du + 1            (bank value as it is)
//   left operand      4*Dl + 0            (tag cleared)
// On equal D the larger tag wins, which is the reference's preference DIAG, UP,
// LEFT (include/seqalign.h:73-105).  Bit 0 of a tag says "not LEFT", bit 1 "DIAG",
// so a count-leading-zeros over a row's tags gives the length of a run of LEFT
// moves (clz / 2) and what ends it (clz odd: UP, even: DIAG).
// Per cell: add, max3, alignbit (the two tag bits go into the row's direction
// word), and (left operand of the next column), or (bank value for the next row)
// -- five instructions, two of them (max3, alignbit) on the slow issue path.
#define SA_TAG_LEFT 0
#define SA_TAG_UP 1
#define SA_TAG_DIAG 3
#define SA_TAG_TABLE(sprime) (4 * (sprime) + 2)     /* table entry for a substitution score S' */

#define SA_CELL_TAG(IN, OUT, C, T, ACC)                                          \
    {                                                                            \
        const int diag_ = ((C) == 0) ? diag0 : IN[((C) == 0) ? 0 : (C) - 1];     \
        const int m_ = diag_ + (T);                                              \
        const int h_ = max3i(m_, IN[C], zl);                                     \
        ACC = __builtin_amdgcn_alignbit((uint32_t)h_, ACC, 2);                   \
        zl = h_ & ~3;                                                            \
        OUT[C] = zl | 1;                                                         \
    }

// Four columns of a TRIANGULAR row.  CUR holds their table entries; the entries of
// the next four columns are requested first so that the LDS latency is covered by
// the arithmetic of this group (a row of the profile is 68 dwords apart from the
// next, so the request after the last group stays inside the row's padding).
// Rows end early: gend = ceil(len1 / 4).
#define SA_GROUP_TAG(G, CUR, NXT, ACC)                                             \
    NXT = *reinterpret_cast<const int4*>(row_base + 4 * ((G) + 1));                \
    __builtin_amdgcn_sched_barrier(0);                                             \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 0, CUR.x, ACC)                                  \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 1, CUR.y, ACC)                                  \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 2, CUR.z, ACC)                                  \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 3, CUR.w, ACC)                                  \
    if (gend == (G) + 1) return;

// One row of tagged cells, TRIANGULAR operands: IN = previous row, OUT = this row,
// columns 0 .. 4*gend-1 (gend is wave-uniform).  sc_in = table entries of columns
// 0..3 for this row; sc_out receives the same for the next row (requested at the top
// of the row, used one row later).  IN and OUT hold 4*D + 1; diag0 is the border
// value in that form (1 on the matrix border), left0 the border value with the tag
// cleared (0).  e[q] receives the tags of columns 16q .. 16q+15, newest on
// top: after k cells column 16q + i sits at bits 32 - 2(k - i).
__device__ __forceinline__ void fill_row_tri_tag(const int (&IN)[SA_LP], int (&OUT)[SA_LP], const int gend,
                                                 const int diag0, const int left0,
                                                 const int32_t* __restrict__ row_base,
                                                 const int32_t* __restrict__ next_row_base, const int4& sc_in,
                                                 int4& sc_out, uint32_t& e0, uint32_t& e1, uint32_t& e2,
                                                 uint32_t& e3) {
    int4 scE = sc_in, scO;
    int zl = left0;
    sc_out = *reinterpret_cast<const int4*>(next_row_base);
    SA_GROUP_TAG(0, scE, scO, e0)
    SA_GROUP_TAG(1, scO, scE, e0)
    SA_GROUP_TAG(2, scE, scO, e0)
    SA_GROUP_TAG(3, scO, scE, e0)
    SA_GROUP_TAG(4, scE, scO, e1)
    SA_GROUP_TAG(5, scO, scE, e1)
    SA_GROUP_TAG(6, scE, scO, e1)
    SA_GROUP_TAG(7, scO, scE, e1)
    SA_GROUP_TAG(8, scE, scO, e2)
    SA_GROUP_TAG(9, scO, scE, e2)
    SA_GROUP_TAG(10, scE, scO, e2)
    SA_GROUP_TAG(11, scO, scE, e2)
    SA_GROUP_TAG(12, scE, scO, e3)
    SA_GROUP_TAG(13, scO, scE, e3)
    SA_GROUP_TAG(14, scE, scO, e3)
    SA_GROUP_TAG(15, scO, scE, e3)
}

// The same row with the table entries requested two groups ahead (three rotating buffers):
// for kernels that run two waves per SIMD, where one group of arithmetic does not cover
// the LDS latency.
#define SA_GROUP_TAG2(G, CUR, FAR, ACC)                                            \
    FAR = *reinterpret_cast<const int4*>(row_base + 4 * ((G) + 2));                \
    __builtin_amdgcn_sched_barrier(0);                                             \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 0, CUR.x, ACC)                                  \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 1, CUR.y, ACC)                                  \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 2, CUR.z, ACC)                                  \
    SA_CELL_TAG(IN, OUT, 4 * (G) + 3, CUR.w, ACC)                                  \
    if (gend == (G) + 1) return;

__device__ __forceinline__ void fill_row_tri_tag2(const int (&IN)[SA_LP], int (&OUT)[SA_LP], const int gend,
                                                  const int diag0, const int left0,
                                                  const int32_t* __restrict__ row_base,
                                                  const int32_t* __restrict__ next_row_base, const int4& sc_in,
                                                  int4& sc_out, uint32_t& e0, uint32_t& e1, uint32_t& e2,
                                                  uint32_t& e3) {
    int4 sA = sc_in, sB, sC;
    int zl = left0;
    sB = *reinterpret_cast<const int4*>(row_base + 4);
    sc_out = *reinterpret_cast<const int4*>(next_row_base);
    SA_GROUP_TAG2(0, sA, sC, e0)
    SA_GROUP_TAG2(1, sB, sA, e0)
    SA_GROUP_TAG2(2, sC, sB, e0)
    SA_GROUP_TAG2(3, sA, sC, e0)
    SA_GROUP_TAG2(4, sB, sA, e1)
    SA_GROUP_TAG2(5, sC, sB, e1)
    SA_GROUP_TAG2(6, sA, sC, e1)
    SA_GROUP_TAG2(7, sB, sA, e1)
    SA_GROUP_TAG2(8, sC, sB, e2)
    SA_GROUP_TAG2(9, sA, sC, e2)
    SA_GROUP_TAG2(10, sB, sA, e2)
    SA_GROUP_TAG2(11, sC, sB, e2)
    SA_GROUP_TAG2(12, sA, sC, e3)
    SA_GROUP_TAG2(13, sB, sA, e3)
    SA_GROUP_TAG2(14, sC, sB, e3)
    SA_GROUP_TAG2(15, sA, sC, e3)
}

// The same row with ADJACENT operands: table entries SA_TAG_TABLE(S') from an LDS copy, row
// selected by the row residue, entry by the column residue (a4 = 4 * residue code).
__device__ __forceinline__ void fill_row_adj_tag(const int (&IN)[SA_LP], int (&OUT)[SA_LP], const int gend,
                                                 const int diag0, const int left0,
                                                 const int32_t* __restrict__ row_base, const int (&a4)[SA_LP],
                                                 uint32_t& e0, uint32_t& e1, uint32_t& e2, uint32_t& e3) {
    const char* rb = reinterpret_cast<const char*>(row_base);
    int zl = left0;
#pragma unroll
    for (int g = 0; g < SA_LP / 4; g++) {
        if (g < gend) {   // wave-uniform
            const int s0 = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 0]);
            const int s1 = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 1]);
            const int s2 = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 2]);
            const int s3 = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 3]);
            uint32_t& acc = g < 4 ? e0 : g < 8 ? e1 : g < 12 ? e2 : e3;
            SA_CELL_TAG(IN, OUT, 4 * g + 0, s0, acc)
            SA_CELL_TAG(IN, OUT, 4 * g + 1, s1, acc)
            SA_CELL_TAG(IN, OUT, 4 * g + 2, s2, acc)
            SA_CELL_TAG(IN, OUT, 4 * g + 3, s3, acc)
        }
    }
}

// value of register column `col` (wave-uniform) of a row bank
#define SA_PICK_CASE(C) case C: v = BANK[C]; break;
__device__ __forceinline__ int pick_column(const int (&BANK)[SA_LP], const int col) {
    int v = 0;
    switch (col) {
        SA_PICK_CASE(0) SA_PICK_CASE(1) SA_PICK_CASE(2) SA_PICK_CASE(3) SA_PICK_CASE(4) SA_PICK_CASE(5) SA_PICK_CASE(6) SA_PICK_CASE(7)
        SA_PICK_CASE(8) SA_PICK_CASE(9) SA_PICK_CASE(10) SA_PICK_CASE(11) SA_PICK_CASE(12) SA_PICK_CASE(13) SA_PICK_CASE(14) SA_PICK_CASE(15)
        SA_PICK_CASE(16) SA_PICK_CASE(17) SA_PICK_CASE(18) SA_PICK_CASE(19) SA_PICK_CASE(20) SA_PICK_CASE(21) SA_PICK_CASE(22) SA_PICK_CASE(23)
        SA_PICK_CASE(24) SA_PICK_CASE(25) SA_PICK_CASE(26) SA_PICK_CASE(27) SA_PICK_CASE(28) SA_PICK_CASE(29) SA_PICK_CASE(30) SA_PICK_CASE(31)
        SA_PICK_CASE(32) SA_PICK_CASE(33) SA_PICK_CASE(34) SA_PICK_CASE(35) SA_PICK_CASE(36) SA_PICK_CASE(37) SA_PICK_CASE(38) SA_PICK_CASE(39)
        SA_PICK_CASE(40) SA_PICK_CASE(41) SA_PICK_CASE(42) SA_PICK_CASE(43) SA_PICK_CASE(44) SA_PICK_CASE(45) SA_PICK_CASE(46) SA_PICK_CASE(47)
        SA_PICK_CASE(48) SA_PICK_CASE(49) SA_PICK_CASE(50) SA_PICK_CASE(51) SA_PICK_CASE(52) SA_PICK_CASE(53) SA_PICK_CASE(54) SA_PICK_CASE(55)
        SA_PICK_CASE(56) SA_PICK_CASE(57) SA_PICK_CASE(58) SA_PICK_CASE(59) SA_PICK_CASE(60) SA_PICK_CASE(61) SA_PICK_CASE(62) SA_PICK_CASE(63)
        default: break;
    }
    return v;
}

#endif
