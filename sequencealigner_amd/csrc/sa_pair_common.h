// sa_pair_common.h -- packed cells: two DP matrices per lane, 16 bits each, shared by the
// paired kernels (sa_kernels_pair.hip, sa_kernels_long_pair.hip).
//
// Every VGPR of the DP row holds the cell of (p1, q) in its low 16 bits and the cell of (p2, q)
// in its high 16 bits.  Tags: 0 LEFT, 2 UP, 3 DIAG: with the newest column in the low bits of a
// direction word a count-trailing-zeros finds a run of LEFT moves (sa_traceback.h, REV = 1).
//
// Cell format: 16*D + two tag fields.  Per four cells (two columns of both pairs): 2 add, 2 three-way
// maxima, and, or, and-or for the operands of the cells to come, 2.5 on average for the tags.
#ifndef SA_PAIR_COMMON_H
#define SA_PAIR_COMMON_H

#include "sa_kernel_common.h"

// Bits 3:2 of a half take the tag of a cell in an even column, bits 1:0 that of a cell in an odd column,
// so one and + one shift-or move the tags of two columns (times two pairs) into the direction word.
// The maximum of the three candidates is one instruction, v_pk_maximum3_f16: bit patterns below 0x7c00
// are the finite non-negative halves (subnormals included, the kernels run with them enabled), which
// order exactly as the unsigned integers do, and the maximum returns the winning operand unchanged
// (sa_pk_max3_check.hip compares it with the integer maximum over 10^9 triples).  Hence the range:
// 16 * 64 * max S' + 15 < 0x7c00, i.e. S' <= 30.
//  even column a: candidates  diag = bank[a-1] (16*D + 2) + (16*S' + 10) -> field 3:2 = DIAG, field 1:0 = 0
//                             up   = bank[a]   (16*D + 8)
//                             left = the raw result of column a-1 (field 3:2 = 0, its own tag in 1:0:
//                                    the low field never decides between candidates whose high fields differ)
//  odd column b:  candidates  diag = bank[b-1] (16*D + 8) + (16*S' - 5) = 16*(D + S') + 3: the carry out of
//                                    the tag fields is taken back by the table entry, hence S' >= 1
//                             up   = bank[b]   (16*D + 2)
//                             left = the result of column a with both fields cleared
// The steps are written as instructions so that they stay where they are: moved to the end of the row by the
// compiler they keep one more register per column alive, and the row no longer fits the register file.
#define SA_NB_UP_A 0x00080008u                       /* bank form of D = 0, even column */
#define SA_NB_UP_B 0x00020002u                       /* bank form of D = 0, odd column; also diag0 of a border */
#define SA_NB_TABLE_A(sprime) (16 * (sprime) + 10)   /* table entries for a substitution score S' >= 1 */
#define SA_NB_TABLE_B(sprime) (16 * (sprime) - 5)
#define SA_NB_SHIFT 4

// Columns C (even) and C + 1 of both pairs as one block of instructions.  The tags of the two columns go to
// bits 4(3-i) .. 4(3-i)+3 of each half of the direction word, i = (C mod 8) / 2: a word that takes fewer than
// eight columns is filled from the top.  The first deposit of a word (a packed 16-bit shift, which also
// drops the D bits that came along from the odd column) overwrites the word.
#define SA_NB_HEAD                                                  \
    "v_pk_maximum3_f16 %[ma], %[ma], %[upa], %[zl]\n\t"             \
    "v_and_b32 %[la], 0xfff0fff0, %[ma]\n\t"                        \
    "v_or_b32 %[outa], 0x80008, %[la]\n\t"                          \
    "v_pk_maximum3_f16 %[mb], %[mb], %[upb], %[la]\n\t"             \
    "v_bfi_b32 %[t], %[fielda], %[ma], %[mb]\n\t"
#define SA_NB_TAIL "v_and_or_b32 %[outb], %[mb], %[clean], %[two]"
#define SA_NB_OPERANDS(ACC_CONSTRAINT, IN, OUT, C, ACC)                                                                      \
    : [ma] "+v"(ma_), [mb] "+v"(mb_), [la] "=&v"(la_), [t] "=&v"(t_), [outa] "=&v"(OUT[C]),                  \
      [outb] "=&v"(OUT[(C) + 1]), [acc] ACC_CONSTRAINT(ACC)                                                   \
    : [upa] "v"(IN[C]), [upb] "v"(IN[(C) + 1]), [zl] "v"(zl), [fielda] "s"(0x000c000cu), [clean] "s"(0xfff0fff0u), \
      [two] "v"(upb), [low] "s"(0x000f000fu)
#define SA_CELL_NB(IN, OUT, C, TA, TB, ACC)                                                     \
    {                                                                                           \
        uint32_t ma_ = (((C) == 0) ? diag0 : IN[((C) == 0) ? 0 : (C) - 1]) + (uint32_t)(TA);    \
        uint32_t mb_ = IN[C] + (uint32_t)(TB);                                                  \
        uint32_t la_, t_;                                                                       \
        if (((C) & 7) == 0) {                                                                   \
            asm volatile(SA_NB_HEAD "v_pk_lshlrev_b16 %[acc], 12, %[t] op_sel_hi:[0,1]\n\t" SA_NB_TAIL SA_NB_OPERANDS("=&v", IN, OUT, C, ACC)); \
        } else if (((C) & 7) == 2) {                                                            \
            asm volatile(SA_NB_HEAD "v_and_b32 %[t], %[low], %[t]\n\tv_lshl_or_b32 %[acc], %[t], 8, %[acc]\n\t" SA_NB_TAIL SA_NB_OPERANDS("+v", IN, OUT, C, ACC)); \
        } else if (((C) & 7) == 4) {                                                            \
            asm volatile(SA_NB_HEAD "v_and_b32 %[t], %[low], %[t]\n\tv_lshl_or_b32 %[acc], %[t], 4, %[acc]\n\t" SA_NB_TAIL SA_NB_OPERANDS("+v", IN, OUT, C, ACC)); \
        } else {                                                                                \
            asm volatile(SA_NB_HEAD "v_and_or_b32 %[acc], %[t], %[low], %[acc]\n\t" SA_NB_TAIL SA_NB_OPERANDS("+v", IN, OUT, C, ACC)); \
        }                                                                                       \
        zl = mb_;                                                                               \
    }

// Four columns.  Rows end early: gend = ceil(longer first sequence / 4).
// AHEAD = false: every group reads its own table entries and waits for them (a request the source places a
// group ahead is moved by the compiler behind the row's end test, to where its value is used).  With three waves
// per SIMD the other waves cover that wait, and hand-placed requests gained nothing.
// AHEAD = true: the entries of group G + 1 are requested, as an instruction of its own, before the cells of
// group G, and waited for by count: requests to the LDS complete in the order they were issued, so "at most
// one outstanding" means that all but the newest have arrived.  The request for the group behind the row's end
// stays inside the profile row and is never used.  For kernels that run two waves per SIMD.
typedef int sa_i32x4 __attribute__((ext_vector_type(4)));
#define SA_LDS_REQUEST(DST, ADDR, BYTES) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(BYTES))
#define SA_GROUP_PK(G, CUR, NXT)                                                   \
    if (AHEAD) {                                                                   \
        if ((G) + 1 < NG) {                                                        \
            SA_LDS_REQUEST(NXT, row_addr, 16 * ((G) + 1));                         \
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(CUR));                      \
        } else {                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR));                      \
        }                                                                          \
    } else {                                                                       \
        CUR = *reinterpret_cast<const sa_i32x4*>(row_base + 4 * (G));              \
    }                                                                              \
    SA_CELL_NB(IN, OUT, 4 * (G) + 0, CUR.x, CUR.y, e[(G) >> 1])                    \
    SA_CELL_NB(IN, OUT, 4 * (G) + 2, CUR.z, CUR.w, e[(G) >> 1])                    \
    if (NG == (G) + 1 || gend == (G) + 1) return;

// One packed DP row of 16*D cells: IN = previous row, OUT = this row, columns 0 .. 4*gend-1.  diag0 = border
// value of the previous row in odd-column bank form (SA_NB_UP_B on the matrix border), left0 = this row's
// border value with the tags cleared (0 on the matrix border).
// e[w] receives the tags of columns 8w .. 8w+7 of both pairs: column 8w+i at bits 2(7-i) of each 16-bit
// half, also when the row ends after four columns of the word.
template <int NG, bool AHEAD = false>
__device__ __forceinline__ void fill_row_pair(const uint32_t (&IN)[4 * NG], uint32_t (&OUT)[4 * NG], const int gend,
                                              const uint32_t diag0, const uint32_t left0,
                                              const int32_t* __restrict__ row_base, uint32_t (&e)[(NG + 1) / 2]) {
    uint32_t zl = left0;
    uint32_t upb = SA_NB_UP_B;
    asm volatile("" : "+v"(upb));          // one register for the whole row, not a copy per use
    sa_i32x4 scE, scO;
    const uint32_t row_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int32_t*)row_base;
    if (AHEAD) SA_LDS_REQUEST(scE, row_addr, 0);
    SA_GROUP_PK(0, scE, scO) SA_GROUP_PK(1, scO, scE) SA_GROUP_PK(2, scE, scO) SA_GROUP_PK(3, scO, scE)
    SA_GROUP_PK(4, scE, scO) SA_GROUP_PK(5, scO, scE) SA_GROUP_PK(6, scE, scO) SA_GROUP_PK(7, scO, scE)
    SA_GROUP_PK(8, scE, scO) SA_GROUP_PK(9, scO, scE) SA_GROUP_PK(10, scE, scO) SA_GROUP_PK(11, scO, scE)
    SA_GROUP_PK(12, scE, scO)
    if constexpr (NG > 13) { SA_GROUP_PK(13, scO, scE) SA_GROUP_PK(14, scE, scO) SA_GROUP_PK(15, scO, scE) }
}


// value of register column `col` (wave-uniform) of a packed row bank
#define SA_PICK_PK(C) case C: v = BANK[(C) < NC ? (C) : 0]; break;
template <int NC>
__device__ __forceinline__ uint32_t pick_packed(const uint32_t (&BANK)[NC], const int col) {
    uint32_t v = 0;
    switch (col) {
        SA_PICK_PK(0) SA_PICK_PK(1) SA_PICK_PK(2) SA_PICK_PK(3) SA_PICK_PK(4) SA_PICK_PK(5) SA_PICK_PK(6) SA_PICK_PK(7)
        SA_PICK_PK(8) SA_PICK_PK(9) SA_PICK_PK(10) SA_PICK_PK(11) SA_PICK_PK(12) SA_PICK_PK(13) SA_PICK_PK(14) SA_PICK_PK(15)
        SA_PICK_PK(16) SA_PICK_PK(17) SA_PICK_PK(18) SA_PICK_PK(19) SA_PICK_PK(20) SA_PICK_PK(21) SA_PICK_PK(22) SA_PICK_PK(23)
        SA_PICK_PK(24) SA_PICK_PK(25) SA_PICK_PK(26) SA_PICK_PK(27) SA_PICK_PK(28) SA_PICK_PK(29) SA_PICK_PK(30) SA_PICK_PK(31)
        SA_PICK_PK(32) SA_PICK_PK(33) SA_PICK_PK(34) SA_PICK_PK(35) SA_PICK_PK(36) SA_PICK_PK(37) SA_PICK_PK(38) SA_PICK_PK(39)
        SA_PICK_PK(40) SA_PICK_PK(41) SA_PICK_PK(42) SA_PICK_PK(43) SA_PICK_PK(44) SA_PICK_PK(45) SA_PICK_PK(46) SA_PICK_PK(47)
        SA_PICK_PK(48) SA_PICK_PK(49) SA_PICK_PK(50) SA_PICK_PK(51) SA_PICK_PK(52) SA_PICK_PK(53) SA_PICK_PK(54) SA_PICK_PK(55)
        SA_PICK_PK(56) SA_PICK_PK(57) SA_PICK_PK(58) SA_PICK_PK(59) SA_PICK_PK(60) SA_PICK_PK(61) SA_PICK_PK(62) SA_PICK_PK(63)
        default: break;
    }
    return v;
}

#endif
