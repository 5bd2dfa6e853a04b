// sa_pair_common.h -- packed cells: two DP matrices per lane, 16 bits each, shared by the
// paired kernels (sa_kernels_pair.hip, sa_kernels_long_pair.hip).
//
// Every VGPR of the DP row holds the cell of (p1, q) in its low 16 bits and the cell of (p2, q)
// in its high 16 bits.  Tags: 0 LEFT, 2 UP, 3 DIAG: with the newest column in the low bits of a
// direction word a count-trailing-zeros finds a run of LEFT moves (sa_traceback.h, traceback_pk).
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

// Columns C (even) and C + 1 of both pairs as one block of instructions, in place: B[C] and B[C + 1] hold the bank
// values of the row above when the block starts and those of this row when it ends (one register row, not two:
// each is read as the UP operand before it is written, and the diagonal operands -- MA for the even column, MB
// for the odd one, table entries already added -- were taken from the row above before the block that
// overwrites their source).  MB leaves as the raw result of the odd column, the LEFT operand of the next block.
// The tags of the two columns go to bits 4(3-i) .. 4(3-i)+3 of each half of the direction word,
// i = (C mod 8) / 2: a word that takes fewer than eight columns is filled from the top.  The first deposit of a
// word (a packed 16-bit shift, which also drops the D bits that came along from the odd column) overwrites it.
#define SA_NB_HEAD                                                  \
    "v_pk_maximum3_f16 %[ma], %[ma], %[ba], %[zl]\n\t"              \
    "v_and_b32 %[la], 0xfff0fff0, %[ma]\n\t"                        \
    "v_or_b32 %[ba], 0x80008, %[la]\n\t"                            \
    "v_pk_maximum3_f16 %[mb], %[mb], %[bb], %[la]\n\t"              \
    "v_bfi_b32 %[t], %[fielda], %[ma], %[mb]\n\t"
// The same with the three-way maximum as two unsigned 16-bit maxima: one instruction more per column, and the whole
// 16-bit range instead of the patterns below 0x7c00 (tables with larger entries, see packed_cells_mode in sa_api.cpp).
#define SA_NB_HEAD_WIDE                                             \
    "v_pk_max_u16 %[ma], %[ma], %[ba]\n\t"                          \
    "v_pk_max_u16 %[ma], %[ma], %[zl]\n\t"                          \
    "v_and_b32 %[la], 0xfff0fff0, %[ma]\n\t"                        \
    "v_or_b32 %[ba], 0x80008, %[la]\n\t"                            \
    "v_pk_max_u16 %[mb], %[mb], %[bb]\n\t"                          \
    "v_pk_max_u16 %[mb], %[mb], %[la]\n\t"                          \
    "v_bfi_b32 %[t], %[fielda], %[ma], %[mb]\n\t"
#define SA_NB_TAIL "v_and_or_b32 %[bb], %[mb], %[clean], %[two]"
#define SA_NB_OPERANDS(ACC_CONSTRAINT, B, C, MA, MB, ACC)                                                    \
    : [ma] "+v"(MA), [mb] "+v"(MB), [ba] "+v"(B[C]), [bb] "+v"(B[(C) + 1]), [la] "=&v"(la_), [t] "=&v"(t_),    \
      [acc] ACC_CONSTRAINT(ACC)                                                                              \
    : [zl] "v"(zl), [fielda] "s"(0x000c000cu), [clean] "s"(0xfff0fff0u), [two] "v"(upb), [low] "s"(0x000f000fu)
#define SA_CELL_NB_FORM(HEAD, B, C, MA, MB, ACC)                                                \
    {                                                                                           \
        uint32_t la_, t_;                                                                       \
        if (((C) & 7) == 0) {                                                                   \
            asm volatile(HEAD "v_pk_lshlrev_b16 %[acc], 12, %[t] op_sel_hi:[0,1]\n\t" SA_NB_TAIL SA_NB_OPERANDS("=&v", B, C, MA, MB, ACC)); \
        } else if (((C) & 7) == 2) {                                                            \
            asm volatile(HEAD "v_and_b32 %[t], %[low], %[t]\n\tv_lshl_or_b32 %[acc], %[t], 8, %[acc]\n\t" SA_NB_TAIL SA_NB_OPERANDS("+v", B, C, MA, MB, ACC)); \
        } else if (((C) & 7) == 4) {                                                            \
            asm volatile(HEAD "v_and_b32 %[t], %[low], %[t]\n\tv_lshl_or_b32 %[acc], %[t], 4, %[acc]\n\t" SA_NB_TAIL SA_NB_OPERANDS("+v", B, C, MA, MB, ACC)); \
        } else {                                                                                \
            asm volatile(HEAD "v_and_or_b32 %[acc], %[t], %[low], %[acc]\n\t" SA_NB_TAIL SA_NB_OPERANDS("+v", B, C, MA, MB, ACC)); \
        }                                                                                       \
        zl = MB;                                                                                \
    }
#define SA_CELL_NB(B, C, MA, MB, ACC)                                                           \
    if constexpr (WIDE) SA_CELL_NB_FORM(SA_NB_HEAD_WIDE, B, C, MA, MB, ACC)                     \
    else SA_CELL_NB_FORM(SA_NB_HEAD, B, C, MA, MB, ACC)

// Four columns.  Rows end early: after gend = ceil(longer first sequence / 4) groups, a run-time value, when the
// template argument END2 is 0; otherwise after END2 = ceil(longer first sequence / 2) column pairs, a constant: no
// test per group then (the row is straight-line code), and a last group of one column pair is half a group.
// Profile rows are stored one column to the left: dword k of a row holds the table entry of column k + 1, the
// entry of column 0 stands in the row's last dword (PS - 1).  The four entries group G reads are then those of
// columns 4G+1 .. 4G+4, all it needs: the diagonal operands of its own columns 4G+1, 4G+2, 4G+3 and of the next
// group's first column, which must be taken from the row above before this group overwrites it.  That last add
// is written as an instruction so that it stays here: left to the compiler it moves into the next group (the
// only place its result is used) behind a copy of the register it reads.
typedef int sa_i32x4 __attribute__((ext_vector_type(4)));
typedef int sa_i32x2 __attribute__((ext_vector_type(2)));
// The table entries of group G + 1 are requested before group G is computed and awaited at the top of group G + 1
// (two register sets take turns), so that the LDS latency -- 8 % of the launch when every group waited for its own
// read -- is covered by a group's arithmetic.  Nothing is requested past the row's last group: a read still in
// flight when the row function returns would land in registers the compiler has given to something else.
// The requests and the wait are instructions the compiler cannot see through: the wait names the registers as its
// outputs, so every use of them stays behind it.
#define SA_PROF_REQUEST(LO, HI, G)                                                                                  \
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"                                       \
                 : "=&v"(LO), "=&v"(HI) : "v"(row_lds), "n"(16 * (G)), "n"(16 * (G) + 8))
#define SA_PROF_AWAIT(LO, HI) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(LO), "+v"(HI))
#define SA_GROUP_PK(G)                                                                          \
    {                                                                                           \
        /* END2 != 0: the row is END2 column pairs long, a constant; else gend groups, a run-time value */ \
        constexpr bool last_ = NG == (G) + 1 || (END2 != 0 && 2 * (G) + 2 >= END2);             \
        constexpr bool half_ = END2 != 0 && 2 * (G) + 1 == END2;                                \
        sa_i32x4 sc_;                                                                           \
        if (((G) & 1) == 0) {                                                                   \
            SA_PROF_AWAIT(pa_lo, pa_hi);                                                        \
            if (!last_ && (END2 != 0 || gend != (G) + 1)) SA_PROF_REQUEST(pb_lo, pb_hi, (G) + 1); \
            sc_.x = pa_lo.x; sc_.y = pa_lo.y; sc_.z = pa_hi.x; sc_.w = pa_hi.y;                 \
        } else {                                                                                \
            SA_PROF_AWAIT(pb_lo, pb_hi);                                                        \
            if (!last_ && (END2 != 0 || gend != (G) + 1)) SA_PROF_REQUEST(pa_lo, pa_hi, (G) + 1); \
            sc_.x = pb_lo.x; sc_.y = pb_lo.y; sc_.z = pb_hi.x; sc_.w = pb_hi.y;                 \
        }                                                                                       \
        uint32_t mb0_ = B[4 * (G)] + (uint32_t)sc_.x;                                           \
        uint32_t ma1_ = B[4 * (G) + 1] + (uint32_t)sc_.y;                                       \
        SA_CELL_NB(B, 4 * (G), ma_, mb0_, e[(G) >> 1])                                          \
        if (half_) return;                                                                      \
        uint32_t mb1_ = B[4 * (G) + 2] + (uint32_t)sc_.z;                                       \
        uint32_t man_ = 0;                                                                      \
        if (!last_) asm volatile("v_add_u32 %0, %1, %2" : "=v"(man_) : "v"(B[4 * (G) + 3]), "v"(sc_.w)); \
        SA_CELL_NB(B, 4 * (G) + 2, ma1_, mb1_, e[(G) >> 1])                                     \
        if (last_ || (END2 == 0 && gend == (G) + 1)) return;                                    \
        ma_ = man_;                                                                             \
    }

// One packed DP row of 16*D cells, in place: B = the row above on entry, this row on return, columns
// 0 .. 4*gend-1.  diag0 = border value of the row above in odd-column bank form (SA_NB_UP_B on the matrix
// border), left0 = this row's border value with the tags cleared (0 on the matrix border).  row_base: the
// lane's profile row, PS dwords long, laid out as described above.
// e[w] receives the tags of columns 8w .. 8w+7 of both pairs: column 8w+i at bits 2(7-i) of each 16-bit
// half, also when the row ends after four columns of the word.
// WIDE: the maxima as unsigned 16-bit instructions (SA_NB_HEAD_WIDE).
template <int NG, int PS, int END2 = 0, bool WIDE = false>
__device__ __forceinline__ void fill_row_pair(uint32_t (&B)[4 * NG], const int gend,
                                              const uint32_t diag0, const uint32_t left0,
                                              const int32_t* __restrict__ row_base, uint32_t (&e)[(NG + 1) / 2]) {
    uint32_t zl = left0;
    uint32_t upb = SA_NB_UP_B;
    asm volatile("" : "+v"(upb));          // one register for the whole row, not a copy per use
    // the lane's profile row as an LDS byte address: the four entries of a group come as two 8-byte reads (rows are
    // 2 banks apart modulo 64, see sa_device.h: 16-byte reads of rows 4 banks apart collide for residues b and b + 16)
    const uint32_t row_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int32_t*)row_base;
    sa_i32x2 pa_lo, pa_hi, pb_lo, pb_hi;
    SA_PROF_REQUEST(pa_lo, pa_hi, 0);
    uint32_t ma_ = diag0 + (uint32_t)row_base[PS - 1];
    SA_GROUP_PK(0) SA_GROUP_PK(1) SA_GROUP_PK(2) SA_GROUP_PK(3) SA_GROUP_PK(4) SA_GROUP_PK(5) SA_GROUP_PK(6)
    SA_GROUP_PK(7) SA_GROUP_PK(8) SA_GROUP_PK(9) SA_GROUP_PK(10) SA_GROUP_PK(11) SA_GROUP_PK(12)
    if constexpr (NG > 13) { SA_GROUP_PK(13) SA_GROUP_PK(14) SA_GROUP_PK(15) }
}

// dword of a profile row that takes the table entry of column c (see above)
#define SA_NB_PROF_INDEX(c, PS) ((c) == 0 ? (PS) - 1 : (c) - 1)


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
