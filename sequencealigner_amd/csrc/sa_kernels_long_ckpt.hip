// sa_kernels_long_ckpt.hip -- TRIANGULAR, sequences longer than 64 residues, two first sequences per wave in
// packed 16-bit cells, WITHOUT a direction matrix: a values-only sweep that leaves checkpoints, then a traceback
// that fills again -- with direction tags -- only the tiles its path goes through.
//
// The kernel it replaces by default (sa_kernels_long_pair.hip) writes two direction bits for every cell of every
// matrix and its walk reads back the tenth of them the paths touch; producing the tags is what makes its fill 2.4
// instructions per cell.  Here (reference: include/seqalign.h:47-88, fill and traceback of one pair):
//
//  pass 1  the sweep of sa_kernels_long_pair.hip (vertical strips of 64 columns, one lane per second sequence, the
//          two first sequences of the wave in the two halves of every register, D minus a per-lane base that
//          follows the strip's border column) on plain values: per column of both pairs one add and one three-way
//          maximum (v_pk_maximum3_f16 on bit patterns below 0x7c00, sa_pair_common.h), no tag, no direction
//          store.  It leaves
//            * the last column of every strip (as before, but of all strips, not the last two), and
//            * every CK-th row of every strip (64 registers per lane),
//          both as the low 16 bits of the values, the two halves side by side: a tile only ever uses such a value
//          minus the tile's base, a difference below 2^12.
//  pass 2  strips from the right to the left.  Every lane keeps, per half, where its path stands (rows and columns
//          still to go).  While some lane's path is in the strip: each lane takes the tile (CK rows x 64 columns)
//          its path stands in -- the lanes of a wave may stand in different tiles of the strip, rows are a lane's
//          own data -- restores the row above it from the checkpoint, fills the tile with the tagged cells of
//          sa_pair_common.h up to the row the path enters at, writes the tags to a small tile area (CK x 1 KiB per
//          half) and walks them until the path leaves the tile upwards (next tile of the strip, next round) or to
//          the left (next strip).  A path crosses about (strips + row blocks) of the strips x row blocks tiles.
//
// Everything a lane reads from global scratch (border columns, checkpoints, tile tags) it wrote itself.
// Records are those of sa_device.h (long form), byte-identical to the other long-sequence kernels'.
#include <type_traits>

#include "sa_pair_common.h"
#include "sa_long_walk.h"

#define SA_LCK_PS SA_PAIR_PROF_STRIDE(16)   /* 66 dwords; the LDS is sized for 68 (SA_LDS_PROF_BYTES) */

// packed 16-bit arithmetic on the two halves of a dword (no carry or borrow from one half into the other)
__device__ __forceinline__ uint32_t lck_pk_sub(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t lck_pk_add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t lck_pk_shl4(uint32_t a) {     // each half shifted left by SA_NB_SHIFT
    uint32_t r;
    asm("v_pk_lshlrev_b16 %0, 4, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}

// ---- pass 1: one row of plain values, in place.  B = the row above on entry, this row on return, both pairs.
// diag0 / left0 = the border values of the row above / of this row (minus the lane's base).  row_base: the lane's
// profile row -- plain S' of the two first sequences side by side, laid out as fill_row_pair's (dword k = column
// k + 1, column 0 in the row's last dword).  Four columns are one block of instructions: the four diagonal operands
// are taken from the row above before the maxima overwrite it.
// A row of values only is two instructions per column, so a group of four columns no longer covers the latency of
// the LDS read that feeds the next one (the tagged row has five per column): table entries come eight columns at a
// time, requested eight columns ahead, two register sets in turn.  The requests and waits are instructions the
// compiler cannot see through; the waits are counts (LDS reads return in order): "all but the request just sent".
// FULL: the strip is 64 columns wide, the row is straight-line code; otherwise it ends after `oend` octets
// (the columns past the first sequences' ends hold padding and are never read).
#define SA_SC_ADDS_MAXS(NEXT_DIAG)                                   \
    "v_add_u32 %[mb0], %[b0], %[sx]\n\t"                             \
    "v_add_u32 %[ma1], %[b1], %[sy]\n\t"                             \
    "v_pk_maximum3_f16 %[b0], %[ma], %[b0], %[zl]\n\t"               \
    "v_add_u32 %[mb1], %[b2], %[sz]\n\t"                             \
    "v_pk_maximum3_f16 %[b1], %[mb0], %[b1], %[b0]\n\t"              \
    NEXT_DIAG                                                        \
    "v_pk_maximum3_f16 %[b2], %[ma1], %[b2], %[b1]\n\t"              \
    "v_pk_maximum3_f16 %[b3], %[mb1], %[b3], %[b2]"
// the same with every three-way maximum as two unsigned 16-bit maxima (tables whose values leave the range of the
// half-precision patterns: three instructions per column instead of two)
#define SA_SC_ADDS_MAXS_WIDE(NEXT_DIAG)                              \
    "v_add_u32 %[mb0], %[b0], %[sx]\n\t"                             \
    "v_add_u32 %[ma1], %[b1], %[sy]\n\t"                             \
    "v_pk_max_u16 %[b0], %[ma], %[b0]\n\t"                           \
    "v_pk_max_u16 %[b0], %[b0], %[zl]\n\t"                           \
    "v_add_u32 %[mb1], %[b2], %[sz]\n\t"                             \
    "v_pk_max_u16 %[b1], %[mb0], %[b1]\n\t"                          \
    "v_pk_max_u16 %[b1], %[b1], %[b0]\n\t"                           \
    NEXT_DIAG                                                        \
    "v_pk_max_u16 %[b2], %[ma1], %[b2]\n\t"                          \
    "v_pk_max_u16 %[b2], %[b2], %[b1]\n\t"                           \
    "v_pk_max_u16 %[b3], %[mb1], %[b3]\n\t"                          \
    "v_pk_max_u16 %[b3], %[b3], %[b2]"
// four columns 4G .. 4G+3 with the table entries SC (x .. w = columns 4G+1 .. 4G+4)
#define SA_SC_GROUP_FORM(TEXT, G, SC)                                                               \
    {                                                                                               \
        uint32_t mb0_, ma1_, mb1_;                                                                  \
        if ((G) == SA_LP / 4 - 1) {                                                                 \
            asm volatile(TEXT("")                                                                   \
                         : [b0] "+v"(B[4 * (G)]), [b1] "+v"(B[4 * (G) + 1]), [b2] "+v"(B[4 * (G) + 2]),       \
                           [b3] "+v"(B[4 * (G) + 3]), [mb0] "=&v"(mb0_), [ma1] "=&v"(ma1_), [mb1] "=&v"(mb1_) \
                         : [ma] "v"(ma_), [sx] "v"(SC.x), [sy] "v"(SC.y), [sz] "v"(SC.z), [zl] "v"(B[4 * (G) - 1]));     \
        } else {                                                                                    \
            asm volatile(TEXT("v_add_u32 %[ma], %[b3], %[sw]\n\t")                                  \
                         : [b0] "+v"(B[4 * (G)]), [b1] "+v"(B[4 * (G) + 1]), [b2] "+v"(B[4 * (G) + 2]),       \
                           [b3] "+v"(B[4 * (G) + 3]), [mb0] "=&v"(mb0_), [ma1] "=&v"(ma1_), [mb1] "=&v"(mb1_), \
                           [ma] "+v"(ma_)                                                           \
                         : [sx] "v"(SC.x), [sy] "v"(SC.y), [sz] "v"(SC.z), [sw] "v"(SC.w),                 \
                           [zl] "v"((G) == 0 ? left0 : B[(G) == 0 ? 0 : 4 * (G) - 1]));                  \
        }                                                                                           \
    }
#define SA_SC_GROUP(G, SC)                                                                          \
    if constexpr (WIDE) SA_SC_GROUP_FORM(SA_SC_ADDS_MAXS_WIDE, G, SC)                               \
    else SA_SC_GROUP_FORM(SA_SC_ADDS_MAXS, G, SC)
typedef int sa_i32x4v __attribute__((ext_vector_type(4)));
#ifdef SA_LCK_EXP_NO_LDS     /* experiment: no table reads at all (wrong results, same arithmetic) */
#define SA_SC_REQ8(LO, HI, O) asm volatile("" : "=v"(LO), "=v"(HI))
#define SA_SC_PENDING "0"
#elif !defined(SA_LCK_READ2_B64)    /* four 8-byte reads: a ds_read2_b64 takes four times the LDS cycles of a ds_read_b64 */
#define SA_SC_REQ8(LO, HI, O)                                                                                       \
    asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\tds_read_b64 %3, %4 offset:%8" \
                 : "=&v"(LO.lo), "=&v"(LO.hi), "=&v"(HI.lo), "=&v"(HI.hi) : "v"(row_lds), "n"(32 * (O)), "n"(32 * (O) + 8), "n"(32 * (O) + 16), "n"(32 * (O) + 24))
#define SA_SC_PENDING "4"
#else
#define SA_SC_PENDING "2"
#define SA_SC_REQ8(LO, HI, O)                                                                                       \
    asm volatile("ds_read2_b64 %0, %2 offset0:%3 offset1:%4\n\tds_read2_b64 %1, %2 offset0:%5 offset1:%6"            \
                 : "=&v"(LO), "=&v"(HI) : "v"(row_lds), "n"(4 * (O)), "n"(4 * (O) + 1), "n"(4 * (O) + 2), "n"(4 * (O) + 3))
#endif
// Between a request and its wait the code is straight-line: every octet but the last requests its successor (in a
// row that ends early the request past the end reads entries nobody uses, still inside the profile row) and waits
// for "all but that request".  With the request under a condition the two paths join again before the wait, and
// the compiler copies the registers of a read still under way at the join.
#define SA_SC_OCTET(O, LO, HI, NLO, NHI)                                                            \
    {                                                                                               \
        constexpr bool last_ = (O) == SA_LP / 8 - 1;                                                \
        if (!last_) {                                                                               \
            SA_SC_REQ8(NLO, NHI, (O) + 1);                                                          \
            if ((O) == 0) asm volatile("s_waitcnt lgkmcnt(" SA_SC_PENDING ")" : "+v"(LO), "+v"(HI), "+v"(c0_));     \
            else asm volatile("s_waitcnt lgkmcnt(" SA_SC_PENDING ")" : "+v"(LO), "+v"(HI));                         \
        } else {                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(LO), "+v"(HI));                              \
        }                                                                                           \
        if ((O) == 0) ma_ = diag0 + c0_;                                                            \
        SA_SC_GROUP(2 * (O), LO)                                                                    \
        SA_SC_GROUP(2 * (O) + 1, HI)                                                                \
        if (last_) return;                                                                          \
        if (!FULL && oend == (O) + 1) {                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(NLO), "+v"(NHI));    /* nothing in flight past the row */ \
            return;                                                                                 \
        }                                                                                           \
    }

template <bool FULL, bool WIDE = false>
__device__ __forceinline__ void fill_row_score(uint32_t (&B)[SA_LP], const int oend, const uint32_t diag0,
                                               const uint32_t left0, const int32_t* __restrict__ row_base) {
    const uint32_t row_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int32_t*)row_base;
    sa_i32x4v pa_lo, pa_hi, pb_lo, pb_hi;
    uint32_t c0_, ma_ = 0;
    // column 0's table entry (the row's last dword), then the first eight columns
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(c0_) : "v"(row_lds), "n"(4 * (SA_LCK_PS - 1)));
    SA_SC_REQ8(pa_lo, pa_hi, 0);
    SA_SC_OCTET(0, pa_lo, pa_hi, pb_lo, pb_hi) SA_SC_OCTET(1, pb_lo, pb_hi, pa_lo, pa_hi)
    SA_SC_OCTET(2, pa_lo, pa_hi, pb_lo, pb_hi) SA_SC_OCTET(3, pb_lo, pb_hi, pa_lo, pa_hi)
    SA_SC_OCTET(4, pa_lo, pa_hi, pb_lo, pb_hi) SA_SC_OCTET(5, pb_lo, pb_hi, pa_lo, pa_hi)
    SA_SC_OCTET(6, pa_lo, pa_hi, pb_lo, pb_hi) SA_SC_OCTET(7, pb_lo, pb_hi, pa_lo, pa_hi)
}


#ifndef SA_LCK_ROW_BENCH     /* tools/row_bench.hip takes the row functions alone */
// G: rows between checkpoints; T: rows of a tile (a multiple of G).  A lane's tile starts on the checkpoint grid, as
// low as leaves the row its path stands in inside the tile's last G rows: with T > G a path then covers at least
// T - G rows of every tile it goes through upwards, where tiles on a grid of their own height are entered anywhere.
template <int G, int T, bool WIDE>
__device__ __forceinline__ void sa_long_ckpt_body(const SaPepParams& P) {
    static_assert(G % 16 == 0 && 64 % G == 0 && T % G == 0 && T % 8 == 0 && T <= 64, "checkpoint rows 16, 32 or 64; tile rows a multiple, at most 64");
    extern __shared__ __align__(16) uint8_t smem[];
    int32_t* const prof = reinterpret_cast<int32_t*>(smem);
    uint8_t* const pseq = smem + SA_LDS_PROF_BYTES + 64;      // the two shared first sequences, lslot bytes each
    const int lane = threadIdx.x;
    const uint32_t rows_cap = P.spill_rows;                    // rows of border columns and checkpoints: a multiple of 64
    const uint32_t strips_cap = P.lslot >> 6;
    // per-workgroup scratch, dwords: tile tags (2 halves x T rows x 256) | border columns (strips x rows x 64) |
    // checkpoints (row blocks x strips x 64 columns x 64 lanes)
    const size_t tile_dw = (size_t)2 * T * 256;
    const size_t bcol_dw = (size_t)strips_cap * rows_cap * 64;
    const size_t ckp_dw = (size_t)(rows_cap / G) * strips_cap * 4096;
    uint32_t* const tile = P.spill + (size_t)blockIdx.x * (tile_dw + bcol_dw + ckp_dw);
    uint32_t* const bcol = tile + tile_dw;
    uint32_t* const ckp = bcol + bcol_dw;
    const int slot_last = (int)P.lslot - 1;
    const int rows_last = (int)rows_cap - 1;
    // optional phase timers: 0 fetch, 2 operands, 3 values-only sweep, 5 tiles filled again, 4 walk + record
    const bool timed = P.phase_cycles != nullptr;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // 6, 7: rows filled again, rows offered to the walk
    unsigned long long tmark = timed ? __builtin_amdgcn_s_memtime() : 0;
#define SA_PHASE(k) do { if (timed) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tmark; tmark = t_; } } while (0)

    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(P.counter, 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= P.nitems) break;
        const SaItem item = P.items[it];
        SA_PHASE(0);

        const uint32_t p1 = item.p, p2 = item.reserved[0];
        const bool has2 = p2 != SA_INVALID_FI;
        const int lenp1 = P.len_fo[p1];
        const int lenp2 = has2 ? (int)P.len_fo[p2] : 0;
        __syncthreads();      // the previous item's walks are done with the copies
        for (uint32_t o = lane * 16; o < P.lslot; o += 64 * 16) {
            *reinterpret_cast<uint4*>(pseq + o) = *reinterpret_cast<const uint4*>(P.codes_fo + (size_t)p1 * P.lslot + o);
            uint4 second = make_uint4(0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u);      // SA_PAD_CODE
            if (has2) second = *reinterpret_cast<const uint4*>(P.codes_fo + (size_t)p2 * P.lslot + o);
            *reinterpret_cast<uint4*>(pseq + P.lslot + o) = second;
        }
        __syncthreads();
        const uint64_t n = P.nseq;
        uint8_t* const out_row1 = P.out + ((uint64_t)p1 * (2 * n - p1 - 1) / 2 - P.out_base_pair) * P.stride;
        uint8_t* const out_row2 = has2 ? P.out + ((uint64_t)p2 * (2 * n - p2 - 1) / 2 - P.out_base_pair) * P.stride : out_row1;
        const int maxlenp = max(lenp1, lenp2);
        const int nstrips = (maxlenp + 63) >> 6;
        const int gend_last = ((maxlenp - 64 * (nstrips - 1)) + 3) >> 2;

        // the packed query profile of columns 64s .. 64s+63 of the two first sequences: plain S' for pass 1, the
        // tagged cells' table entries for pass 2
        auto build_profile = [&](int s, bool tagged) {
            const int a1 = pseq[64 * s + lane], a2 = pseq[P.lslot + 64 * s + lane];
            __syncthreads();
            // all forty table values are requested before the first is used (one round trip, not twenty)
            int t1[20], t2[20];
#pragma unroll
            for (int b = 0; b < 20; b++) {
                t1[b] = P.sprime[a1 * SA_SPRIME_STRIDE + b];
                t2[b] = P.sprime[a2 * SA_SPRIME_STRIDE + b];
            }
            const int idx = SA_NB_PROF_INDEX(lane, SA_LCK_PS);
#pragma unroll
            for (int b = 0; b < 20; b++) {
                // padding scores 1 instead of 0 (an odd column's table entry must stay positive); padded cells lie to
                // the right of and below every cell that is read
                const int s1 = max(t1[b], 1), s2 = max(t2[b], 1);
                uint32_t v = (uint32_t)s1 | ((uint32_t)s2 << 16);
                if (tagged) v = (lane & 1) ? SA_NB_TABLE_B(s1) | (SA_NB_TABLE_B(s2) << 16) : SA_NB_TABLE_A(s1) | (SA_NB_TABLE_A(s2) << 16);
                prof[b * SA_LCK_PS + idx] = (int32_t)v;
            }
            uint32_t vpad = 0x00010001u;
            if (tagged) vpad = (lane & 1) ? SA_NB_TABLE_B(1) * 0x10001u : SA_NB_TABLE_A(1) * 0x10001u;
            prof[20 * SA_LCK_PS + idx] = (int32_t)vpad;
            __syncthreads();
        };

        for (uint32_t task = 0; task < item.count; task++) {
            // ---------------------------------------------------------- operands
            const uint32_t vbase = item.v0 + task * 64;
            const uint32_t vl = vbase + lane;
            const uint32_t nvalid = P.nseq - 1 - p1;
            const bool active = vl < nvalid;
            if (!__any(active)) continue;
            uint32_t sidx = 0, cum = item.c0;
            const uint32_t* cur_row = P.cursor + (size_t)p1 * P.nb;
            for (uint32_t b = item.b0; b < P.nb; b++) {
                const uint32_t first = P.bucket_base[b] + cur_row[b];
                const uint32_t cnt = P.bucket_base[b + 1] - first;
                if (vl >= cum && vl < cum + cnt) sidx = first + (vl - cum);
                cum += cnt;
                if (cum > vbase + 63) break;
            }
            const uint32_t fiq = P.fi_sorted[sidx];
            const int lenq = active ? (int)P.len_sorted[sidx] : 0;
            const bool active2 = active && has2 && fiq > p2;
            const uint8_t* const bslot = P.codes_fo + (size_t)fiq * P.lslot;      // sequence 2 as one padded slot
            const int maxlen = wave_max_nonneg(lenq);
            SA_PHASE(2);

            // ------------------------------------------------- pass 1: values only, checkpoints
            int dfin1 = 0, dfin2 = 0;           // D at the last cell of each half, absolute
            for (int s = 0; s < nstrips; s++) {
                const int gend = (s == nstrips - 1) ? gend_last : SA_LP / 4;
                build_profile(s, false);
                uint32_t R[SA_LP];        // the DP row of the strip, updated in place: D - base, both halves
#pragma unroll
                for (int c = 0; c < SA_LP; c++) R[c] = 0;
                int capA = 0, capB = 0;   // absolute D at the last column of each half, taken at the lane's last row
                const int colA = (lenp1 - 1) & 63, colB = (max(lenp2, 1) - 1) & 63;
                int base1 = 0, base2 = 0;     // the bases as whole numbers (for the scores) ...
                uint32_t base12 = 0;          // ... and their low 16 bits side by side, as the border column has them
                // border column of strip s: [row / 4][lane][row % 4], one dword per row and lane
                const uint4* bin = reinterpret_cast<const uint4*>(bcol + (size_t)max(s - 1, 0) * rows_cap * 64) + lane;
                uint4* bout = reinterpret_cast<uint4*>(bcol + (size_t)s * rows_cap * 64) + lane;
                const bool first_strip = s == 0;
                auto residues4 = [&](int r) -> uint32_t {
                    return *reinterpret_cast<const uint32_t*>(bslot + min(r, slot_last - 3));
                };
                auto border4 = [&](int r) -> uint4 {
                    return first_strip ? make_uint4(0, 0, 0, 0) : bin[(size_t)(min(r, rows_last - 3) >> 2) * 64];
                };
                uint32_t bres = residues4(0);
                uint4 bord = border4(0);
                uint32_t prevw = 0;                              // border dword of the row above the block
                // Every lane computes every row of its wave, also lanes whose sequence 2 has ended (padding residues,
                // results never used); a lane takes its scores when its last row is done.
#define SA_LCK_ROW1(RR, DW, LW, B)                                                                         \
    {                                                                                                      \
        int oe_ = oend;                                                                                    \
        asm volatile("" : "+s"(oe_));                                                                      \
        fill_row_score<decltype(full_tag)::value, WIDE>(R, oe_, (DW), (LW), prof + (B) * SA_LCK_PS);             \
    }                                                                                                      \
    if (decltype(cap_tag)::value && (RR) + 1 == lenq) {                                                    \
        capA = (int)(pick_packed<SA_LP>(R, colA) & 0xffffu) + base1;                                       \
        capB = (int)(pick_packed<SA_LP>(R, colB) >> 16) + base2;                                           \
    }
                const int oend = (gend + 1) >> 1;     // octets of a row
                // full_tag: the strip is 64 columns wide; cap_tag: a half's last column lies in this strip, so the rows
                // look for the lane's last row (elsewhere the loop carries no such test and no column pick).
                // Rows go in blocks of four inside runs of G rows.  Every block issues two loads (next residues, next
                // border values) and one store (its last-column values; in the last strip nobody reads them), always, so
                // that the wait for the loads at the top of the next block is a count that leaves the store in flight;
                // the checkpoint's sixteen stores stand after the run, outside the loop of the blocks.  (With the
                // checkpoint under a condition inside that loop the compiler waits for everything in flight at the top
                // of every block: 13 % more time for the sweep.)
                auto sweep = [&](auto full_tag, auto cap_tag) {
                for (int rb = 0; rb < maxlen; rb += G) {
                    const int rend = min(rb + G, maxlen);
                    for (int r = rb; r < rend; r += 4) {
                        const uint32_t nres = residues4(r + 4);
                        const uint4 nbord = border4(r + 4);
                        const uint32_t rb0 = bres & 0xff, rb1 = (bres >> 8) & 0xff, rb2 = (bres >> 16) & 0xff, rb3 = bres >> 24;
                        if ((r & 127) == 0 && r > 0 && !first_strip) {
                            // move the bases up to the border of the row above; the register row follows (values stay
                            // below (129 + 64) * max S', far from 0x7c00)
                            const uint32_t d = lck_pk_sub(prevw, base12);
#pragma unroll
                            for (int c = 0; c < SA_LP; c++) R[c] -= d;
                            base1 += (int)(d & 0xffffu); base2 += (int)(d >> 16);
                            base12 = prevw;
                        }
                        const uint32_t l0 = lck_pk_sub(prevw, base12), l1 = lck_pk_sub(bord.x, base12), l2 = lck_pk_sub(bord.y, base12),
                                       l3 = lck_pk_sub(bord.z, base12), l4 = lck_pk_sub(bord.w, base12);
                        SA_LCK_ROW1(r, l0, l1, rb0)
                        const uint32_t y0 = R[SA_LP - 1];
                        SA_LCK_ROW1(r + 1, l1, l2, rb1)
                        const uint32_t y1 = R[SA_LP - 1];
                        SA_LCK_ROW1(r + 2, l2, l3, rb2)
                        const uint32_t y2 = R[SA_LP - 1];
                        SA_LCK_ROW1(r + 3, l3, l4, rb3)
                        const uint32_t y3 = R[SA_LP - 1];
                        bout[(size_t)(r >> 2) * 64] = make_uint4(lck_pk_add(y0, base12), lck_pk_add(y1, base12), lck_pk_add(y2, base12), lck_pk_add(y3, base12));
                        prevw = bord.w;
                        bres = nres;
                        bord = nbord;
                    }
                    if (rb + G < maxlen) {
                        // checkpoint: the row just done, every column, low 16 bits of the values
                        uint4* ck = reinterpret_cast<uint4*>(ckp + ((size_t)((rb / G) * strips_cap + s) * 16) * 256) + lane;
#pragma unroll
                        for (int c4 = 0; c4 < 16; c4++)
                            ck[(size_t)c4 * 64] = make_uint4(lck_pk_add(R[4 * c4], base12), lck_pk_add(R[4 * c4 + 1], base12),
                                                            lck_pk_add(R[4 * c4 + 2], base12), lck_pk_add(R[4 * c4 + 3], base12));
                    }
                }
                };
                const bool cap_here = ((lenp1 - 1) >> 6) == s || (has2 && ((lenp2 - 1) >> 6) == s);
                if (gend == SA_LP / 4) {
                    if (cap_here) sweep(std::true_type(), std::true_type()); else sweep(std::true_type(), std::false_type());
                } else {
                    sweep(std::false_type(), std::true_type());      // the last strip holds a last column
                }
#undef SA_LCK_ROW1
                // the scores sit in the strips that hold each half's last column
                if (((lenp1 - 1) >> 6) == s) dfin1 = capA;
                if (has2 && ((lenp2 - 1) >> 6) == s) dfin2 = capB;
            }
            SA_PHASE(3);

            // ------------------------------------------------- pass 2: tiles on the paths, tags, walk
            // per half: i rows and j columns still to go (the path stands in cell (i, j)); run = LEFT moves already
            // taken in row i (a run of LEFT moves may go through several strips)
            int wi[2] = {active ? lenq : 0, active2 ? lenq : 0};
            int wj[2] = {lenp1, lenp2};
            int wrun[2] = {0, 0}, wmatch[2] = {0, 0}, wgaps[2] = {0, 0};
            // A path finishes its rows from the last to the first, so a record is written eight rows (16 bytes) at a time:
            // the rows' codes are shifted into a 128-bit register set, newest lowest, and the set is stored when the row
            // that starts a group of eight is done.  (One 2-byte store per row and lane -- 64 partial lines per wave and
            // row -- cost a third of the launch.)  Rows of the first group beyond the sequence's end stay zero.
            uint32_t wacc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            uint8_t* const rec0 = out_row1 + (size_t)(fiq - p1 - 1) * P.stride;
            uint8_t* const rec1 = out_row2 + (size_t)(fiq - p2 - 1) * P.stride;
#ifdef SA_LCK_EXP_NO_PASS2
            wi[0] = wi[1] = 0;       // experiment: the values-only sweep alone (records are wrong)
#endif
            for (int s = nstrips - 1; s >= 0; s--) {
                const int cbase = 64 * s;
                bool in0 = wi[0] > 0 && wj[0] > cbase, in1 = wi[1] > 0 && wj[1] > cbase;
                if (!__any(in0 || in1)) continue;
                const int gend = (s == nstrips - 1) ? gend_last : SA_LP / 4;
                const bool first_strip = s == 0;
                build_profile(s, true);
                const uint4* bin = reinterpret_cast<const uint4*>(bcol + (size_t)max(s - 1, 0) * rows_cap * 64) + lane;
                while (__any(in0 || in1)) {
                    // the tile this lane fills: T rows from a checkpoint, placed so that the lower of its two paths stands
                    // in the tile's last G rows (or, near the top of the matrix, in the rows there are)
                    const int rlow = max(in0 ? wi[0] - 1 : -1, in1 ? wi[1] - 1 : -1);          // -1: neither path is in the strip
                    const int kl = max(rlow / G - (T / G - 1), 0);                              // checkpoint grid index of the tile's first row
                    const int r0 = kl * G;
                    const bool t0 = in0 && wi[0] - 1 >= r0, t1 = in1 && wi[1] - 1 >= r0;        // halves that walk this tile
                    // rows 0 .. tneed of the tile are wanted, columns 0 .. jneed - 1 of the strip
                    const int tneed = max(t0 ? wi[0] - 1 - r0 : -1, t1 ? wi[1] - 1 - r0 : -1);
                    const int jneed = max(t0 ? wj[0] - cbase : 0, t1 ? wj[1] - cbase : 0);
                    const int tmax = wave_max_nonneg(tneed + 1) - 1;
                    const int gneed = min(gend, (wave_max_nonneg(jneed) + 3) >> 2);
                    // ---- the row above the tile, from the checkpoint; base = the border value of that row
                    // (all loads are issued for every lane, whatever its tile, and the lanes of tile 0 mask the values
                    // out: with the loads under per-lane conditions every register gets its own branch and its own wait)
                    uint32_t R[SA_LP];
                    const uint32_t have_above = kl > 0 ? 0xffffffffu : 0u;            // rows above the tile exist
                    uint32_t prevw = first_strip ? 0u : (bin[(size_t)(max(r0 - 4, 0) >> 2) * 64].w & have_above);
                    uint32_t base12 = prevw;
                    {
                        const uint4* ck = reinterpret_cast<const uint4*>(ckp + ((size_t)((max(kl, 1) - 1) * strips_cap + s) * 16) * 256) + lane;
                        uint4 cv[16];
#pragma unroll
                        for (int c4 = 0; c4 < 16; c4++) cv[c4] = ck[(size_t)c4 * 64];
#pragma unroll
                        for (int c4 = 0; c4 < 16; c4++) {
                            const uint32_t w[4] = {cv[c4].x, cv[c4].y, cv[c4].z, cv[c4].w};
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const uint32_t up = (q & 1) ? SA_NB_UP_B : SA_NB_UP_A;
                                uint32_t v;
                                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(v) : "v"(lck_pk_shl4(lck_pk_sub(w[q], base12))), "v"(have_above), "v"(up));
                                R[4 * c4 + q] = v;
                            }
                        }
                    }
                    uint32_t e[8];
#pragma unroll
                    for (int w = 0; w < 8; w++) e[w] = 0;
                    uint32_t* const drowA = tile + (size_t)lane * 4;
                    uint32_t* const drowB = drowA + (size_t)T * 256;
                    auto residues4 = [&](int r) -> uint32_t {
                        return *reinterpret_cast<const uint32_t*>(bslot + min(r, slot_last - 3));
                    };
                    auto border4 = [&](int r) -> uint4 {
                        return first_strip ? make_uint4(0, 0, 0, 0) : bin[(size_t)(min(r, rows_last - 3) >> 2) * 64];
                    };
                    auto rel = [&](uint32_t w) -> uint32_t { return lck_pk_shl4(lck_pk_sub(w, base12)); };
                    const int tnA = t0 ? wi[0] - 1 - r0 : -1, tnB = t1 ? wi[1] - 1 - r0 : -1;
                    auto store_dirs = [&](int t) {
                        uint32_t lo[4], hi[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t w0 = e[2 * k], w1 = e[2 * k + 1];
                            lo[k] = __builtin_amdgcn_perm(w0, w1, 0x05040100u);     // low halves:  w0.lo : w1.lo
                            hi[k] = __builtin_amdgcn_perm(w0, w1, 0x07060302u);     // high halves: w0.hi : w1.hi
                        }
                        // a lane keeps the rows its own paths can reach: up to the row each stands in (the wave fills the
                        // tile as far as its lowest path needs)
                        if (t <= tnA) *reinterpret_cast<uint4*>(drowA + (size_t)t * 256) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                        if (t <= tnB) *reinterpret_cast<uint4*>(drowB + (size_t)t * 256) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    };
                    uint32_t bres = residues4(r0);
                    uint4 bord = border4(r0);
#define SA_LCK_ROW2(TT, DW, LW, B)                                                                         \
    {                                                                                                      \
        int gr_ = gneed;                                                                                   \
        asm volatile("" : "+s"(gr_));                                                                      \
        fill_row_pair<SA_LP / 4, SA_LCK_PS, 0, WIDE>(R, gr_, (DW) | SA_NB_UP_B, (LW), prof + (B) * SA_LCK_PS, e);  \
    }                                                                                                      \
    store_dirs(TT);
                    if (timed) tph[6] += (unsigned)((tmax + 4) & ~3);
                    for (int t = 0; t <= tmax; t += 4) {
                        const uint32_t nres = residues4(r0 + t + 4);
                        const uint4 nbord = border4(r0 + t + 4);
                        const uint32_t rb0 = bres & 0xff, rb1 = (bres >> 8) & 0xff, rb2 = (bres >> 16) & 0xff, rb3 = bres >> 24;
                        if ((t & (int)P.rebase_mask) == 0 && t > 0 && !first_strip) {
                            // move the bases up to the border of the row above; the register row follows
                            const uint32_t dd = lck_pk_shl4(lck_pk_sub(prevw, base12));
#pragma unroll
                            for (int c = 0; c < SA_LP; c++) R[c] -= dd;
                            base12 = prevw;
                        }
                        const uint32_t l0 = rel(prevw), l1 = rel(bord.x), l2 = rel(bord.y), l3 = rel(bord.z), l4 = rel(bord.w);
                        SA_LCK_ROW2(t, l0, l1, rb0)
                        SA_LCK_ROW2(t + 1, l1, l2, rb1)
                        SA_LCK_ROW2(t + 2, l2, l3, rb2)
                        SA_LCK_ROW2(t + 3, l3, l4, rb3)
                        prevw = bord.w;
                        bres = nres;
                        bord = nbord;
                    }
#undef SA_LCK_ROW2
                    // The walk reads what this lane has just stored, and a load may overtake a store to the same address on
                    // its way to the L2 (seen: one wrong tag in 10^7 rows without this wait): the stores are acknowledged
                    // before the first request of the walk leaves.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    SA_PHASE(5);

                    // ---- walk, both halves side by side, eight rows per request.  A chunk's sixteen tag rows and its eight
                    // residues are requested and awaited by one block of instructions: left to the compiler a row's load
                    // moves into the branch that uses it and is waited for on the spot -- a trip to memory inside a row --
                    // and requests kept in flight across compiler-generated code are not safe (the compiler takes their
                    // registers for written at the request, and copies or spills them before the data has arrived).
                    int tc0 = t0 ? wi[0] - 1 - r0 : -1, tc1 = t1 ? wi[1] - 1 - r0 : -1;      // the row of the tile a path stands in, -1: none
                    const uint32_t voff = (uint32_t)lane * 16;
                    const uint8_t* const tile_a = reinterpret_cast<const uint8_t*>(tile);
                    const uint8_t* const tile_b = tile_a + (size_t)T * 1024;
                    const uint8_t* const resp = bslot + r0;
                    sa_u32x4v a0x, a1x, a2x, a3x, a4x, a5x, a6x, a7x, b0x, b1x, b2x, b3x, b4x, b5x, b6x, b7x;
                    sa_u32x2v rx;
                    for (int ch = tmax >> 3; ch >= 0; --ch) {
                        if (!__any(tc0 >= 0 || tc1 >= 0)) break;
                        if (timed) tph[7] += 8;
                        SA_LCK_FETCH(ch);
                        SA_LCK_ROWSTEP(ch, 7) SA_LCK_ROWSTEP(ch, 6) SA_LCK_ROWSTEP(ch, 5) SA_LCK_ROWSTEP(ch, 4)
                        SA_LCK_ROWSTEP(ch, 3) SA_LCK_ROWSTEP(ch, 2) SA_LCK_ROWSTEP(ch, 1) SA_LCK_ROWSTEP(ch, 0)
                    }
                    in0 = wi[0] > 0 && wj[0] > cbase;
                    in1 = wi[1] > 0 && wj[1] > cbase;
                    __syncthreads();      // the tile area is free for the next tile
                    SA_PHASE(4);
                }
            }
            // what is left of a path at the matrix's left border goes straight up; then header and zero tail
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool act = h ? active2 : active;
                uint8_t* const rec = h ? rec1 : rec0;
                if (act) {
                    for (int r = wi[h] - 1; r >= 0; r--) SA_LCK_PUSH(h, rec, r, 1)
                    wgaps[h] += wi[h];
                    const int lenp = h ? lenp2 : lenp1;
                    const int score = (h ? dfin2 : dfin1) + (lenp + lenq) * P.gap;
                    *reinterpret_cast<uint4*>(rec) = make_uint4((uint32_t)score, (uint32_t)wmatch[h], (uint32_t)wgaps[h], 0u);
                    // rows past the last one are zero, so that equal inputs give equal records
                    for (uint32_t o = SA_LONG_HEADER_BYTES + 2 * (((uint32_t)lenq + 7u) & ~7u); o < P.stride; o += 16)
                        *reinterpret_cast<uint4*>(rec + o) = make_uint4(0, 0, 0, 0);
                }
            }
            SA_PHASE(4);
        }
    }
    if (timed && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) P.phase_cycles[(size_t)blockIdx.x * 8 + k] = tph[k];
    }
#undef SA_PHASE
}

template <int G, int T>
__global__ void __launch_bounds__(64, 4) sa_long_ckpt_kernel(const SaPepParams P) {
    sa_long_ckpt_body<G, T, false>(P);
}
// the same kernel with its maxima as unsigned 16-bit instructions: tables whose values leave the half-precision range
template <int G, int T>
__global__ void __launch_bounds__(64, 4) sa_long_ckpt_wide_kernel(const SaPepParams P) {
    sa_long_ckpt_body<G, T, true>(P);
}

extern "C" void sa_launch_long_ckpt(int ck_rows, int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (wide)
        hipLaunchKernelGGL((sa_long_ckpt_wide_kernel<64, 64>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    else if (ck_rows == 32)
        hipLaunchKernelGGL((sa_long_ckpt_kernel<32, 64>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    else
        hipLaunchKernelGGL((sa_long_ckpt_kernel<64, 64>), dim3(grid), dim3(64), lds_bytes, stream, *p);
}
#endif
