// sa_kernels_long_pair.hip -- TRIANGULAR, sequences longer than 64 residues, two first
// sequences per wave in packed 16-bit cells (sa_pair_common.h).
//
// The sweep is the long-sequence kernel's (sa_kernels_long.hip): vertical strips of 64
// columns, rows in blocks of four, the last column of a strip handed to the next strip
// through a small global buffer.  D grows with the row index (up to length * max S'), far
// beyond 16 bits, but inside one strip it cannot run away from the strip's left border:
//   D[r][c] - D[r][c-1] <= max S'   and   D[r][c] - D[r-1][c] <= max S',   D monotone both ways,
// so every lane keeps, per half, a base that follows its border column and the registers hold
// D - base in the 16*D cell format of the peptide kernel (sa_pair_common.h: two tag fields,
// three-way half-precision maximum, values below 0x7c00).  The recurrence only compares the
// three operands, so a common offset changes neither the winner nor its tag; every 16 rows the
// base is moved up to the border value of the row above and the difference is subtracted from
// the register row (64 subtractions per 16 rows).  Between two moves a register holds at most
// 16 * (17 + 64) * max S' + 15 (border growth plus growth along the strip): the host uses this
// kernel only for tables with 1 <= S' <= 24 (BLOSUM50 with gap -4: 3 .. 23).
// The border column travels between strips as the low 16 bits of its values, the two halves side by side in one
// dword per row: a strip only uses a border value minus its base, a difference below 2^12 that the low 16 bits give
// whatever the whole values are (they pass 2^16 for sequences of a few thousand residues).
#include "sa_pair_common.h"
#include "sa_long_walk.h"

#define SA_LONG_PAIR_PROF_STRIDE SA_PAIR_PROF_STRIDE(16)   /* 66 dwords; the LDS is sized for 68 (SA_LDS_PROF_BYTES) */
#include "sa_traceback.h"

// packed 16-bit arithmetic on the two halves of a dword (no carry or borrow from one half into the other)
__device__ __forceinline__ uint32_t sa_pk_sub_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sa_pk_add_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sa_pk_shl4_u16(uint32_t a) {     // each half shifted left by SA_NB_SHIFT
    uint32_t r;
    asm("v_pk_lshlrev_b16 %0, 4, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}

template <bool WIDE>
__device__ __forceinline__ void sa_long_pair_body(const SaPepParams& P) {
    extern __shared__ __align__(16) uint8_t smem[];
    int32_t* const prof = reinterpret_cast<int32_t*>(smem);
    uint8_t* const pseq = smem + SA_LDS_PROF_BYTES + 64;      // the two shared first sequences, lslot bytes each
    const int lane = threadIdx.x;
    const uint32_t rows_cap = P.lslot;
    const uint32_t strips_cap = P.lslot >> 6;
    // direction rows: as many per strip as the longest uploaded sequence needs (a multiple of 8), not the slot size
    const uint32_t dir_rows = P.spill_rows;
    const size_t half_dirs = (size_t)strips_cap * dir_rows * 256;              // dwords of direction rows per half
    uint32_t* const dirs = P.spill + (size_t)blockIdx.x * 2 * half_dirs;
    const size_t half_bcol = (size_t)2 * rows_cap * 64;                        // ints of border column per half
    int32_t* const bcol = P.bcol + (size_t)blockIdx.x * 2 * half_bcol;
    const int last_row = (int)rows_cap - 1;
    // optional phase timers: 0 fetch, 2 operands, 3 fill, 4 traceback + record
    const bool timed = P.phase_cycles != nullptr;
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
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

            // -------------------------------------------------------------- fill
            int dfin1 = 0, dfin2 = 0;           // D at the last cell of each half, absolute
            for (int s = 0; s < nstrips; s++) {
                const int gend = (s == nstrips - 1) ? gend_last : SA_LP / 4;
                {
                    // packed query profile of columns 64s .. 64s+63 of the two first sequences
                    const int a1 = pseq[64 * s + lane], a2 = pseq[P.lslot + 64 * s + lane];
                    __syncthreads();
                    // all forty table values are requested before the first is used (one round trip, not twenty)
                    int t1[20], t2[20];
#pragma unroll
                    for (int b = 0; b < 20; b++) {
                        t1[b] = P.sprime[a1 * SA_SPRIME_STRIDE + b];
                        t2[b] = P.sprime[a2 * SA_SPRIME_STRIDE + b];
                    }
#pragma unroll
                    for (int b = 0; b < 20; b++) {
                        // padding scores 1 instead of 0 (an odd column's table entry must stay positive); padded
                        // cells lie to the right of and below every cell that is read
                        const int s1 = max(t1[b], 1), s2 = max(t2[b], 1);
                        prof[b * SA_LONG_PAIR_PROF_STRIDE + SA_NB_PROF_INDEX(lane, SA_LONG_PAIR_PROF_STRIDE)] = (lane & 1) ? SA_NB_TABLE_B(s1) | (SA_NB_TABLE_B(s2) << 16)
                                                                     : SA_NB_TABLE_A(s1) | (SA_NB_TABLE_A(s2) << 16);
                    }
                    prof[20 * SA_LONG_PAIR_PROF_STRIDE + SA_NB_PROF_INDEX(lane, SA_LONG_PAIR_PROF_STRIDE)] = (lane & 1) ? SA_NB_TABLE_B(1) * 0x10001 : SA_NB_TABLE_A(1) * 0x10001;
                    __syncthreads();
                }
                uint32_t R[SA_LP];        // the DP row of the strip, updated in place
#pragma unroll
                for (int c = 0; c < SA_LP; c++) R[c] = (c & 1) ? SA_NB_UP_B : SA_NB_UP_A;                // border row, base 0
                int capA = 0, capB = 0;   // absolute D at the last column of each half, taken at the lane's last row
                const int colA = (lenp1 - 1) & 63, colB = (max(lenp2, 1) - 1) & 63;
                uint32_t e[8];
#pragma unroll
                for (int w = 0; w < 8; w++) e[w] = 0;
                int base1 = 0, base2 = 0;     // the bases as whole numbers (for the scores) ...
                uint32_t base12 = 0;          // ... and their low 16 bits side by side, as the border column has them
                // Border column: [strip parity][row / 4][lane][row % 4], one dword per row: the low 16 bits of the value of
                // the first pair's matrix in the low half, of the second pair's in the high half.  The strip to the right
                // only ever uses a border value minus its base (the border value of at most twenty rows earlier): the
                // difference is below 2^12 and comes out of the low 16 bits alone, whatever the whole values are.
                // 8 bytes per row and lane of memory traffic instead of 16, one load and one store per block of four
                // rows instead of two, and the packed forms save the shifts and masks of two separate 32-bit values.
                const uint4* bin = reinterpret_cast<const uint4*>(bcol) + (size_t)(s & 1) * (rows_cap >> 2) * 64 + lane;
                uint4* bout = reinterpret_cast<uint4*>(bcol) + (size_t)((s + 1) & 1) * (rows_cap >> 2) * 64 + lane;
                uint32_t* drowA = dirs + ((size_t)s * dir_rows * 64 + lane) * 4;
                uint32_t* drowB = drowA + half_dirs;
                const bool first_strip = s == 0, more_strips = s + 1 < nstrips;

                auto residues4 = [&](int r) -> uint32_t {
                    return *reinterpret_cast<const uint32_t*>(bslot + min(r, last_row - 3));
                };
                auto border4 = [&](int r) -> uint4 {
                    return first_strip ? make_uint4(0, 0, 0, 0) : bin[(size_t)(min(r, last_row - 3) >> 2) * 64];
                };
                // a border dword relative to the bases, as a left operand: 16 * (D - base) in each half
                auto rel = [&](uint32_t w) -> uint32_t { return sa_pk_shl4_u16(sa_pk_sub_u16(w, base12)); };
                // every lane stores its direction rows, also lanes whose sequence 2 has ended (never read):
                // the stores of a block are then the same for every wave and the waits can be counted
                auto store_dirs = [&](int r) {
                    uint32_t lo[4], hi[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t w0 = e[2 * k], w1 = e[2 * k + 1];
                        lo[k] = __builtin_amdgcn_perm(w0, w1, 0x05040100u);     // low halves:  w0.lo : w1.lo
                        hi[k] = __builtin_amdgcn_perm(w0, w1, 0x07060302u);     // high halves: w0.hi : w1.hi
                    }
                    *reinterpret_cast<uint4*>(drowA + (size_t)r * 256) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                    *reinterpret_cast<uint4*>(drowB + (size_t)r * 256) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                };
                // the row's last cells (16 * (D - base) + tags in each half) as a border dword
                auto border_word = [&](uint32_t y) -> uint32_t { return sa_pk_add_u16((y >> SA_NB_SHIFT) & 0x0fff0fffu, base12); };

                uint32_t bres = residues4(0);
                uint4 bord = border4(0);
                uint32_t prevw = 0;                              // border dword of the row above the block
                // Every lane computes and stores every row, also lanes whose sequence 2 has ended (those rows are never
                // read): with the row under a per-lane condition the compiler keeps a second copy of the register row
                // for the lanes that skip it.  A lane takes its scores when its last row is done.
#define SA_LP_ROW(RR, DW, LW, B)                                                                                           \
    {                                                                                                                      \
        /* the row's end tests compare this copy of gend where they stand: hoisted out of the loop each of them holds a */  \
        /* pair of scalar registers (more than there are) and comes back as two vector instructions per group            */ \
        int gr_ = gend;                                                                                                    \
        asm volatile("" : "+s"(gr_));                                                                                      \
        fill_row_pair<SA_LP / 4, SA_LONG_PAIR_PROF_STRIDE, 0, WIDE>(R, gr_, (DW) | SA_NB_UP_B, (LW), prof + (B) * SA_LONG_PAIR_PROF_STRIDE, e);           \
    }                                                                                                                      \
    store_dirs(RR);                                                                                                        \
    if ((RR) + 1 == lenq) {                                                                                                \
        capA = (int)((pick_packed<SA_LP>(R, colA) & 0xffffu) >> SA_NB_SHIFT) + base1;                                      \
        capB = (int)(pick_packed<SA_LP>(R, colB) >> (16 + SA_NB_SHIFT)) + base2;                                           \
    }
                for (int r = 0; r < maxlen; r += 4) {
                    const uint32_t nres = residues4(r + 4);
                    const uint4 nbord = border4(r + 4);
                    const uint32_t rb0 = bres & 0xff, rb1 = (bres >> 8) & 0xff, rb2 = (bres >> 16) & 0xff, rb3 = bres >> 24;
                    if ((r & (int)P.rebase_mask) == 0 && r > 0 && !first_strip) {
                        // move the bases up to the border of the row above; the register row follows
                        const uint32_t d = sa_pk_sub_u16(prevw, base12);
                        const uint32_t dd = sa_pk_shl4_u16(d);
#pragma unroll
                        for (int c = 0; c < SA_LP; c++) R[c] -= dd;
                        base1 += (int)(d & 0xffffu); base2 += (int)(d >> 16);
                        base12 = prevw;
                    }
                    const uint32_t l0 = rel(prevw), l1 = rel(bord.x), l2 = rel(bord.y), l3 = rel(bord.z), l4 = rel(bord.w);
                    SA_LP_ROW(r, l0, l1, rb0)
                    const uint32_t y0 = R[SA_LP - 1];
                    SA_LP_ROW(r + 1, l1, l2, rb1)
                    const uint32_t y1 = R[SA_LP - 1];
                    SA_LP_ROW(r + 2, l2, l3, rb2)
                    const uint32_t y2 = R[SA_LP - 1];
                    SA_LP_ROW(r + 3, l3, l4, rb3)
                    const uint32_t y3 = R[SA_LP - 1];
                    if (more_strips)
                        bout[(size_t)(r >> 2) * 64] = make_uint4(border_word(y0), border_word(y1), border_word(y2), border_word(y3));
                    prevw = bord.w;
                    bres = nres;
                    bord = nbord;
                }
#undef SA_LP_ROW
                // the scores sit in the strips that hold each half's last column
                if (((lenp1 - 1) >> 6) == s) dfin1 = capA;
                if (has2 && ((lenp2 - 1) >> 6) == s) dfin2 = capB;
                __syncthreads();   // border column and direction rows are complete before anyone reads them
            }
            SA_PHASE(3);

            // --------------------------------------------------------- traceback: strip by strip from the right, both
            // halves side by side, eight rows per request (sa_long_walk.h).  A lane whose path is not in the strip, or
            // stands above the rows of a request, takes no part in it (and fetches nothing); a run of LEFT moves that
            // reaches a strip's left edge goes on in the next strip.
            {
                int wi[2] = {active ? lenq : 0, active2 ? lenq : 0};
                int wj[2] = {lenp1, lenp2};
                int wrun[2] = {0, 0}, wmatch[2] = {0, 0}, wgaps[2] = {0, 0};
                uint32_t wacc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                uint8_t* const rec0 = out_row1 + (size_t)(fiq - p1 - 1) * P.stride;
                uint8_t* const rec1 = out_row2 + (size_t)(fiq - p2 - 1) * P.stride;
#ifdef SA_LONG_NO_TB
                wi[0] = wi[1] = 0;       // experiment: the fill alone (records are wrong)
#endif
                const int r0 = 0;                                  // the rows on offer are the strip's rows
                const uint32_t voff = (uint32_t)lane * 16;
                const uint8_t* const resp = bslot;
                sa_u32x4v a0x, a1x, a2x, a3x, a4x, a5x, a6x, a7x, b0x, b1x, b2x, b3x, b4x, b5x, b6x, b7x;
                sa_u32x2v rx;
                for (int s = nstrips - 1; s >= 0; s--) {
                    const int cbase = 64 * s;
                    int tc0 = (wi[0] > 0 && wj[0] > cbase) ? wi[0] - 1 : -1;
                    int tc1 = (wi[1] > 0 && wj[1] > cbase) ? wi[1] - 1 : -1;
                    if (!__any(tc0 >= 0 || tc1 >= 0)) continue;
                    const uint8_t* const tile_a = reinterpret_cast<const uint8_t*>(dirs + (size_t)s * dir_rows * 256);
                    const uint8_t* const tile_b = tile_a + (size_t)half_dirs * 4;
                    const int tmax = wave_max_nonneg(max(tc0, tc1) + 1) - 1;
                    for (int ch = tmax >> 3; ch >= 0; --ch) {
                        if (!__any(tc0 >= 0 || tc1 >= 0)) break;
                        SA_LCK_FETCH(ch);
                        SA_LCK_ROWSTEP(ch, 7) SA_LCK_ROWSTEP(ch, 6) SA_LCK_ROWSTEP(ch, 5) SA_LCK_ROWSTEP(ch, 4)
                        SA_LCK_ROWSTEP(ch, 3) SA_LCK_ROWSTEP(ch, 2) SA_LCK_ROWSTEP(ch, 1) SA_LCK_ROWSTEP(ch, 0)
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
            }
            __syncthreads();
            SA_PHASE(4);
        }
    }
    if (timed && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) P.phase_cycles[(size_t)blockIdx.x * 8 + k] = tph[k];
    }
#undef SA_PHASE
}

template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_long_pair_kernel(const SaPepParams P) {
    sa_long_pair_body<false>(P);
}
// the same kernel with its maxima as unsigned 16-bit instructions: tables whose values leave the half-precision range
template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_long_pair_wide_kernel(const SaPepParams P) {
    sa_long_pair_body<true>(P);
}

extern "C" void sa_launch_long_pair(int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (wide)
        hipLaunchKernelGGL((sa_long_pair_wide_kernel<4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    else
        hipLaunchKernelGGL((sa_long_pair_kernel<4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
}
