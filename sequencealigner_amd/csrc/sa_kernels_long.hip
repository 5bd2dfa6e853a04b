// sa_kernels_long.hip -- Needleman-Wunsch fill + traceback for sequences longer
// than 64 residues (proteins) on gfx950, still one pair per lane.
//
// Same arithmetic as the peptide kernel (see sa_kernels.hip: D = H - (i+j)*gap,
// tagged cells of five instructions, sa_kernel_common.h).  The matrix is
// swept in vertical strips of 64 columns: inside a strip the DP row lives in
// two alternating banks of 64 VGPRs exactly as in the peptide kernel; the last
// column of a strip is written per row to a small global buffer and read back
// as the left border of the next strip.  Direction tags (16 bytes per lane, row
// and strip: column c of the strip at bits 2(c%16) of dword c/16) go to a
// per-workgroup global area, 1 KiB per wave and row,
// coalesced.  The traceback reads, for every row, the 16 bytes of the strip its
// column is in; a run of LEFT moves may continue into the strip on the left.
// The per-row output is a uint16 (LEFT moves * 2 + DIAG/UP).
//
// Why not one workgroup per pair sweeping anti-diagonals: the pair space of an
// all-vs-all run is huge (5e7 pairs for 10 000 proteins), so lanes are kept
// independent; every lane then runs its own traceback instead of 63 lanes
// waiting for one, and no cross-lane exchange is needed in the fill.
#include "sa_kernel_common.h"

template <int MODE, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_long_kernel(const SaPepParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    int32_t* const prof = reinterpret_cast<int32_t*>(smem);
    uint8_t* const pseq = smem + SA_LDS_PROF_BYTES + 64;      // TRIANGULAR: the shared first sequence, lslot bytes
    const int lane = threadIdx.x;
    const uint32_t rows_cap = P.lslot;
    const uint32_t strips_cap = P.lslot >> 6;
    // direction rows: as many per strip as the longest uploaded sequence needs (a multiple of 8), not the slot size
    const uint32_t dir_rows = P.spill_rows;
    uint32_t* const dirs = P.spill + (size_t)blockIdx.x * strips_cap * dir_rows * 256;
    int32_t* const bcol = P.bcol + (size_t)blockIdx.x * 2 * rows_cap * 64;
    // optional phase timers: 0 fetch, 2 operands, 3 fill, 4 traceback + record
    const bool timed = P.phase_cycles != nullptr;
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tmark = timed ? __builtin_amdgcn_s_memtime() : 0;
#define SA_PHASE(k) do { if (timed) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tmark; tmark = t_; } } while (0)

    if (MODE == SA_MODE_ADJ) {
        for (int e = lane; e < 20 * 32; e += 64) {
            const int b = e >> 5, a = e & 31;
            prof[e] = SA_TAG_TABLE((a <= 20) ? P.sprime[a * SA_SPRIME_STRIDE + b] : 0);
        }
        __syncthreads();
    }

    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(P.counter, 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= P.nitems) break;
        const SaItem item = P.items[it];
        SA_PHASE(0);

        int lenp_u = 0;
        uint8_t* out_row;
        uint32_t ntasks;
        if (MODE == SA_MODE_TRI) {
            lenp_u = P.len_fo[item.p];
            __syncthreads();      // the previous item's walks are done with the copy
            for (uint32_t o = lane * 16; o < P.lslot; o += 64 * 16)
                *reinterpret_cast<uint4*>(pseq + o) = *reinterpret_cast<const uint4*>(P.codes_fo + (size_t)item.p * P.lslot + o);
            __syncthreads();
            const uint64_t n = P.nseq;
            const uint64_t row_start = (uint64_t)item.p * (2 * n - item.p - 1) / 2;
            out_row = P.out + (row_start - P.out_base_pair) * P.stride;
            ntasks = item.count;
        } else {
            out_row = P.out + ((uint64_t)item.p - P.out_base_pair) * P.stride;
            ntasks = 1;
        }

        for (uint32_t task = 0; task < ntasks; task++) {
            // ---------------------------------------------------------- operands
            bool active;
            int lenq, lenp, maxlen, maxlenp;
            uint32_t rel;
            const uint8_t* brow;          // row residue r of this lane at brow[r * bstep]
            size_t bstep;
            const uint8_t* acol;          // column residues of this lane's sequence 1
            const uint8_t* bslot;         // sequence 2 as one padded slot of lslot bytes
            if (MODE == SA_MODE_TRI) {
                const uint32_t vbase = item.v0 + task * 64;
                const uint32_t vl = vbase + lane;
                const uint32_t nvalid = P.nseq - 1 - item.p;
                active = vl < nvalid;
                if (!__any(active)) continue;
                uint32_t sidx = 0, cum = item.c0;
                const uint32_t* cur_row = P.cursor + (size_t)item.p * P.nb;
                for (uint32_t b = item.b0; b < P.nb; b++) {
                    const uint32_t first = P.bucket_base[b] + cur_row[b];
                    const uint32_t cnt = P.bucket_base[b + 1] - first;
                    if (vl >= cum && vl < cum + cnt) sidx = first + (vl - cum);
                    cum += cnt;
                    if (cum > vbase + 63) break;
                }
                const uint32_t fiq = P.fi_sorted[sidx];
                lenq = active ? P.len_sorted[sidx] : 0;
                lenp = lenp_u;
                maxlenp = lenp_u;
                rel = fiq - item.p - 1;
                brow = P.codes_t + sidx;
                bstep = P.npad;
                bslot = P.codes_fo + (size_t)fiq * P.lslot;
                acol = P.codes_fo + (size_t)item.p * P.lslot;
            } else {
                active = (uint32_t)lane < item.count;
                const uint32_t s1 = item.p + (active ? lane : 0);
                const uint32_t s2 = s1 + 1;
                lenp = P.len_fo[s1];
                lenq = active ? P.len_fo[s2] : 0;
                maxlenp = wave_max_nonneg(active ? lenp : 1);
                rel = lane;
                acol = P.codes_fo + (size_t)s1 * P.lslot;
                brow = P.codes_fo + (size_t)s2 * P.lslot;
                bslot = brow;
                bstep = 1;
            }
            maxlen = wave_max_nonneg(lenq);
            const int nstrips = (maxlenp + 63) >> 6;
            const int gend_last = ((maxlenp - 64 * (nstrips - 1)) + 3) >> 2;
            const int last_row = (int)rows_cap - 1;
            SA_PHASE(2);

            // -------------------------------------------------------------- fill
            int dfinal = 0;
            for (int s = 0; s < nstrips; s++) {
                const int gend = (s == nstrips - 1) ? gend_last : SA_LP / 4;
                int a4[SA_LP];
                if (MODE == SA_MODE_TRI) {
                    // query profile of columns 64s .. 64s+63 of the shared first sequence
                    const int a = acol[64 * s + lane];
                    __syncthreads();
#pragma unroll
                    for (int b = 0; b < 20; b++) prof[b * SA_PROF_STRIDE + lane] = SA_TAG_TABLE(P.sprime[a * SA_SPRIME_STRIDE + b]);
                    prof[20 * SA_PROF_STRIDE + lane] = SA_TAG_TABLE(0);
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < SA_LP; c++) a4[c] = 0;
                } else {
                    const uint32_t* aw = reinterpret_cast<const uint32_t*>(acol + 64 * s);
#pragma unroll
                    for (int w = 0; w < SA_LP / 4; w++) {
                        const uint32_t v = aw[w];
                        a4[4 * w + 0] = (v & 0xff) << 2;
                        a4[4 * w + 1] = ((v >> 8) & 0xff) << 2;
                        a4[4 * w + 2] = ((v >> 16) & 0xff) << 2;
                        a4[4 * w + 3] = (v >> 24) << 2;
                    }
                }
                int R[SA_LP], Q[SA_LP];
#pragma unroll
                for (int c = 0; c < SA_LP; c++) { R[c] = 1; Q[c] = 1; }     // border row: 4*0 + 1
                uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
                // border column: four rows per lane and 16 bytes, [strip parity][row / 4][lane][row % 4]
                const int4* bin = reinterpret_cast<const int4*>(bcol + (size_t)(s & 1) * rows_cap * 64) + lane;
                int4* bout = reinterpret_cast<int4*>(bcol + (size_t)((s + 1) & 1) * rows_cap * 64) + lane;
                uint32_t* drow = dirs + ((size_t)s * dir_rows * 64 + lane) * 4;
                const bool first_strip = s == 0, more_strips = s + 1 < nstrips;
                // the last dword a row writes took 4*gend - 16*(wl) cells: its first column is moved down to bit 0
                const int wl = (gend - 1) >> 2;
                const int lsh = 32 - 2 * (4 * gend - 16 * wl);
                const int sh0 = wl == 0 ? lsh : 0, sh1 = wl == 1 ? lsh : 0, sh2 = wl == 2 ? lsh : 0, sh3 = wl == 3 ? lsh : 0;

                // Rows go four at a time.  The four row residues (one dword of sequence 2's slot) and the four
                // border values (last column of the previous strip in bank form 4*D + 1; 1 on the matrix border)
                // of the next block are requested at the top of a block, four rows of arithmetic before they
                // are needed, and the block's own last-column values leave in one 16-byte store.  The direction
                // rows are stored by every lane, also by lanes whose sequence 2 has ended (their rows are never
                // read): the number of stores between a request and its use is then the same for every wave, and
                // the wait before the use can leave those stores in flight.
                auto residues4 = [&](int r) -> uint32_t {
                    return *reinterpret_cast<const uint32_t*>(bslot + min(r, last_row - 3));
                };
                auto border4 = [&](int r) -> int4 {
                    return first_strip ? make_int4(1, 1, 1, 1) : bin[(size_t)(min(r, last_row - 3) >> 2) * 64];
                };
                auto prow = [&](uint32_t b) -> const int32_t* {
                    return (MODE == SA_MODE_TRI) ? prof + b * SA_PROF_STRIDE : prof + (b < 20 ? b : 0) * 32;
                };
                uint32_t bres = residues4(0);
                int4 bord = border4(0);
                int dprev = 1;                                   // border value of the previous row = diag of column 0
                int4 sc0 = make_int4(0, 0, 0, 0), sc1 = sc0;
                if (MODE == SA_MODE_TRI) sc0 = *reinterpret_cast<const int4*>(prof + (bres & 0xff) * SA_PROF_STRIDE);
#define SA_LONG_ROW(IN, OUT, RR, DIAG0, LEFTY, B, BNEXT, SCI, SCO)                                                              \
    if ((RR) < lenq) {                                                                                                           \
        if (MODE == SA_MODE_TRI)                                                                                                 \
            fill_row_tri_tag2(IN, OUT, gend, DIAG0, (LEFTY) - 1, prow(B), prow(BNEXT), SCI, SCO, e0, e1, e2, e3);                \
        else                                                                                                                     \
            fill_row_adj_tag(IN, OUT, gend, DIAG0, (LEFTY) - 1, prow(B), a4, e0, e1, e2, e3);                                   \
    }                                                                                                                            \
    *reinterpret_cast<uint4*>(drow + (size_t)(RR) * 256) = make_uint4(e0 >> sh0, e1 >> sh1, e2 >> sh2, e3 >> sh3);
                for (int r = 0; r < maxlen; r += 4) {
                    const uint32_t nres = residues4(r + 4);
                    const int4 nbord = border4(r + 4);
                    const uint32_t rb0 = bres & 0xff, rb1 = (bres >> 8) & 0xff, rb2 = (bres >> 16) & 0xff, rb3 = bres >> 24;
                    int4 lastc;
                    SA_LONG_ROW(R, Q, r, dprev, bord.x, rb0, rb1, sc0, sc1)
                    lastc.x = Q[SA_LP - 1];
                    SA_LONG_ROW(Q, R, r + 1, bord.x, bord.y, rb1, rb2, sc1, sc0)
                    lastc.y = R[SA_LP - 1];
                    SA_LONG_ROW(R, Q, r + 2, bord.y, bord.z, rb2, rb3, sc0, sc1)
                    lastc.z = Q[SA_LP - 1];
                    SA_LONG_ROW(Q, R, r + 3, bord.z, bord.w, rb3, (nres & 0xff), sc1, sc0)
                    lastc.w = R[SA_LP - 1];
                    if (more_strips) bout[(size_t)(r >> 2) * 64] = lastc;
                    dprev = bord.w;
                    bres = nres;
                    bord = nbord;
                }
#undef SA_LONG_ROW
                // the score sits in the strip that holds this lane's last column
                if (((lenp - 1) >> 6) == s) {
                    const int col = (lenp - 1) & 63;
                    if (MODE == SA_MODE_TRI) {
                        const int dq = pick_column(Q, col), dr = pick_column(R, col);
                        dfinal = (lenq & 1) ? dq : dr;
                    } else {
#pragma unroll
                        for (int c = 0; c < SA_LP; c++)
                            if (c == col) dfinal = (lenq & 1) ? Q[c] : R[c];
                    }
                }
                __syncthreads();   // border column and direction rows are complete before anyone reads them
            }
            const int score = (dfinal >> 2) + (lenp + lenq) * P.gap;      // banks hold 4*D + 1
            SA_PHASE(3);

            // --------------------------------------------------------- traceback
            // One row per step, bottom row first, eight rows per 16-byte piece of the record.  A row needs the
            // 16 bytes of direction tags of the strip its column is in, which depends on where the walk stands.
            // TRIANGULAR: the tags are requested eight rows ahead for the strip the walk is in at that moment
            // (it changes strip once in 64 columns; a wrong guess is loaded again when it is used), the eight
            // row residues come as two dwords a piece ahead, and the shared first sequence is read from LDS.
            uint8_t* rec = out_row + (size_t)rel * P.stride;
            int j = lenp, matches = 0, gaps = 0;
            auto load_tags = [&](int sj, int r) -> uint4 {
                return *reinterpret_cast<const uint4*>(dirs + (((size_t)sj * dir_rows + min(r, (int)dir_rows - 1)) * 64 + lane) * 4);
            };
            constexpr bool AHEAD = MODE == SA_MODE_TRI;
            uint4 ring[AHEAD ? 8 : 1];
            int ring_s[AHEAD ? 8 : 1];
            uint2 resq = make_uint2(0, 0);
            const int ch0 = (maxlen - 1) >> 3;
            if (AHEAD) {
                const int s0 = (lenp - 1) >> 6;
#pragma unroll
                for (int k = 0; k < 8; k++) { ring[k] = load_tags(s0, 8 * ch0 + k); ring_s[k] = s0; }
                resq = *reinterpret_cast<const uint2*>(bslot + min(8 * ch0, last_row - 7));
            }
            for (int ch = ch0; ch >= 0; --ch) {
                uint32_t w[4] = {0, 0, 0, 0};
                const uint2 res = resq;
                if (AHEAD && ch > 0) resq = *reinterpret_cast<const uint2*>(bslot + 8 * (ch - 1));
#pragma unroll
                for (int k = 7; k >= 0; k--) {
                    const int r = 8 * ch + k;
                    if (r < lenq) {
                        int run_total = 0, t = 1;
                        bool first = true;
                        while (j > 0) {
                            // tags of the strip column j-1 is in; L = leading zeros from that column down =
                            // 2 * (LEFT moves) + (1 if an UP ends the run, 0 if a DIAG does)
                            const int sj = (j - 1) >> 6, jl = ((j - 1) & 63) + 1;
                            uint4 d;
                            if (AHEAD && first && ring_s[AHEAD ? k : 0] == sj) d = ring[AHEAD ? k : 0];
                            else d = load_tags(sj, r);
                            first = false;
                            const uint64_t v0 = ((uint64_t)d.y << 32) | d.x;
                            const uint64_t v1 = ((uint64_t)d.w << 32) | d.z;
                            const bool upper = jl > 32;
                            const int jj = upper ? jl - 32 : jl;
                            const uint64_t x = (upper ? v1 : v0) << ((64 - 2 * jj) & 63);
                            const int lz0 = v0 ? __builtin_clzll(v0) : 64;
                            const int L = x ? __builtin_clzll(x) : (upper ? 2 * jj + lz0 : 64);
                            const int run = min(L >> 1, jl);
                            j -= run;
                            run_total += run;
                            if (run == jl && j > 0) continue;                     // the run goes on in the strip to the left
                            if (run < jl) t = L & 1;
                            break;
                        }
                        if (t == 0) {
                            uint32_t a, b;
                            if (AHEAD) {
                                a = pseq[j - 1];
                                b = ((k < 4 ? res.x : res.y) >> (8 * (k & 3))) & 0xff;
                            } else {
                                a = acol[j - 1];
                                b = brow[(size_t)r * bstep];
                            }
                            matches += (a == b) ? 1 : 0;
                            j--;
                        }
                        gaps += t;
                        w[k >> 1] |= (uint32_t)(run_total * 2 + t) << (16 * (k & 1));
                    }
                    if (AHEAD && ch > 0) {
                        // the same row of the next piece, for the strip the walk is in now
                        const int sn = (max(j, 1) - 1) >> 6;
                        ring[AHEAD ? k : 0] = load_tags(sn, r - 8);
                        ring_s[AHEAD ? k : 0] = sn;
                    }
                }
                if (active)
                    *reinterpret_cast<uint4*>(rec + SA_LONG_HEADER_BYTES + 16 * ch) = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if (active) {
                *reinterpret_cast<uint4*>(rec) = make_uint4((uint32_t)score, (uint32_t)matches, (uint32_t)gaps, 0u);
                // rows past the last chunk are zero, so that equal inputs give equal records
                for (uint32_t o = SA_LONG_HEADER_BYTES + 16 * (((maxlen - 1) >> 3) + 1); o < P.stride; o += 16)
                    *reinterpret_cast<uint4*>(rec + o) = make_uint4(0, 0, 0, 0);
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

extern "C" void sa_launch_long(int mode, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (mode == SA_MODE_TRI)
        hipLaunchKernelGGL((sa_long_kernel<SA_MODE_TRI, 2>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    else
        hipLaunchKernelGGL((sa_long_kernel<SA_MODE_ADJ, 2>), dim3(grid), dim3(64), lds_bytes, stream, *p);
}
