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
    const int lane = threadIdx.x;
    const uint32_t rows_cap = P.lslot;
    const uint32_t strips_cap = P.lslot >> 6;
    uint32_t* const dirs = P.spill + (size_t)blockIdx.x * strips_cap * rows_cap * 256;
    int32_t* const bcol = P.bcol + (size_t)blockIdx.x * 2 * rows_cap * 64;

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

        int lenp_u = 0;
        uint8_t* out_row;
        uint32_t ntasks;
        if (MODE == SA_MODE_TRI) {
            lenp_u = P.len_fo[item.p];
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
                acol = P.codes_fo + (size_t)item.p * P.lslot;
            } else {
                active = (uint32_t)lane < item.count;
                const uint32_t s1 = item.p + (active ? lane : 0);
                const uint32_t s2 = s1 + 1;
                lenp = P.len_fo[s1];
                lenq = active ? P.len_fo[s2] : 0;
                maxlenp = __builtin_amdgcn_readfirstlane(wave_max(active ? lenp : 1));
                rel = lane;
                acol = P.codes_fo + (size_t)s1 * P.lslot;
                brow = P.codes_fo + (size_t)s2 * P.lslot;
                bstep = 1;
            }
            maxlen = __builtin_amdgcn_readfirstlane(wave_max(lenq));
            const int nstrips = (maxlenp + 63) >> 6;
            const int gend_last = ((maxlenp - 64 * (nstrips - 1)) + 3) >> 2;
            const int last_row = (int)rows_cap - 1;

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
                const int32_t* bin = bcol + (size_t)(s & 1) * rows_cap * 64 + lane;
                int32_t* bout = bcol + (size_t)((s + 1) & 1) * rows_cap * 64 + lane;
                uint32_t* drow = dirs + ((size_t)s * rows_cap * 64 + lane) * 4;
                const bool first_strip = s == 0, more_strips = s + 1 < nstrips;
                // the last dword a row writes took 4*gend - 16*(wl) cells: its first column is moved down to bit 0
                const int wl = (gend - 1) >> 2;
                const int lsh = 32 - 2 * (4 * gend - 16 * wl);
                const int sh0 = wl == 0 ? lsh : 0, sh1 = wl == 1 ? lsh : 0, sh2 = wl == 2 ? lsh : 0, sh3 = wl == 3 ? lsh : 0;

                auto residue = [&](int r) -> int { return (int)brow[(size_t)min(r, last_row) * bstep]; };
                // last column of the previous strip in bank form (4*D + 1); 1 on the matrix border
                auto border = [&](int r) -> int { return first_strip ? 1 : bin[(size_t)min(r, last_row) * 64]; };
                int b0 = residue(0), b1 = residue(1);
                int l0 = border(0), l1 = border(1);
                int dprev = 1;                                   // border value of the previous row = diag of column 0
                int4 sc0 = make_int4(0, 0, 0, 0), sc1 = sc0;
                if (MODE == SA_MODE_TRI) sc0 = *reinterpret_cast<const int4*>(prof + b0 * SA_PROF_STRIDE);
                for (int r = 0; r < maxlen; r += 2) {
                    const int b2 = residue(r + 2), b3 = residue(r + 3);
                    const int l2 = border(r + 2), l3 = border(r + 3);
                    if (r < lenq) {
                        if (MODE == SA_MODE_TRI)
                            fill_row_tri_tag(R, Q, gend, dprev, l0 - 1, prof + b0 * SA_PROF_STRIDE, prof + b1 * SA_PROF_STRIDE, sc0, sc1,
                                             e0, e1, e2, e3);
                        else
                            fill_row_adj_tag(R, Q, gend, dprev, l0 - 1, prof + (b0 < 20 ? b0 : 0) * 32, a4, e0, e1, e2, e3);
                        *reinterpret_cast<uint4*>(drow + (size_t)r * 256) = make_uint4(e0 >> sh0, e1 >> sh1, e2 >> sh2, e3 >> sh3);
                        if (more_strips) bout[(size_t)r * 64] = Q[SA_LP - 1];
                    }
                    if (r + 1 < lenq) {
                        if (MODE == SA_MODE_TRI)
                            fill_row_tri_tag(Q, R, gend, l0, l1 - 1, prof + b1 * SA_PROF_STRIDE, prof + b2 * SA_PROF_STRIDE, sc1, sc0,
                                             e0, e1, e2, e3);
                        else
                            fill_row_adj_tag(Q, R, gend, l0, l1 - 1, prof + (b1 < 20 ? b1 : 0) * 32, a4, e0, e1, e2, e3);
                        *reinterpret_cast<uint4*>(drow + (size_t)(r + 1) * 256) = make_uint4(e0 >> sh0, e1 >> sh1, e2 >> sh2, e3 >> sh3);
                        if (more_strips) bout[(size_t)(r + 1) * 64] = R[SA_LP - 1];
                    }
                    dprev = l1;
                    b0 = b2; b1 = b3; l0 = l2; l1 = l3;
                }
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

            // --------------------------------------------------------- traceback
            uint8_t* rec = out_row + (size_t)rel * P.stride;
            int j = lenp, matches = 0, gaps = 0;
            for (int ch = (maxlen - 1) >> 3; ch >= 0; --ch) {
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 7; k >= 0; k--) {
                    const int r = 8 * ch + k;
                    if (r < lenq) {
                        int run_total = 0, t = 1;
                        while (j > 0) {
                            // tags of the strip column j-1 is in; L = leading zeros from that column down =
                            // 2 * (LEFT moves) + (1 if an UP ends the run, 0 if a DIAG does)
                            const int sj = (j - 1) >> 6, jl = ((j - 1) & 63) + 1;
                            const uint4 d = *reinterpret_cast<const uint4*>(dirs + (((size_t)sj * rows_cap + r) * 64 + lane) * 4);
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
                            matches += (acol[j - 1] == brow[(size_t)r * bstep]) ? 1 : 0;
                            j--;
                        }
                        gaps += t;
                        w[k >> 1] |= (uint32_t)(run_total * 2 + t) << (16 * (k & 1));
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
        }
    }
}

extern "C" void sa_launch_long(int mode, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (mode == SA_MODE_TRI)
        hipLaunchKernelGGL((sa_long_kernel<SA_MODE_TRI, 2>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    else
        hipLaunchKernelGGL((sa_long_kernel<SA_MODE_ADJ, 2>), dim3(grid), dim3(64), lds_bytes, stream, *p);
}
