// sa_kernels.hip -- Needleman-Wunsch fill + traceback for peptides (length <= 64)
// on gfx950, one pair per lane.
//
// What is computed (reference: include/seqalign.h:13-113):
//   H[i][j] = max(H[i-1][j-1] + S[a_j][b_i], H[i-1][j] + gap, H[i][j-1] + gap)
//   traceback from (len2,len1): DIAG if H == diag + S, else UP if H == up + gap,
//   else LEFT; UP on the left border, LEFT on the top border.
//
// How: the kernel works on D[i][j] = H[i][j] - (i+j)*gap.  In D the recurrence is
//   D[i][j] = max(D[i-1][j-1] + S'[a_j][b_i], D[i-1][j], D[i][j-1]),  S' = S - 2*gap
// with an all-zero border, and the two equality tests of the traceback become
// "D == diag + S'" and "D == up".  All of it is integer arithmetic, so results
// are identical to the reference; per cell it costs one add and one max3
// instead of three adds and two max.
//
// The DP row lives in VGPRs (64 columns, the sequence of the columns is
// right-aligned so the last column is register 63; padding columns score 0 and
// stay 0).  Two register banks alternate between rows so no copies are needed.
// Per cell two direction bits are produced from sign bits, (m-h)<0 and (up-h)<0,
// and shifted into per-row bit planes with v_alignbit.  Rows of direction bits
// go to LDS while they fit and to a per-workgroup global spill area after that.
// The traceback walks one row per step: the run of LEFT moves inside a row is
// a count-trailing-ones on (plane_nd & plane_nu).
//
// TRIANGULAR mode: all lanes of a wave share sequence 1 (file row p); its
// substitution scores are expanded once into an LDS "query profile"
// prof[b][c] = S'[a_c][b], so lane l fetches the scores of four columns with
// one ds_read_b128 indexed by its own row residue b.  Lanes hold 64 different
// second sequences taken from one length bucket.
// ADJACENT mode: lane l aligns file rows (k+l, k+l+1); scores are looked up in
// an LDS copy of S'.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sa_device.h"

#define SA_MODE_TRI 0
#define SA_MODE_ADJ 1

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// (hi:lo) >> sh, low 32 bits: shifting in the sign bit of lo with sh = 31
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t acc, int v) {
    return __builtin_amdgcn_alignbit(acc, (uint32_t)v, 31);
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// One DP row: IN = previous row, OUT = this row, columns 4*g0 .. 63.
// sc  = in: substitution scores of the first active group for this row
//       out: those of the next row (prefetched while the last group runs)
template <int MODE>
__device__ __forceinline__ void fill_row(const int (&IN)[SA_LP], int (&OUT)[SA_LP], const int g0,
                                         const int32_t* __restrict__ row_base,
                                         const int32_t* __restrict__ next_row_base,
                                         const int (&a4)[SA_LP], int4& sc,
                                         uint32_t& alo, uint32_t& ahi, uint32_t& blo, uint32_t& bhi) {
#pragma unroll
    for (int g = 0; g < SA_LP / 4; g++) {
        if (g >= g0) {   // wave-uniform
            int4 nx;
            if (MODE == SA_MODE_TRI) {
                // next group of this row, or the first active group of the next row
                const int32_t* src = (g == SA_LP / 4 - 1) ? next_row_base + 4 * g0 : row_base + 4 * (g + 1);
                nx = *reinterpret_cast<const int4*>(src);
            } else {
                if (g == SA_LP / 4 - 1) {
                    nx.x = nx.y = nx.z = nx.w = 0;   // reloaded by the caller for the next row
                } else {
                    const char* rb = reinterpret_cast<const char*>(row_base);
                    nx.x = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 4]);
                    nx.y = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 5]);
                    nx.z = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 6]);
                    nx.w = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 7]);
                }
            }
            const int s[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = 4 * g + k;
                const int diag = (c == 0) ? 0 : IN[c - 1];
                const int left = (c == 0) ? 0 : OUT[c - 1];
                const int up = IN[c];
                const int m = diag + s[k];
                const int h = max3i(m, up, left);
                if (c < 32) {
                    alo = shift_in_sign(alo, m - h);
                    blo = shift_in_sign(blo, up - h);
                } else {
                    ahi = shift_in_sign(ahi, m - h);
                    bhi = shift_in_sign(bhi, up - h);
                }
                OUT[c] = h;
            }
            sc = nx;
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(64, 2) sa_peptide_kernel(const SaPepParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    int32_t* const prof = reinterpret_cast<int32_t*>(smem);
    uint64_t* const emask = reinterpret_cast<uint64_t*>(smem + SA_LDS_PROF_BYTES);
    uint32_t* const stage = reinterpret_cast<uint32_t*>(smem + SA_LDS_PROF_BYTES + SA_LDS_EMASK_BYTES);
    uint32_t* const dirs = reinterpret_cast<uint32_t*>(smem + SA_LDS_FIXED_BYTES);
    const int lane = threadIdx.x;
    uint32_t* const spill = P.spill + (size_t)blockIdx.x * P.spill_rows * 256;
    const uint32_t cpr = P.stride >> 4;   // 16-byte chunks per record

    if (MODE == SA_MODE_ADJ) {
        // LDS copy of S' laid out [b][a] so that a row residue selects a 128-byte row
        for (int e = lane; e < 20 * 32; e += 64) {
            const int b = e >> 5, a = e & 31;
            prof[e] = (a <= 20) ? P.sprime[a * SA_SPRIME_STRIDE + b] : 0;
        }
        __syncthreads();
    }

    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(P.counter, 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= P.nitems) break;
        const SaItem item = P.items[it];

        int a4[SA_LP];          // ADJACENT: byte offset of column residue inside a table row
        int lenp_u = 0;         // TRIANGULAR: length of the shared first sequence
        uint8_t* out_row;       // record of canonical pair (row base + rel)
        uint32_t ntasks;
        if (MODE == SA_MODE_TRI) {
            const uint32_t p = item.p;
            lenp_u = P.len_fo[p];
            const int a = P.codes_fo[(size_t)p * SA_LP + lane];
            __syncthreads();   // previous item's readers are done with prof/emask
#pragma unroll
            for (int b = 0; b < 20; b++) prof[b * SA_LP + lane] = P.sprime[a * SA_SPRIME_STRIDE + b];
            prof[20 * SA_LP + lane] = 0;
#pragma unroll
            for (int b = 0; b < 20; b++) {
                const uint64_t mk = __ballot(a == b);
                if (lane == b) emask[b] = mk;
            }
            if (lane == 20) emask[20] = 0;
            __syncthreads();
            const uint64_t n = P.nseq;
            const uint64_t row_start = (uint64_t)p * (2 * n - p - 1) / 2;
            out_row = P.out + (row_start - P.out_base_pair) * P.stride;
            ntasks = item.count;
#pragma unroll
            for (int c = 0; c < SA_LP; c++) a4[c] = 0;
        } else {
            out_row = P.out + ((uint64_t)item.p - P.out_base_pair) * P.stride;
            ntasks = 1;
        }

        for (uint32_t task = 0; task < ntasks; task++) {
            // ---------------------------------------------------------- operands
            bool active;
            int lenq, lenp, maxlen, g0;
            uint32_t rel;                 // record index relative to out_row
            const uint8_t* brow;          // row residue r of this lane at brow[r * bstep]
            int bstep;
            const uint8_t* acol = nullptr;   // ADJACENT: slot of sequence 1 (for match counting)
            if (MODE == SA_MODE_TRI) {
                const uint32_t tile = item.tile_begin + task;
                const uint32_t q = tile * 64 + lane;
                const uint32_t fiq = P.tile_fi[q];
                active = fiq != SA_INVALID_FI && fiq > item.p;
                if (!__any(active)) continue;
                lenq = active ? P.tile_len[q] : 0;
                lenp = lenp_u;
                maxlen = P.tile_maxlen[tile];
                g0 = (SA_LP - lenp_u) >> 2;
                rel = fiq - item.p - 1;
                brow = P.tiles + (size_t)tile * 4096 + lane;
                bstep = 64;
            } else {
                active = (uint32_t)lane < item.count;
                const uint32_t s1 = item.p + (active ? lane : 0);
                const uint32_t s2 = s1 + 1;
                lenp = P.len_fo[s1];
                lenq = active ? P.len_fo[s2] : 0;
                maxlen = __builtin_amdgcn_readfirstlane(wave_max(lenq));
                g0 = __builtin_amdgcn_readfirstlane(wave_min(SA_LP - lenp)) >> 2;
                rel = lane;
                acol = P.codes_fo + (size_t)s1 * SA_LP;
                brow = P.codes_fo + (size_t)s2 * SA_LP + (SA_LP - P.len_fo[s2]);
                bstep = 1;
                const uint32_t* aw = reinterpret_cast<const uint32_t*>(acol);
#pragma unroll
                for (int w = 0; w < SA_LP / 4; w++) {
                    const uint32_t v = aw[w];
                    a4[4 * w + 0] = (v & 0xff) << 2;
                    a4[4 * w + 1] = ((v >> 8) & 0xff) << 2;
                    a4[4 * w + 2] = ((v >> 16) & 0xff) << 2;
                    a4[4 * w + 3] = (v >> 24) << 2;
                }
            }
            const bool wide = g0 < 8;                       // columns 0..31 in use: four plane words per row
            const int wpr = wide ? 4 : 2;                   // dwords per direction row per lane
            const int rows_lds = (int)(P.lds_dir_bytes / (uint32_t)(wpr * 256));

            // -------------------------------------------------------------- fill
            int R[SA_LP], Q[SA_LP];
#pragma unroll
            for (int c = 0; c < SA_LP; c++) { R[c] = 0; Q[c] = 0; }
            uint32_t alo = 0, ahi = 0, blo = 0, bhi = 0;

            int b0 = brow[0];                               // residue of row 0 (pad code if none)
            int b1 = (1 < maxlen) ? brow[bstep] : SA_PAD_CODE;
            if (MODE == SA_MODE_TRI) { if (!active) { b0 = SA_PAD_CODE; b1 = SA_PAD_CODE; } }
            else { if (b0 > SA_PAD_CODE) b0 = SA_PAD_CODE; if (b1 > SA_PAD_CODE) b1 = SA_PAD_CODE; }
            int4 sc;
            auto first_scores = [&](int b) {
                if (MODE == SA_MODE_TRI) {
                    sc = *reinterpret_cast<const int4*>(prof + b * SA_LP + 4 * g0);
                } else {
                    const char* rb = reinterpret_cast<const char*>(prof + (b < 20 ? b : 0) * 32);
                    // first active group is wave-uniform, the register index must be static
#pragma unroll
                    for (int g = 0; g < SA_LP / 4; g++) {
                        if (g == g0) {
                            sc.x = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 0]);
                            sc.y = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 1]);
                            sc.z = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 2]);
                            sc.w = *reinterpret_cast<const int32_t*>(rb + a4[4 * g + 3]);
                        }
                    }
                }
            };
            first_scores(b0);

            auto store_dirs = [&](int r) {
                if (r < rows_lds) {
                    if (wide) *reinterpret_cast<uint4*>(dirs + (r * 64 + lane) * 4) = make_uint4(alo, ahi, blo, bhi);
                    else *reinterpret_cast<uint2*>(dirs + (r * 64 + lane) * 2) = make_uint2(ahi, bhi);
                } else {
                    const int rr = r - rows_lds;
                    if (wide) *reinterpret_cast<uint4*>(spill + (rr * 64 + lane) * 4) = make_uint4(alo, ahi, blo, bhi);
                    else *reinterpret_cast<uint2*>(spill + (rr * 64 + lane) * 2) = make_uint2(ahi, bhi);
                }
            };

            for (int r = 0; r < maxlen; r += 2) {
                // residues two rows ahead (latency hidden behind two rows of arithmetic)
                int b2 = (r + 2 < maxlen) ? brow[(r + 2) * bstep] : SA_PAD_CODE;
                int b3 = (r + 3 < maxlen) ? brow[(r + 3) * bstep] : SA_PAD_CODE;
                if (MODE == SA_MODE_ADJ) { if (b2 > SA_PAD_CODE) b2 = SA_PAD_CODE; if (b3 > SA_PAD_CODE) b3 = SA_PAD_CODE; }
                const int32_t* base0;
                const int32_t* base1;
                const int32_t* base2;
                if (MODE == SA_MODE_TRI) {
                    base0 = prof + b0 * SA_LP; base1 = prof + b1 * SA_LP; base2 = prof + b2 * SA_LP;
                } else {
                    base0 = prof + (b0 < 20 ? b0 : 0) * 32;
                    base1 = prof + (b1 < 20 ? b1 : 0) * 32;
                    base2 = prof + (b2 < 20 ? b2 : 0) * 32;
                }
                if (r < lenq) {
                    fill_row<MODE>(R, Q, g0, base0, base1, a4, sc, alo, ahi, blo, bhi);
                    store_dirs(r);
                }
                if (MODE == SA_MODE_ADJ) first_scores(b1);
                if (r + 1 < lenq) {
                    fill_row<MODE>(Q, R, g0, base1, base2, a4, sc, alo, ahi, blo, bhi);
                    store_dirs(r + 1);
                }
                if (MODE == SA_MODE_ADJ) first_scores(b2);
                b0 = b2; b1 = b3;
            }
            const int dlast = (lenq & 1) ? Q[SA_LP - 1] : R[SA_LP - 1];
            const int score = dlast + (lenp + lenq) * P.gap;
            __syncthreads();   // direction rows written by this wave are visible to its loads

            // --------------------------------------------------------- traceback
            // One step per row, bottom row first.  The direction words of row r-1,
            // the row residue of row r-2 and the equality mask of row r-1 are
            // requested while row r is being processed.
            int j = lenp, matches = 0, gaps = 0;
            uint32_t acc = 0;
            const int padl = SA_LP - lenp;
            auto load_dirs = [&](int r) -> uint4 {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r < rows_lds) {
                    if (wide) v = *reinterpret_cast<const uint4*>(dirs + (r * 64 + lane) * 4);
                    else { const uint2 t2 = *reinterpret_cast<const uint2*>(dirs + (r * 64 + lane) * 2); v.y = t2.x; v.w = t2.y; }
                } else {
                    const int rr = r - rows_lds;
                    if (wide) v = *reinterpret_cast<const uint4*>(spill + (rr * 64 + lane) * 4);
                    else { const uint2 t2 = *reinterpret_cast<const uint2*>(spill + (rr * 64 + lane) * 2); v.y = t2.x; v.w = t2.y; }
                }
                return v;
            };
            auto load_res = [&](int r) -> int {
                int b = (r >= 0) ? (int)brow[r * bstep] : SA_PAD_CODE;
                return b > SA_PAD_CODE ? SA_PAD_CODE : b;
            };
            uint4 dcur = load_dirs(maxlen - 1);
            int bcur = load_res(maxlen - 1);
            int bnxt = load_res(maxlen - 2);
            uint64_t ecur = (MODE == SA_MODE_TRI) ? emask[bcur] : 0;
            for (int r = maxlen - 1; r >= 0; --r) {
                uint4 dnxt = make_uint4(0, 0, 0, 0);
                if (r > 0) dnxt = load_dirs(r - 1);
                const int bnn = load_res(r - 2);
                const uint64_t enxt = (MODE == SA_MODE_TRI) ? emask[bnxt] : 0;
                if (r < lenq) {
                    const uint64_t A = ((uint64_t)dcur.x << 32) | dcur.y;   // bit (63-c): cell != diag + S'
                    const uint64_t B = ((uint64_t)dcur.z << 32) | dcur.w;   // bit (63-c): cell != up
                    int run = 0;
                    if (j > 0) {
                        const uint64_t x = ~((A & B) >> (lenp - j));        // zero bit = first non-LEFT column
                        run = x ? __builtin_ctzll(x) : 64;
                        run = min(run, j);
                        j -= run;
                    }
                    int t = 1;                                              // left border: UP
                    if (j > 0) t = (int)((A >> (lenp - j)) & 1);            // 0 = DIAG, 1 = UP
                    if (t == 0) {
                        const int c = padl + j - 1;
                        if (MODE == SA_MODE_TRI) matches += (int)((ecur >> c) & 1);
                        else matches += ((int)acol[c] == bcur) ? 1 : 0;
                        j--;
                    }
                    gaps += t;
                    const int k = lenq - 1 - r;
                    acc |= (uint32_t)(run * 2 + t) << (8 * (k & 3));
                    if ((k & 3) == 3 || r == 0) {
                        stage[lane * SA_LDS_STAGE_DWORDS + 2 + (k >> 2)] = acc;
                        acc = 0;
                    }
                }
                dcur = dnxt; bcur = bnxt; bnxt = bnn; ecur = enxt;
            }
            stage[lane * SA_LDS_STAGE_DWORDS + 0] = (uint32_t)score;
            stage[lane * SA_LDS_STAGE_DWORDS + 1] = (uint32_t)matches | ((uint32_t)gaps << 16);
            __syncthreads();

            // ------------------------------------------- records, 16 B per lane store
            for (uint32_t id = lane; id < 64 * cpr; id += 64) {
                const uint32_t rec = id / cpr, ch = id - rec * cpr;
                const uint32_t rrel = __shfl(rel, rec, 64);
                const int ract = __shfl((int)active, rec, 64);
                if (ract) {
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + rec * SA_LDS_STAGE_DWORDS + ch * 4);
                    *reinterpret_cast<uint4*>(out_row + (uint64_t)rrel * P.stride + ch * 16) = v;
                }
            }
            __syncthreads();   // staging area is reused by the next task
        }
    }
}

extern "C" void sa_launch_peptide(int mode, const SaPepParams* p, unsigned grid, unsigned lds_bytes,
                                  hipStream_t stream) {
    if (mode == SA_MODE_TRI)
        hipLaunchKernelGGL(sa_peptide_kernel<SA_MODE_TRI>, dim3(grid), dim3(64), lds_bytes, stream, *p);
    else
        hipLaunchKernelGGL(sa_peptide_kernel<SA_MODE_ADJ>, dim3(grid), dim3(64), lds_bytes, stream, *p);
}
