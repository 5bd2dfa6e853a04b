// sa_kernels_pair.hip -- TRIANGULAR peptides, two first sequences per wave.
//
// Same recurrence as sa_kernels.hip, but every VGPR of the DP row holds two cells: the low
// 16 bits belong to the pair (p1, q), the high 16 bits to (p2, q), where q is the lane's
// second sequence and p1 < p2 are two first sequences of the same length bucket.  A lane's
// row residue b is the same for both matrices, so one LDS lookup
// prof[b][c] = T(p1)[b][c] | T(p2)[b][c] << 16 feeds both.  Cell format, instruction
// blocks and the in-place register row: sa_pair_common.h.  About 2.4 instructions per cell
// against 5 for the unpaired kernel.
//
// Tags: 0 LEFT, 2 UP, 3 DIAG, so that with the newest column in the low bits of a direction
// word a count-trailing-zeros finds a run of LEFT moves (sa_traceback.h, traceback_pk).
//
// The host uses this kernel only when every half provably stays in range: S' >= 1 and
// longest sequence * max S' <= 1983 (BLOSUM50 with gap -4: S' = 3 .. 23).
//
// The lanes of a task are 64 consecutive entries of p1's valid list; the pair
// (p2, q) exists only where q's file index is above p2 (all but the few sequences
// between p1 and p2), elsewhere the high half is computed and dropped.
#include "sa_pair_common.h"
#include "sa_traceback.h"

// WIDE: the three-way maxima as two unsigned 16-bit maxima each (sa_pair_common.h): longest sequence * max S' <= 4095.
// LONGREC: the kernel runs the tiles of peptides inside a set that also holds sequences of more than 64 residues: the
// file-order slots are P.lslot bytes apart and the records have the long form (16-byte header, two bytes per row,
// every 16-byte chunk of the stride written, zeros past the rows) -- byte for byte what the long-sequence kernels
// write for the same pair.
template <int NG, bool WIDE, bool LONGREC = false>
__device__ __forceinline__ void sa_peptide_pair_body(const SaPepParams& P) {
    constexpr int NC = 4 * NG;
    constexpr int NACC = (NG + 1) / 2;
    extern __shared__ __align__(16) uint8_t smem[];
    int32_t* const prof = reinterpret_cast<int32_t*>(smem);
    // residue codes of the first sequence whose pairs are being traced back: column c at pcol[4 + 2 * c];
    // pcol[192 + c], pcol[256 + c]: those of the item's two first sequences
    constexpr int PS = SA_PAIR_PROF_STRIDE(NG);          // dwords between profile rows
    constexpr int PROF_BYTES = 21 * PS * 4;
    constexpr int SD = SA_PAIR_STAGE_DWORDS(NG);         // dwords of staging area per lane
    uint8_t* const pcol = smem + PROF_BYTES;
    uint32_t* const stage = reinterpret_cast<uint32_t*>(smem + PROF_BYTES + 2 * SA_LDS_EMASK_BYTES);
    const int lane = threadIdx.x;
    // The traceback addresses pcol by its offset inside the workgroup's LDS, a constant folded into the instruction: the
    // dynamic area must start at 0 (it does: the kernel has no other LDS object).  Should a toolchain ever place it
    // elsewhere, the launch reports it through the status word behind the queue head and does nothing; every wave leaves.
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)smem != 0u) {
        if (lane == 0) atomicOr(P.counter + SA_STATUS_WORD, SA_STATUS_LDS_BASE);
        return;
    }
    if (lane < 4) pcol[lane] = 0xff;     // the bytes in front of column 0's residue code (see traceback_pk)
    // the workgroup's part of the spill area (128 KiB): the direction rows of the low halves, then those of the high
    // halves, each row as long as the task's first sequences need (see store_dirs)
    const __amdgpu_buffer_rsrc_t spill = __builtin_amdgcn_make_buffer_rsrc(
        P.spill + (size_t)blockIdx.x * P.spill_rows * 256, 0, P.spill_rows * 1024, SA_BUFFER_FLAGS);
    const uint32_t cpr = P.stride >> 4;   // 16-byte chunks per record
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
        {
            const uint32_t slot_bytes = LONGREC ? P.lslot : (uint32_t)SA_LP;
            const int a1 = P.codes_fo[(size_t)p1 * slot_bytes + lane];
            const int a2 = has2 ? (int)P.codes_fo[(size_t)p2 * slot_bytes + lane] : SA_PAD_CODE;
            __syncthreads();   // previous item's readers are done with prof/pcol
            // all forty table values are requested before the first is used: one trip to memory per item, not twenty
            // (the loads of a row's two values used to sit between the previous row's LDS write and its own)
            int t1[20], t2[20];
#pragma unroll
            for (int b = 0; b < 20; b++) {
                t1[b] = P.sprime[a1 * SA_SPRIME_STRIDE + b];
                t2[b] = P.sprime[a2 * SA_SPRIME_STRIDE + b];
            }
#pragma unroll
            for (int b = 0; b < 20; b++) {
                // padding scores 1 instead of 0 (an odd column's table entry must stay positive); padded cells
                // lie to the right of and below every cell that is read
                const int s1 = max(t1[b], 1), s2 = max(t2[b], 1);
                if (lane < PS) prof[b * PS + SA_NB_PROF_INDEX(lane, PS)] = (lane & 1) ? SA_NB_TABLE_B(s1) | (SA_NB_TABLE_B(s2) << 16)
                                                             : SA_NB_TABLE_A(s1) | (SA_NB_TABLE_A(s2) << 16);
            }
            if (lane < PS) prof[20 * PS + SA_NB_PROF_INDEX(lane, PS)] = (lane & 1) ? SA_NB_TABLE_B(1) * 0x10001 : SA_NB_TABLE_A(1) * 0x10001;
            pcol[192 + lane] = (uint8_t)a1;      // kept for the tracebacks: see the loop over the two halves
            pcol[256 + lane] = (uint8_t)a2;
            __syncthreads();
        }
        const uint64_t n = P.nseq;
        uint8_t* const out_row1 = P.out + ((uint64_t)p1 * (2 * n - p1 - 1) / 2 - P.out_base_pair) * P.stride;
        uint8_t* const out_row2 = has2 ? P.out + ((uint64_t)p2 * (2 * n - p2 - 1) / 2 - P.out_base_pair) * P.stride : out_row1;
        const int gend = (max(lenp1, lenp2) + 3) >> 2;
        SA_PHASE(1);

        for (uint32_t task = 0; task < item.count; task++) {
            // ---------------------------------------------------------- operands
            // 64 consecutive entries of p1's valid list; the scan over buckets is wave-uniform
            const uint32_t vbase = item.v0 + task * 64;
            const uint32_t vl = vbase + lane;
            const uint32_t nvalid = P.nseq - 1 - p1;
            const bool active = vl < nvalid;
            if (!__any(active)) continue;
            uint32_t sidx = 0, cum = item.c0;
            // read through the constant address space: the tables do not change during the launch, the addresses are
            // wave-uniform, and loads of this kind go through the scalar cache instead of the vector memory pipeline
            typedef const __attribute__((address_space(4))) uint32_t* sa_const_u32;
            const sa_const_u32 cur_row = (sa_const_u32)(P.cursor + (size_t)p1 * P.nb);
            const sa_const_u32 bucket_base = (sa_const_u32)P.bucket_base;
            for (uint32_t b = item.b0; b < P.nb; b++) {
                const uint32_t first = bucket_base[b] + cur_row[b];
                const uint32_t cnt = bucket_base[b + 1] - first;
                if (vl >= cum && vl < cum + cnt) sidx = first + (vl - cum);
                cum += cnt;
                if (cum > vbase + 63) break;
            }
            const uint32_t fiq = P.fi_sorted[sidx];
            const int lenq = active ? (int)P.len_sorted[sidx] : 0;
            // the first 16-byte piece of the lane's second sequence is always needed: it is requested with the two loads
            // above, the further pieces (as far as the wave's longest second sequence reaches) in one batch below -- two
            // trips to memory per task where there were up to five, each piece's load waited for before the next was sent
            const uint4* const qsrc = reinterpret_cast<const uint4*>(P.codes_s) + sidx;
            const uint4 q0 = qsrc[0];
            const bool active2 = active && has2 && fiq > p2;
            const int maxlen = wave_max_nonneg(lenq);
            // only the pieces that hold residues of the wave's longest second sequence (plus the one the row loop's
            // look-ahead may touch); the pieces are stored piece-major, so what is not requested is not fetched either
            // (64-byte slots cost 55 GB of reads per launch of 50 000 peptides)
            const int nq = min(4, (maxlen >> 4) + 1);
            uint4 q1 = q0, q2 = q0, q3 = q0;
            if (nq > 1) q1 = qsrc[(size_t)P.npad];
            if (nq > 2) q2 = qsrc[(size_t)2 * P.npad];
            if (nq > 3) q3 = qsrc[(size_t)3 * P.npad];
            SA_PHASE(2);

            // -------------------------------------------------------------- fill
            uint32_t R[NC];      // the DP row, updated in place
#pragma unroll
            for (int c = 0; c < NC; c++) R[c] = (c & 1) ? SA_NB_UP_B : SA_NB_UP_A;
            uint32_t e[NACC];
#pragma unroll
            for (int w = 0; w < NACC; w++) e[w] = 0;

            // The lane's second sequence goes to its part of the staging area (bytes 8 .. 71, where the record's
            // row bytes will stand): the row loop then reads its residues from the LDS and contains no memory
            // loads, so nothing in it waits for the direction stores to be acknowledged.  (With the residues
            // loaded from memory row by row, every wait for them was also a wait for the stores of the row
            // before: requests complete in order.)
            uint8_t* const qres = reinterpret_cast<uint8_t*>(stage + lane * SD + 2);
            {
                uint2* dst = reinterpret_cast<uint2*>(qres);
                dst[0] = make_uint2(q0.x, q0.y); dst[1] = make_uint2(q0.z, q0.w);
                if (nq > 1) { dst[2] = make_uint2(q1.x, q1.y); dst[3] = make_uint2(q1.z, q1.w); }
                if (nq > 2) { dst[4] = make_uint2(q2.x, q2.y); dst[5] = make_uint2(q2.z, q2.w); }
                if (nq > 3) { dst[6] = make_uint2(q3.x, q3.y); dst[7] = make_uint2(q3.z, q3.w); }
            }
            // Direction rows, one pair at a time: 16-bit pieces of eight columns, n16 per row and lane; the rows of the
            // low halves first, then those of the high halves (layout and register form: traceback_pk).
            // (an odd piece that holds only the row's last group of four columns is stored as its upper byte)
            const int n16 = (gend + 1) >> 1;
            const int row_bytes = 256 * (n16 >> 1) + ((n16 & 1) ? ((gend & 1) ? 64 : 128) : 0);
            const int half_off = SA_LP * row_bytes;
            // GEND = the row's length in groups as a constant: what is stored, and how, is then decided at compile time
            auto store_dirs = [&](int r, auto GC) {
                constexpr int GEND = decltype(GC)::value;
                constexpr int N16 = (GEND + 1) / 2, WD = N16 / 2, ODD = N16 & 1, PB = (GEND & 1) ? 1 : 2;
                constexpr int ROW_BYTES = 256 * WD + 64 * PB * ODD;
                const int row_off = r * ROW_BYTES;
                const int hoff = SA_LP * ROW_BYTES + row_off;
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < WD) {
                        const uint32_t w0 = e[2 * k < NACC ? 2 * k : 0], w1 = e[2 * k + 1 < NACC ? 2 * k + 1 : 0];
                        lo[k] = __builtin_amdgcn_perm(w0, w1, 0x05040100u);     // low halves:  w0.lo : w1.lo
                        hi[k] = __builtin_amdgcn_perm(w0, w1, 0x07060302u);     // high halves: w0.hi : w1.hi
                    } else {
                        lo[k] = hi[k] = 0;
                    }
                }
                // Highest columns first; the dwords of the higher columns (WH of them) of all lanes stand in front
                // of those of the lower columns (see traceback_pk, which stops requesting the former).
                constexpr int WH = SA_DIR_WH(WD), WL = WD - WH;
                const int high_lane = lane * (WH * 4), low_lane = 256 * WH + lane * (WL * 4);
                if constexpr (WH == 2) {
                    const sa_u32x2 a = {lo[WD - 1], lo[WD - 2]}, b = {hi[WD - 1], hi[WD - 2]};
                    __builtin_amdgcn_raw_buffer_store_b64(a, spill, high_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(b, spill, high_lane, hoff, 0);
                } else if constexpr (WH == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(lo[WD - 1], spill, high_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(hi[WD - 1], spill, high_lane, hoff, 0);
                }
                if constexpr (WL == 4) {
                    const sa_u32x4 a = {lo[3], lo[2], lo[1], lo[0]}, b = {hi[3], hi[2], hi[1], hi[0]};
                    __builtin_amdgcn_raw_buffer_store_b128(a, spill, low_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(b, spill, low_lane, hoff, 0);
                } else if constexpr (WL == 3) {
                    const sa_u32x3 a = {lo[2], lo[1], lo[0]}, b = {hi[2], hi[1], hi[0]};
                    __builtin_amdgcn_raw_buffer_store_b96(a, spill, low_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b96(b, spill, low_lane, hoff, 0);
                } else if constexpr (WL == 2) {
                    const sa_u32x2 a = {lo[1], lo[0]}, b = {hi[1], hi[0]};
                    __builtin_amdgcn_raw_buffer_store_b64(a, spill, low_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(b, spill, low_lane, hoff, 0);
                } else if constexpr (WL == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(lo[0], spill, low_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(hi[0], spill, low_lane, hoff, 0);
                }
                if constexpr (ODD == 1) {
                    // the row's last eight columns: the word's two halves are the two pairs' pieces as they stand
                    const uint32_t w = e[2 * WD < NACC ? 2 * WD : 0];
                    const int piece_lane = 256 * WD + lane * PB;
                    if constexpr (PB == 2) {
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w & 0xffffu), spill, piece_lane, row_off, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w >> 16), spill, piece_lane, hoff, 0);
                    } else {       // four columns: the upper byte of each pair's piece
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((w >> 8) & 0xffu), spill, piece_lane, row_off, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(w >> 24), spill, piece_lane, hoff, 0);
                    }
                }
            };

            uint32_t dfin1 = 0, dfin2 = 0;       // bank values of the last cells of the two matrices
            // The row loop, once per row length: with the number of groups a constant the row is straight-line code
            // (no compare and branch per group, no test before a table request) and the shape of the direction stores
            // is fixed.  Items come sorted by length, so the waves of a CU run the same copy most of the time.
            auto fill_rows = [&](auto EC) {
                constexpr int END2 = decltype(EC)::value;                      // column pairs per row
                const std::integral_constant<int, (END2 + 1) / 2> GC;          // groups of four columns
                uint32_t o0 = qres[0] * PS;      // profile row (as a dword offset) of the row to come
                for (int r = 0; r < maxlen; r++) {
                    const uint32_t b1 = qres[min(r + 1, SA_LP - 1)];
                    // Every lane computes and stores every row, also lanes whose sequence 2 has ended (padding
                    // residues; those rows are never read): with the row under a per-lane condition the compiler
                    // keeps a second copy of the register row for the lanes that skip it.  A lane takes its
                    // scores when its last row is done.
                    fill_row_pair<NG, PS, END2, WIDE>(R, (END2 + 1) / 2, SA_NB_UP_B, 0u, prof + o0, e);
                    store_dirs(r, GC);
                    if (r + 1 == lenq) {
                        dfin1 = pick_packed<NC>(R, lenp1 - 1);
                        dfin2 = pick_packed<NC>(R, max(lenp2, 1) - 1);
                    }
                    o0 = b1 * PS;
                }
            };
            // The row loop is one long run of vector instructions; everything else (operands, the walk, records) is short
            // runs between waits.  Those phases get the higher issue priority, so that a wave in them sends its next
            // requests as soon as it can and the waves in the row loop take the cycles that are left: 1 % on the same box
            // (103.4-104.2 -> 102.0-102.3 ms; the other way round, +2.4 %).
            __builtin_amdgcn_s_setprio(0);
#define SA_ROWS(E) case E: if constexpr ((E) <= 2 * NG) fill_rows(std::integral_constant<int, (E) <= 2 * NG ? (E) : 1>()); break;
            switch ((max(lenp1, lenp2) + 1) >> 1) {
                SA_ROWS(1) SA_ROWS(2) SA_ROWS(3) SA_ROWS(4) SA_ROWS(5) SA_ROWS(6) SA_ROWS(7) SA_ROWS(8)
                SA_ROWS(9) SA_ROWS(10) SA_ROWS(11) SA_ROWS(12) SA_ROWS(13) SA_ROWS(14) SA_ROWS(15) SA_ROWS(16)
                SA_ROWS(17) SA_ROWS(18) SA_ROWS(19) SA_ROWS(20) SA_ROWS(21) SA_ROWS(22) SA_ROWS(23) SA_ROWS(24)
                SA_ROWS(25) SA_ROWS(26) SA_ROWS(27) SA_ROWS(28) SA_ROWS(29) SA_ROWS(30) SA_ROWS(31) SA_ROWS(32)
                default: break;
            }
#undef SA_ROWS
            __builtin_amdgcn_s_setprio(2);
            // the bank holds 16*D + a tag field per half
            const int d1 = (int)((dfin1 & 0xffffu) >> SA_NB_SHIFT);
            const int d2 = (int)(dfin2 >> (16 + SA_NB_SHIFT));
            __syncthreads();   // direction rows written by this wave are visible to its loads
            SA_PHASE(3);

            // the lane's second sequence, 64 residue codes in 16 registers (the staging area takes the records)
            uint32_t bw[16];
            {
                const uint2* src = reinterpret_cast<const uint2*>(qres);
#pragma unroll
                for (int q2 = 0; q2 < 8; q2++) {
                    const uint2 v = src[q2];
                    bw[2 * q2] = v.x; bw[2 * q2 + 1] = v.y;
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                const bool act = half ? active2 : active;
                if (!__any(act)) continue;
                const int lenp = half ? lenp2 : lenp1;
                const int lq = act ? lenq : 0;
                reinterpret_cast<uint16_t*>(pcol + 4)[lane] = (uint16_t)(0xff00u | pcol[192 + 64 * half + lane]);
                __syncthreads();
                const int mlen = wave_max_nonneg(lq);
                const int base = half ? half_off : 0;
                int matches = 0, gaps = 0;
#define SA_TRACEBACK(WD, ODD, PB, CLAMP)                                                                   \
    traceback_pk<WD, ODD, PB, CLAMP, PROF_BYTES + 4, SD>(lane, mlen, lq, lenp, spill, base, bw,            \
                                                         stage + lane * SD, matches, gaps)
                switch (gend) {          // groups of four columns -> dwords, odd piece, bytes of the piece
                    case 1: SA_TRACEBACK(0, 1, 1, false); break;
                    case 2: SA_TRACEBACK(0, 1, 2, false); break;
                    case 3: SA_TRACEBACK(1, 0, 2, false); break;
                    case 4: SA_TRACEBACK(1, 0, 2, true); break;
                    case 5: SA_TRACEBACK(1, 1, 1, false); break;
                    case 6: SA_TRACEBACK(1, 1, 2, false); break;
                    case 7: SA_TRACEBACK(2, 0, 2, false); break;
                    case 8: SA_TRACEBACK(2, 0, 2, true); break;
                    case 9: SA_TRACEBACK(2, 1, 1, true); break;
                    case 10: SA_TRACEBACK(2, 1, 2, true); break;
                    case 11: case 12: SA_TRACEBACK(3, 0, 2, true); break;
                    case 13: SA_TRACEBACK(3, 1, 1, true); break;
                    default:
                        if constexpr (NG > 13) { if (gend == 14) SA_TRACEBACK(3, 1, 2, true); else SA_TRACEBACK(4, 0, 2, true); }
                        break;
                }
#undef SA_TRACEBACK
                SA_PHASE(4);
                const int score = (half ? d2 : d1) + (lenp + lenq) * P.gap;
                stage[lane * SD + 0] = (uint32_t)score;
                stage[lane * SD + 1] = (uint32_t)matches | ((uint32_t)gaps << 16);
                __syncthreads();

                // ------------------------------------------- records, 16 B per lane store
                uint8_t* const out_row = half ? out_row2 : out_row1;
                const uint32_t rel = fiq - (half ? p2 : p1) - 1;
                if constexpr (LONGREC) {
                    // Chunk after chunk through the 64 records of the half: consecutive lanes write consecutive 16-byte chunks,
                    // 1 KiB per store instruction wherever the records follow each other in the row.  Chunk 0 is the header,
                    // chunks 1-8 can hold rows (64 rows of two bytes), the rest of the stride is zeros.
                    const uint32_t magic = 0xffffffffu / cpr + 1u;        // id / cpr = id * magic >> 32 for id < 64 * cpr
                    for (uint32_t id = lane; id < 64 * cpr; id += 64) {
                        const uint32_t rec = __umulhi(id, magic), ch = id - rec * cpr;
                        const uint32_t rrel = __shfl(rel, rec, 64);
                        const int ract = __shfl((int)act, rec, 64);
                        if (ract) {
                            sa_u32x4 v = {0u, 0u, 0u, 0u};
                            if (ch <= 8) {
                                const uint2 b = *reinterpret_cast<const uint2*>(stage + rec * SD + 2 * ch);    // ch = 0: score, matches | gaps << 16
                                if (ch == 0) {
                                    v = sa_u32x4{b.x, b.y & 0xffffu, b.y >> 16, 0u};
                                } else {         // rows 8 (ch - 1) .. + 7: a byte each in the staging area, two bytes in the record
                                    v = sa_u32x4{__builtin_amdgcn_perm(0u, b.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, b.x, 0x0c030c02u),
                                                 __builtin_amdgcn_perm(0u, b.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, b.y, 0x0c030c02u)};
                                }
                            }
                            __builtin_nontemporal_store(v, reinterpret_cast<sa_u32x4*>(out_row + (uint64_t)rrel * P.stride + ch * 16));
                        }
                    }
                } else if (cpr == 4) {
                    // 64-byte records (no sequence above 56 residues): four lanes per record, sixteen records per round; the
                    // record's place in the row and whether its pair exists come from its lane in one exchange, and the
                    // staging addresses differ from round to round by a constant.  (With the record index and chunk found
                    // by a division by a run-time chunk count, this loop was 280 of a task's 5 700 vector instructions.)
                    const uint32_t relact = act ? rel : 0xffffffffu;
                    const uint32_t* const src = stage + (lane >> 2) * SD + (lane & 3) * 4;
                    uint8_t* const dst = out_row + (lane & 3) * 16;
                    // the four exchanges first, then the eight staging reads, then the stores: two trips to the LDS
                    // for the task's records of this half instead of eight
                    uint32_t rrel[4];
#pragma unroll
                    for (int round = 0; round < 4; round++) rrel[round] = __shfl(relact, (lane >> 2) + 16 * round, 64);
                    uint2 va[4], vb[4];
#pragma unroll
                    for (int round = 0; round < 4; round++) {
                        va[round] = *reinterpret_cast<const uint2*>(src + round * 16 * SD);
                        vb[round] = *reinterpret_cast<const uint2*>(src + round * 16 * SD + 2);
                    }
#pragma unroll
                    for (int round = 0; round < 4; round++) {
                        if (rrel[round] != 0xffffffffu) {
                            const sa_u32x4 v = {va[round].x, va[round].y, vb[round].x, vb[round].y};
                            __builtin_nontemporal_store(v, reinterpret_cast<sa_u32x4*>(dst + (uint64_t)rrel[round] * 64));
                        }
                    }
                } else {
                    for (uint32_t id = lane; id < 64 * cpr; id += 64) {
                        const uint32_t rec = id / cpr, ch = id - rec * cpr;
                        const uint32_t rrel = __shfl(rel, rec, 64);
                        const int ract = __shfl((int)act, rec, 64);
                        if (ract) {
                            // written once and not read again on the device: kept out of the way of the direction rows
                            // (two 8-byte reads: a lane's part of the staging area is 8-byte aligned)
                            const uint2 va = *reinterpret_cast<const uint2*>(stage + rec * SD + ch * 4);
                            uint2 vb = make_uint2(0u, 0u);      // bytes 72 .. 79 of an 80-byte record: padding
                            if (ch * 4 + 2 < (uint32_t)SD) vb = *reinterpret_cast<const uint2*>(stage + rec * SD + ch * 4 + 2);
                            const sa_u32x4 v = {va.x, va.y, vb.x, vb.y};
                            __builtin_nontemporal_store(v, reinterpret_cast<sa_u32x4*>(out_row + (uint64_t)rrel * P.stride + ch * 16));
                        }
                    }
                }
                __syncthreads();   // staging area is reused
                SA_PHASE(5);
            }
        }
    }
    if (timed && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) P.phase_cycles[(size_t)blockIdx.x * 8 + k] = tph[k];
    }
#undef SA_PHASE
}

template <int NG, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_peptide_pair_kernel(const SaPepParams P) {
    sa_peptide_pair_body<NG, false>(P);
}
// the same kernel for tables whose values leave the range of the half-precision patterns
template <int NG, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_peptide_pair_wide_kernel(const SaPepParams P) {
    sa_peptide_pair_body<NG, true>(P);
}

// the peptide tiles of a set with long sequences in it (records in the long form)
template <int NG, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_peptide_pair_longrec_kernel(const SaPepParams P) {
    sa_peptide_pair_body<NG, false, true>(P);
}
template <int NG, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_peptide_pair_wide_longrec_kernel(const SaPepParams P) {
    sa_peptide_pair_body<NG, true, true>(P);
}

// ng: 4-column groups the register row holds (13: sequences up to 52 residues, 16: up to 64).  Both fit 128
// registers, four waves per SIMD; the shorter row leaves room in the LDS for one more workgroup per CU.
extern "C" void sa_launch_peptide_pair(int ng, int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (wide) {
        if (ng <= 13)
            hipLaunchKernelGGL((sa_peptide_pair_wide_kernel<13, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
        else
            hipLaunchKernelGGL((sa_peptide_pair_wide_kernel<16, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    } else if (ng <= 13) {
        hipLaunchKernelGGL((sa_peptide_pair_kernel<13, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    } else {
        hipLaunchKernelGGL((sa_peptide_pair_kernel<16, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    }
}

// the same kernels writing long-form records (p->lslot > 64, p->stride >= 160)
extern "C" void sa_launch_peptide_pair_longrec(int ng, int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (wide) {
        if (ng <= 13)
            hipLaunchKernelGGL((sa_peptide_pair_wide_longrec_kernel<13, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
        else
            hipLaunchKernelGGL((sa_peptide_pair_wide_longrec_kernel<16, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    } else if (ng <= 13) {
        hipLaunchKernelGGL((sa_peptide_pair_longrec_kernel<13, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    } else {
        hipLaunchKernelGGL((sa_peptide_pair_longrec_kernel<16, 4>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    }
}

// Loads this file's code object on the current device without launching anything (the runtime otherwise loads it at
// the first launch of one of its kernels: milliseconds on the critical path of a short job).
extern "C" int sa_load_code_peptide_pair(void) {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(sa_peptide_pair_kernel<13, 4>));
}
