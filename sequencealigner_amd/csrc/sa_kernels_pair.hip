// sa_kernels_pair.hip -- TRIANGULAR peptides, two first sequences per wave.
//
// Same recurrence and the same tagged cells as sa_kernels.hip, but every VGPR of
// the DP row holds two cells: the low 16 bits belong to the pair (p1, q), the high
// 16 bits to (p2, q), where q is the lane's second sequence and p1 < p2 are two
// first sequences of the same length bucket.  A lane's row residue b is the same
// for both matrices, so one LDS lookup prof[b][c] = T(p1)[b][c] | T(p2)[b][c] << 16
// feeds both; the add is a plain 32-bit add (no carry can cross: every half stays
// below 2^15), the two max are v_pk_max_i16, and the tag/clean/bank steps are
// 32-bit bit operations on both halves at once.  Per two cells: add, 2 x pk_max,
// and (tags), shift-or (direction word), and (left operand), or (bank value) --
// 3.5 instructions per cell against 5 for the unpaired kernel.
//
// Tags: 0 LEFT, 2 UP, 3 DIAG (bank values 4*D + 2, table entries 4*S' + 1), so that
// with the newest column in the low bits of a direction word a count-trailing-zeros
// finds a run of LEFT moves (sa_traceback.h, REV = 1).
//
// The host uses this kernel only when every half provably stays in range:
// 0 <= S' <= 120 for the whole table (BLOSUM50 with gap -4: 3 .. 23).
//
// The lanes of a task are 64 consecutive entries of p1's valid list; the pair
// (p2, q) exists only where q's file index is above p2 (all but the few sequences
// between p1 and p2), elsewhere the high half is computed and dropped.
#include "sa_traceback.h"

typedef short sa_i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(sa_i16x2, a), __builtin_bit_cast(sa_i16x2, b)));
}

#define SA_PK_TAGS 0x00030003u
#define SA_PK_CLEAN 0xfffcfffcu
#define SA_PK_UP 0x00020002u                         /* bank form of D = 0 in both halves */
#define SA_PK_TABLE(sprime) (4 * (sprime) + 1)       /* table entry for a substitution score S' */

// One packed cell.  The direction word of eight columns starts afresh at the first of
// them (nothing is shifted across the 16-bit boundary: a word takes at most eight tags).
// The tag and bank steps are written as instructions so that they stay where they are:
// moved to the end of the row by the compiler they keep one more register per column
// alive, and the row no longer fits the register file.
#define SA_CELL_PK(IN, OUT, C, T, ACC)                                                   \
    {                                                                                    \
        const uint32_t diag_ = ((C) == 0) ? SA_PK_UP : IN[((C) == 0) ? 0 : (C) - 1];     \
        const uint32_t m_ = diag_ + (uint32_t)(T);                                       \
        const uint32_t h_ = pk_max(pk_max(m_, IN[C]), zl);                               \
        if (((C) & 7) == 0) {                                                            \
            asm volatile("v_and_b32 %0, 0x30003, %1" : "=v"(ACC) : "v"(h_));             \
        } else {                                                                         \
            uint32_t t_;                                                                 \
            asm volatile("v_and_b32 %0, 0x30003, %1" : "=v"(t_) : "v"(h_));              \
            asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ACC) : "v"(t_));           \
        }                                                                                \
        zl = h_ & SA_PK_CLEAN;                                                           \
        asm volatile("v_or_b32 %0, 0x20002, %1" : "=v"(OUT[C]) : "v"(zl));               \
    }

// Four columns.  Rows end early: gend = ceil(longer first sequence / 4).
#define SA_GROUP_PK(G, CUR, NXT)                                                   \
    NXT = *reinterpret_cast<const int4*>(row_base + 4 * ((G) + 1));                \
    __builtin_amdgcn_sched_barrier(0);                                             \
    SA_CELL_PK(IN, OUT, 4 * (G) + 0, CUR.x, e[(G) >> 1])                           \
    SA_CELL_PK(IN, OUT, 4 * (G) + 1, CUR.y, e[(G) >> 1])                           \
    SA_CELL_PK(IN, OUT, 4 * (G) + 2, CUR.z, e[(G) >> 1])                           \
    SA_CELL_PK(IN, OUT, 4 * (G) + 3, CUR.w, e[(G) >> 1])                           \
    if (NG == (G) + 1 || gend == (G) + 1) return;

// One packed DP row: IN = previous row, OUT = this row, columns 0 .. 4*gend-1.
// e[w] receives the tags of columns 8w .. 8w+7 of both pairs: column 8w+i at bits
// 2(7-i) of each 16-bit half -- or, when the row ends after four columns of the word
// (gend odd), column 8w+i at bits 2(3-i).
template <int NG>
__device__ __forceinline__ void fill_row_pair(const uint32_t (&IN)[4 * NG], uint32_t (&OUT)[4 * NG], const int gend,
                                              const int32_t* __restrict__ row_base,
                                              const int32_t* __restrict__ next_row_base, const int4& sc_in, int4& sc_out,
                                              uint32_t (&e)[(NG + 1) / 2]) {
    int4 scE = sc_in, scO;
    uint32_t zl = 0;
    sc_out = *reinterpret_cast<const int4*>(next_row_base);
    SA_GROUP_PK(0, scE, scO)
    SA_GROUP_PK(1, scO, scE)
    SA_GROUP_PK(2, scE, scO)
    SA_GROUP_PK(3, scO, scE)
    SA_GROUP_PK(4, scE, scO)
    SA_GROUP_PK(5, scO, scE)
    SA_GROUP_PK(6, scE, scO)
    SA_GROUP_PK(7, scO, scE)
    SA_GROUP_PK(8, scE, scO)
    SA_GROUP_PK(9, scO, scE)
    SA_GROUP_PK(10, scE, scO)
    SA_GROUP_PK(11, scO, scE)
    SA_GROUP_PK(12, scE, scO)
    if constexpr (NG > 13) {
        SA_GROUP_PK(13, scO, scE)
        SA_GROUP_PK(14, scE, scO)
        SA_GROUP_PK(15, scO, scE)
    }
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

template <int NG, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) sa_peptide_pair_kernel(const SaPepParams P) {
    constexpr int NC = 4 * NG;
    constexpr int NACC = (NG + 1) / 2;
    extern __shared__ __align__(16) uint8_t smem[];
    int32_t* const prof = reinterpret_cast<int32_t*>(smem);
    uint64_t* const emask = reinterpret_cast<uint64_t*>(smem + SA_LDS_PROF_BYTES);                 // [2][24]
    uint32_t* const stage = reinterpret_cast<uint32_t*>(smem + SA_LDS_PROF_BYTES + 2 * SA_LDS_EMASK_BYTES);
    const int lane = threadIdx.x;
    // rows 0..63 of the spill area hold the direction rows of the low halves, rows 64..127 those of the high halves
    const __amdgpu_buffer_rsrc_t spill = __builtin_amdgcn_make_buffer_rsrc(
        P.spill + (size_t)blockIdx.x * P.spill_rows * 256, 0, P.spill_rows * 1024, SA_BUFFER_FLAGS);
    const __amdgpu_buffer_rsrc_t codes_t = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(P.codes_t), 0, SA_LP * P.npad, SA_BUFFER_FLAGS);
    // the same descriptor as four scalar words, for the hand-written residue requests
    const uint64_t ct_addr = reinterpret_cast<uint64_t>(P.codes_t);
    const sa_u32x4 codes_t_words = {(uint32_t)ct_addr, (uint32_t)(ct_addr >> 32) & 0xffffu, SA_LP * P.npad, SA_BUFFER_FLAGS};
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
            const int a1 = P.codes_fo[(size_t)p1 * SA_LP + lane];
            const int a2 = has2 ? (int)P.codes_fo[(size_t)p2 * SA_LP + lane] : SA_PAD_CODE;
            __syncthreads();   // previous item's readers are done with prof/emask
#pragma unroll
            for (int b = 0; b < 20; b++)
                prof[b * SA_PROF_STRIDE + lane] = SA_PK_TABLE(P.sprime[a1 * SA_SPRIME_STRIDE + b]) |
                                                  (SA_PK_TABLE(P.sprime[a2 * SA_SPRIME_STRIDE + b]) << 16);
            prof[20 * SA_PROF_STRIDE + lane] = SA_PK_TABLE(0) | (SA_PK_TABLE(0) << 16);
#pragma unroll
            for (int b = 0; b < 20; b++) {
                const uint64_t m1 = __ballot(a1 == b), m2 = __ballot(a2 == b);
                if (lane == b) { emask[b] = m1; emask[24 + b] = m2; }
            }
            if (lane == 20) { emask[20] = 0; emask[24 + 20] = 0; }
            __syncthreads();
        }
        const uint64_t n = P.nseq;
        uint8_t* const out_row1 = P.out + ((uint64_t)p1 * (2 * n - p1 - 1) / 2 - P.out_base_pair) * P.stride;
        uint8_t* const out_row2 = has2 ? P.out + ((uint64_t)p2 * (2 * n - p2 - 1) / 2 - P.out_base_pair) * P.stride : out_row1;
        const int gend = (max(lenp1, lenp2) + 3) >> 2;
        const int wpr = (gend + 3) >> 2;                // dwords per direction row, lane and pair
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
            const int maxlen = __builtin_amdgcn_readfirstlane(wave_max(lenq));
            SA_PHASE(2);

            // -------------------------------------------------------------- fill
            uint32_t R[NC], Q[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) { R[c] = SA_PK_UP; Q[c] = SA_PK_UP; }
            uint32_t e[NACC];
#pragma unroll
            for (int w = 0; w < NACC; w++) e[w] = 0;

            // Row residue r of the lane's second sequence.  The request is written as an instruction of its own,
            // outside the compiler's bookkeeping of outstanding memory operations: the compiler would wait for
            // *everything* in flight before the first use, direction stores included; here the wait is counted by
            // hand (see the end of the row loop).
            auto residue = [&](int r) -> uint32_t {
                uint32_t v;
                const int soff = min(r, SA_LP - 1) * (int)P.npad;
                asm volatile("buffer_load_ubyte %0, %1, %2, %3 offen" : "=v"(v) : "v"(sidx), "s"(codes_t_words), "s"(soff) : "memory");
                return v;
            };
            const int dir_lane = lane * wpr * 4;      // byte offset of this lane inside a direction row
            const int half_off = SA_LP * 256 * wpr;   // byte offset of the high halves' rows
            // direction row of one pair: dword k = columns 16k .. 16k+15, column 16k+i at bits 2(15-i)
            auto store_dirs = [&](int r, int g) {
                const int wr = (g + 3) >> 2;
                const int half_word = (g & 1) ? (g - 1) >> 1 : -1;     // index of a half-full direction word, if any
                const int row_off = r * 256 * wpr;
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // a last word that took only four columns moves them up to where the first four of a full word sit
                    uint32_t w0 = 0, w1 = 0;
                    if (2 * k < NACC) w0 = e[(2 * k < NACC) ? 2 * k : 0] << ((half_word == 2 * k) ? 8 : 0);
                    if (2 * k + 1 < NACC) w1 = e[(2 * k + 1 < NACC) ? 2 * k + 1 : 0] << ((half_word == 2 * k + 1) ? 8 : 0);
                    lo[k] = __builtin_amdgcn_perm(w0, w1, 0x05040100u);     // low halves:  w0.lo : w1.lo
                    hi[k] = __builtin_amdgcn_perm(w0, w1, 0x07060302u);     // high halves: w0.hi : w1.hi
                }
                if (wr == 4) {
                    const sa_u32x4 a = {lo[0], lo[1], lo[2], lo[3]}, b = {hi[0], hi[1], hi[2], hi[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(a, spill, dir_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(b, spill, dir_lane, half_off + row_off, 0);
                } else if (wr == 3) {
                    const sa_u32x3 a = {lo[0], lo[1], lo[2]}, b = {hi[0], hi[1], hi[2]};
                    __builtin_amdgcn_raw_buffer_store_b96(a, spill, dir_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b96(b, spill, dir_lane, half_off + row_off, 0);
                } else if (wr == 2) {
                    const sa_u32x2 a = {lo[0], lo[1]}, b = {hi[0], hi[1]};
                    __builtin_amdgcn_raw_buffer_store_b64(a, spill, dir_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(b, spill, dir_lane, half_off + row_off, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(lo[0], spill, dir_lane, row_off, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(hi[0], spill, dir_lane, half_off + row_off, 0);
                }
            };

            // Profile rows (as dword offsets) of the two rows to come.  The residues of rows r+2, r+3 are requested
            // at the top of an iteration and turned into offsets at its end, after the four direction stores, so
            // that the wait for them stands where those stores may stay in flight.
            const uint32_t ra = residue(0), rb = residue(1);
            uint32_t o0, o1;
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mul_u32_u24 %0, %2, %4\n\tv_mul_u32_u24 %1, %3, %4"
                         : "=&v"(o0), "=&v"(o1) : "v"(ra), "v"(rb), "s"(SA_PROF_STRIDE));
            int4 sc0 = *reinterpret_cast<const int4*>(prof + o0);
            for (int r = 0; r < maxlen; r += 2) {
                const uint32_t b2 = residue(r + 2), b3 = residue(r + 3);
                // the row's end tests compare this copy of gend where they stand: hoisted out of the loop they
                // would each hold a pair of scalar registers, more than there are
                int gr = gend;
                asm volatile("" : "+s"(gr));
                // every lane stores its direction row, also lanes whose sequence 2 has ended (those rows are never
                // read): every wave then issues exactly four stores per iteration
                if (r < lenq) fill_row_pair<NG>(R, Q, gr, prof + o0, prof + o1, sc0, sc0, e);
                store_dirs(r, gr);
                // Memory operations finish in the order they were issued: with at most the four direction stores
                // of this iteration still in flight, the two residues requested before them have arrived.
                uint32_t o2, o3;
                asm volatile("s_waitcnt vmcnt(2)\n\tv_mul_u32_u24 %0, %2, %4\n\tv_mul_u32_u24 %1, %3, %4"
                             : "=&v"(o2), "=&v"(o3) : "v"(b2), "v"(b3), "s"(SA_PROF_STRIDE));
                if (r + 1 < lenq) fill_row_pair<NG>(Q, R, gr, prof + o1, prof + o2, sc0, sc0, e);
                store_dirs(r + 1, gr);
                o0 = o2; o1 = o3;
            }
            // banks hold 4*D + 2 per half
            const uint32_t y1q = pick_packed<NC>(Q, lenp1 - 1), y1r = pick_packed<NC>(R, lenp1 - 1);
            const uint32_t y2q = pick_packed<NC>(Q, max(lenp2, 1) - 1), y2r = pick_packed<NC>(R, max(lenp2, 1) - 1);
            const int d1 = (int)((((lenq & 1) ? y1q : y1r) & 0xffffu) >> 2);
            const int d2 = (int)(((lenq & 1) ? y2q : y2r) >> 18);
            __syncthreads();   // direction rows written by this wave are visible to its loads
            SA_PHASE(3);

            // the lane's second sequence, 64 residue codes in 16 registers
            uint32_t bw[16];
            {
                const uint4* src = reinterpret_cast<const uint4*>(P.codes_s + (size_t)sidx * SA_LP);
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const uint4 v = src[q4];
                    bw[4 * q4 + 0] = v.x; bw[4 * q4 + 1] = v.y; bw[4 * q4 + 2] = v.z; bw[4 * q4 + 3] = v.w;
                }
            }
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                const bool act = half ? active2 : active;
                if (!__any(act)) continue;
                const int lenp = half ? lenp2 : lenp1;
                const int lq = act ? lenq : 0;
                const int mlen = __builtin_amdgcn_readfirstlane(wave_max(lq));
                const int base = half ? half_off : 0;
                const uint64_t* em = emask + 24 * half;
                int matches = 0, gaps = 0;
#define SA_TRACEBACK(WW)                                                                                   \
    traceback<SA_MODE_TRI, WW, 1>(lane, mlen, lq, lenp, spill, base, bw, nullptr, nullptr, em,               \
                                  stage + lane * SA_LDS_STAGE_DWORDS, matches, gaps)
                if (wpr == 4) SA_TRACEBACK(4);
                else if (wpr == 3) SA_TRACEBACK(3);
                else if (wpr == 2) SA_TRACEBACK(2);
                else SA_TRACEBACK(1);
#undef SA_TRACEBACK
                SA_PHASE(4);
                const int score = (half ? d2 : d1) + (lenp + lenq) * P.gap;
                stage[lane * SA_LDS_STAGE_DWORDS + 0] = (uint32_t)score;
                stage[lane * SA_LDS_STAGE_DWORDS + 1] = (uint32_t)matches | ((uint32_t)gaps << 16);
                __syncthreads();

                // ------------------------------------------- records, 16 B per lane store
                uint8_t* const out_row = half ? out_row2 : out_row1;
                const uint32_t rel = fiq - (half ? p2 : p1) - 1;
                for (uint32_t id = lane; id < 64 * cpr; id += 64) {
                    const uint32_t rec = id / cpr, ch = id - rec * cpr;
                    const uint32_t rrel = __shfl(rel, rec, 64);
                    const int ract = __shfl((int)act, rec, 64);
                    if (ract) {
                        const uint4 v = *reinterpret_cast<const uint4*>(stage + rec * SA_LDS_STAGE_DWORDS + ch * 4);
                        *reinterpret_cast<uint4*>(out_row + (uint64_t)rrel * P.stride + ch * 16) = v;
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

// ng: 4-column groups the register row holds (13: sequences up to 52 residues, 16: up to 64)
extern "C" void sa_launch_peptide_pair(int ng, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream) {
    if (ng <= 13)
        hipLaunchKernelGGL((sa_peptide_pair_kernel<13, 3>), dim3(grid), dim3(64), lds_bytes, stream, *p);
    else
        hipLaunchKernelGGL((sa_peptide_pair_kernel<16, 3>), dim3(grid), dim3(64), lds_bytes, stream, *p);
}
