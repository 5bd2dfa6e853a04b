// sa_kernels_coop.hip -- one wave per pair: the anti-diagonal sweep for submissions of FEW long pairs.
//
// The pair-per-lane kernels (sa_kernels_long*.hip) need 64 pairs to fill one wave and thousands of
// waves to fill the chip; a submission of a few hundred protein pairs (ADJACENT over a small file,
// a short TRIANGULAR row range) leaves the GPU idle and takes as long as 64 pairs on one lane each.
// Here the 64 lanes of a wave share ONE pair (reference: include/seqalign.h:47-88, one call of
// align_sequences):
//
//  * fill: the matrix is swept in strips of 64 columns, lane l owns column 64 s + l of the strip, and
//    at step t it computes the cell of row t - l: the cells of one step lie on an anti-diagonal.
//    Its three operands: UP is its own result of the step before; LEFT is lane l-1's result of the
//    step before, fetched with one DPP wave shift (wave_shr:1); DIAG is what LEFT was one step earlier.
//    The row residue travels the same way: lane l needs seq2[t - l], which lane l-1 held a step
//    before -- a second DPP shift, lane 0 taking the next residue.  The substitution score is one LDS
//    read of the table S' indexed by the lane's column residue and the travelling row residue.
//    Cells are the tagged 32-bit cells of sa_kernel_common.h (4 D + tag, DIAG 3 > UP 1 > LEFT 0 decided by
//    the three-way maximum itself).  Rows above and below the matrix are fed a residue code whose score is
//    hugely negative: above, those cells stay at the border value D = 0 by induction, so a lane needs no
//    test for "has my column started"; below, a column keeps the value of the matrix's last row (UP wins:
//    D never decreases along a row), so the score is read from lane (len1 - 1) mod 64 after the last strip.  The last column of a strip is collected lane by lane
//    (v_readlane + v_writelane) and handed to the next strip as its left border, 64 rows at a time.
//  * directions: every lane shifts its cell's tag into a dword, 16 steps per dword, and the wave stores
//    the 64 dwords of a 16-step block with one coalesced 256-byte store.  The tag of cell (r, c) is at
//    bits 2 (t mod 16) of dword [strip][t / 16][c mod 64], t = r + c mod 64.
//  * traceback: one walk per wave, all in scalar registers; the 64 dwords of the block the walk is in
//    sit in one VGPR (one lane each) and are picked with v_readlane; leaving the block loads the next.
//    Per row the walk notes how the row was left (LEFT moves * 2 + UP/DIAG, the record's uint16) and
//    where; matches and gaps are then counted 64 rows at a time.
//
// Work items: one pair each, {p = sequence 1, v0 = sequence 2 (file indices), b0 | c0 << 32 = record
// index relative to the submission's first pair}.  Records are the long-sequence records of
// sa_device.h, byte-identical to those of sa_long_kernel.
#include "sa_kernel_common.h"

#define SA_COOP_AHEAD 3                     /* steps between the request of a table entry and its use */
#define SA_COOP_NEG_CODE 21                 /* row residue of the rows above the matrix */
#define SA_COOP_DPP_WAVE_SHR1 0x138         /* DPP control: lane l reads lane l-1 across the whole wave */

__device__ __forceinline__ int sa_wave_shr1(int old_for_lane0, int v) {
    // lane 0 has no source lane: with bound_ctrl off it keeps `old`
    return __builtin_amdgcn_update_dpp(old_for_lane0, v, SA_COOP_DPP_WAVE_SHR1, 0xf, 0xf, false);
}

// v_writelane_b32: one lane of `dst` takes a wave-uniform value.  In the fill the lane index is a literal.  In the
// walk value and lane index are both scalars, and a vector instruction of this generation reads one scalar register
// only: the lane index goes through M0 (one wait state after the scalar move), and as it was computed from v_readlane
// results the move itself must come four wait states after them.
#define SA_WRITELANE_CONST(DST, VALUE, LANE) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(DST) : "s"(VALUE), "n"(LANE))
__device__ __forceinline__ uint32_t sa_writelane(int value, int lane_index, uint32_t dst) {
    value = __builtin_amdgcn_readfirstlane(value);
    lane_index = __builtin_amdgcn_readfirstlane(lane_index);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(dst) : "s"(value), "s"(lane_index) : "m0");
    return dst;
}

// One step of the sweep, for both kernels of this file (the names are the kernels' locals); TL = t mod 64 is a literal,
// so that lane indices are immediates.
#define SA_COOP_STEP(TL)                                                                                              \
    {                                                                                                                 \
        const int T = Tq[0];                                                                                          \
        _Pragma("unroll") for (int k = 0; k + 1 < SA_COOP_AHEAD; k++) Tq[k] = Tq[k + 1];                              \
        Tq[SA_COOP_AHEAD - 1] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(tab) + a32 + b * 4);  \
        /* the residues of the step after: from the lane on the left, lane 0 takes the next residue */               \
        b = sa_wave_shr1(__builtin_amdgcn_readlane(resin, (TL)), b);                                                  \
        const int m = left_prev + T;                                                                                  \
        const int h = max3i(m, up, left);                                                                             \
        acc = __builtin_amdgcn_alignbit((uint32_t)h, acc, 2);                                                         \
        const int zl = h & ~3;                                                                                        \
        up = zl | 1;                                                                                                  \
        left_prev = left;                                                                                             \
        /* the strip's last column, row t - 63, for the next strip */                                                 \
        const int last = __builtin_amdgcn_readlane(zl, 63);                                                           \
        SA_WRITELANE_CONST(colbuf, last, (TL));                                                                       \
        /* next step's LEFT comes from the lane on the left; lane 0 takes the border column's next row */             \
        left = sa_wave_shr1(__builtin_amdgcn_readlane(colin, (TL)), zl);                                              \
    }
#define SA_COOP_BLOCK(B)                                                                                              \
    SA_COOP_STEP(16 * (B) + 0) SA_COOP_STEP(16 * (B) + 1) SA_COOP_STEP(16 * (B) + 2) SA_COOP_STEP(16 * (B) + 3)       \
    SA_COOP_STEP(16 * (B) + 4) SA_COOP_STEP(16 * (B) + 5) SA_COOP_STEP(16 * (B) + 6) SA_COOP_STEP(16 * (B) + 7)       \
    SA_COOP_STEP(16 * (B) + 8) SA_COOP_STEP(16 * (B) + 9) SA_COOP_STEP(16 * (B) + 10) SA_COOP_STEP(16 * (B) + 11)     \
    SA_COOP_STEP(16 * (B) + 12) SA_COOP_STEP(16 * (B) + 13) SA_COOP_STEP(16 * (B) + 14) SA_COOP_STEP(16 * (B) + 15)   \
    dirs[((size_t)s * blocks_cap + 4 * sb + (B)) * 64 + lane] = acc;

__global__ void __launch_bounds__(64, 4) sa_coop_kernel(const SaPepParams P) {
    __shared__ int32_t tab[21 * 32];        // tab[a * 32 + b] = 4 S'[a][b] + 3; b = SA_COOP_NEG_CODE: far below any score
    const int lane = threadIdx.x;
    const uint32_t lslot = P.lslot;
    const uint32_t strips_cap = lslot >> 6;
    const uint32_t steps_cap = lslot + 128;                 // steps of a strip, rounded up to 64, at most
    const uint32_t blocks_cap = steps_cap >> 4;
    // per-workgroup scratch: direction blocks | two border columns | per-row notes of the walk
    uint8_t* const area = reinterpret_cast<uint8_t*>(P.spill) + (size_t)blockIdx.x * P.spill_rows * 1024;
    uint32_t* const dirs = reinterpret_cast<uint32_t*>(area);
    int32_t* const bcol = reinterpret_cast<int32_t*>(area + (size_t)strips_cap * blocks_cap * 256);
    uint16_t* const note_l = reinterpret_cast<uint16_t*>(bcol + 2 * steps_cap);
    uint16_t* const note_c = note_l + lslot + 64;

    for (int e = lane; e < 21 * 32; e += 64) {
        const int a = e >> 5, b = e & 31;
        int v = SA_TAG_TABLE(P.sprime[a * SA_SPRIME_STRIDE + (b <= 20 ? b : 20)]) + 1;     // + 1: DIAG operand is LEFT's clean value
        if (b == SA_COOP_NEG_CODE) v = -(1 << 29);
        tab[e] = v;
    }
    __syncthreads();

    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(P.counter, 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= P.nitems) break;
        const SaItem item = P.items[it];
        const uint32_t s1 = item.p, s2 = item.v0;
        const int len1 = __builtin_amdgcn_readfirstlane((int)P.len_fo[s1]);
        const int len2 = __builtin_amdgcn_readfirstlane((int)P.len_fo[s2]);
        const uint8_t* const seq1 = P.codes_fo + (size_t)s1 * lslot;
        const uint8_t* const seq2 = P.codes_fo + (size_t)s2 * lslot;
        uint8_t* const rec = P.out + (((uint64_t)item.c0 << 32) | item.b0) * P.stride;
        const int nstrips = (len1 + 63) >> 6;
        const int nsuper = (len2 + 63 + 63) >> 6;          // 64-step super-blocks of a strip: steps 0 .. len2 + 62
        const int lstar = (len1 - 1) & 63;
        int dfinal = 0;

        // ------------------------------------------------------------------ fill
        for (int s = 0; s < nstrips; s++) {
            const int a32 = (int)seq1[64 * s + lane] * 32 * 4;       // byte offset of the lane's table row (padding: code 20)
            const int32_t* bin = bcol + (size_t)(s & 1) * steps_cap;
            int32_t* bout = bcol + (size_t)((s + 1) & 1) * steps_cap;
            const bool first_strip = s == 0, more = s + 1 < nstrips;
            int up = 1;                                  // 4 * 0 + 1: the border row
            int left = first_strip ? 0 : bin[0];         // lane 0: clean value of (row 0, column 64 s - 1); other lanes: rows above the matrix
            if (lane != 0) left = 0;
            int left_prev = 0;                           // what LEFT was a step ago = the DIAG operand's clean value
            int b = lane == 0 ? (int)seq2[0] : SA_COOP_NEG_CODE;
            uint32_t acc = 0;
            int colbuf = 0;
            // Table entries are requested SA_COOP_AHEAD steps before they are used: the row residues travel along the
            // lanes independently of the cells, so their wave runs ahead of the cells' and the LDS latency (several
            // steps of arithmetic with one wave per SIMD, the regime this kernel is for) stays covered.
            // b = travelling residues of step t + AHEAD; Tq[k] = table entry of step t + k.
            int Tq[SA_COOP_AHEAD];
#pragma unroll
            for (int k = 0; k < SA_COOP_AHEAD; k++) {
                Tq[k] = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(tab) + a32 + b * 4);
                b = sa_wave_shr1(k + 1 < len2 ? (int)seq2[k + 1] : SA_COOP_NEG_CODE, b);
            }
            for (int sb = 0; sb < nsuper; sb++) {
                // rows 64 sb + 1 .. 64 sb + 64 for lane 0's next steps: residues and the border column
                const int rown = 64 * sb + 1 + lane;
                const int rres = rown + SA_COOP_AHEAD;           // the residues run ahead of the cells
                const int resin = rres < len2 ? (int)seq2[rres] : SA_COOP_NEG_CODE;      // below the matrix as above it
                // below the matrix the border is 0 (what the buffer holds there is stale, and a LEFT operand above the
                // frozen column value would thaw it)
                const int colin = (first_strip || rown >= len2) ? 0 : bin[rown];
                SA_COOP_BLOCK(0) SA_COOP_BLOCK(1) SA_COOP_BLOCK(2) SA_COOP_BLOCK(3)
                // colbuf lane i = last column of row 64 sb + i - 63
                if (more) {
                    const int r = 64 * sb + lane - 63;
                    if (r >= 0) bout[r] = colbuf;
                }
            }
            // rows below the matrix score like the rows above it: every lane's UP value has stayed what it was in the
            // matrix's last row (D never decreases along a row, so LEFT cannot pass it)
            if (s == nstrips - 1) dfinal = __builtin_amdgcn_readlane(up, lstar) >> 2;
            __syncthreads();      // border column and direction blocks are complete before anyone reads them
        }

        // ------------------------------------------------------------- traceback
        // State in scalar registers: i rows and j columns still to go, run = LEFT moves taken in the current row.
        int i = len2, j = len1, run = 0;
        int cur_s = -1, cur_k = -1;
        uint32_t w = 0;                       // the direction block the walk is in: lane l holds column l's dword
        uint32_t out_l = 0, out_c = 0;        // notes of 64 rows, lane r mod 64
        while (i > 0) {
            int tag = SA_TAG_UP;              // left border: UP
            if (j > 0) {
                const int c = j - 1, l = c & 63, sidx = c >> 6, t = i - 1 + l, k = t >> 4;
                if (sidx != cur_s || k != cur_k) {
                    w = dirs[((size_t)sidx * blocks_cap + k) * 64 + lane];
                    cur_s = sidx; cur_k = k;
                }
                tag = (int)((__builtin_amdgcn_readlane(w, l) >> (2 * (t & 15))) & 3u);
            }
            if (tag == SA_TAG_LEFT) { run++; j--; continue; }
            const int row = i - 1;
            const int note = 2 * run + (tag == SA_TAG_UP ? 1 : 0);
            out_l = sa_writelane(note, row & 63, out_l);
            out_c = sa_writelane(j > 0 ? j - 1 : 0, row & 63, out_c);
            if ((row & 63) == 0) {            // rows row .. row + 63 are noted: write them
                if (row + lane < len2) { note_l[row + lane] = (uint16_t)out_l; note_c[row + lane] = (uint16_t)out_c; }
            }
            if (tag == SA_TAG_DIAG) j--;
            i--;
            run = 0;
        }
        __syncthreads();

        // ------------------------------------------------------ record, 64 rows at a time
        int matches = 0, gaps = 0;
        for (int r0 = 0; r0 < len2; r0 += 64) {
            const int r = r0 + lane;
            if (r < len2) {
                const uint32_t L = note_l[r];
                if (L & 1) gaps++;
                else if (seq1[note_c[r]] == seq2[r]) matches++;
                *reinterpret_cast<uint16_t*>(rec + SA_LONG_HEADER_BYTES + 2 * (size_t)r) = (uint16_t)L;
            }
        }
        // body bytes past the last row are zero, so that equal inputs give equal records
        for (uint32_t o = SA_LONG_HEADER_BYTES + 2 * (uint32_t)len2 + 2 * lane; o + 1 < P.stride; o += 128)
            *reinterpret_cast<uint16_t*>(rec + o) = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            matches += __shfl_xor(matches, o, 64);
            gaps += __shfl_xor(gaps, o, 64);
        }
        if (lane == 0) {
            const int score = dfinal + (len1 + len2) * P.gap;
            *reinterpret_cast<uint4*>(rec) = make_uint4((uint32_t)score, (uint32_t)matches, (uint32_t)gaps, 0u);
        }
        __syncthreads();          // the notes and blocks are reused by the next pair
    }
}

// ------------------------------------------------------------------------------------------------------------
// A workgroup of W waves per pair, for submissions with fewer pairs than the chip has SIMDs (64 adjacent pairs of
// 4096 residues on one wave each keep 64 of 1024 SIMDs busy).  Wave w sweeps strips w, w + W, w + 2 W ... exactly as
// above; what changes is how a strip's last column reaches the strip to its right.  It goes through LDS, 64 rows (one
// block) at a time, while both strips are being swept: the wave of strip s + 1 runs two blocks behind the wave of
// strip s (it needs rows 64 b + 1 .. 64 b + 64 for its block b, which strip s completes with its block b + 1).
//
// Hand-over: boundary q = s mod W has a ring of SA_COOP_RING blocks and two counters, both counting blocks over all
// the rounds of the pair: produced[q] (written by the wave of strip s) and consumed[q] (written by the wave of strip
// s + 1).  The producer writes a block's 64 values, then the counter (release); the consumer waits for the counter
// (acquire), reads the values, then writes its own counter (release); the producer does not overwrite a slot before
// the block in it is consumed (acquire of that counter).
// The boundary between the last wave and wave 0 is different: wave 0 only comes to strip W r after it has finished
// strip W (r - 1), a whole strip later, so what wave W - 1 hands it must wait that long.  A ring there would stop wave
// W - 1 after a few blocks, then wave W - 2 behind it, and so on down to wave 0 itself -- for ever, once a strip has
// more blocks than the rings hold together.  That boundary therefore keeps the whole column (in LDS too, sized by the
// longest sequence, one copy per parity of the round: wave W - 1 writes round r while wave 0 reads round r - 1) and
// its producer never waits.  With it no wave waits in a circle: wave W - 1 depends on wave W - 2's blocks only, wave
// W - 2 on wave W - 3's blocks and on wave W - 1 taking them, and so on; wave 0 waits for columns of the round before.
//
// The walk and the record's header are wave 0's; the per-row work of the record is shared by all waves.
#define SA_COOP_RING 4

// (volatile accesses to LDS are compiled as flat accesses, which wait for every global store in flight; workgroup-scope
// atomics stay LDS instructions and cost one wait for the LDS queue)
#define SA_LDS_ACQUIRE(X) ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&(X), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)))
#define SA_LDS_RELEASE(X, V) __hip_atomic_store(&(X), (V), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)

// Dynamic LDS of sa_coop_wg_kernel, byte offsets (slot = padded sequence length, steps = slot + 128):
//   row residues  slot + 208 bytes: sequence 2 with 64 codes of "above the matrix" before it and as many as the steps
//                 (and the requests that run ahead of them) reach after it
//   zeros         64 dwords: the border column of the first strip and of rows below the matrix
//   whole columns 2 x (64 + steps) dwords (64 in front: the rows above the matrix that lane 63 also computes)
//   rings         W x SA_COOP_RING x 64 dwords
//   lane tables   W x 22 x 64 dwords: every lane's own row of the score table, entry b of lane l at (b * 64 + l) * 4,
//                 so that a wave's 64 reads go to 64 different addresses in bank order whatever the residues are
__host__ __device__ inline uint32_t sa_cwg_rowres_bytes(uint32_t lslot) { return (lslot + 208u) & ~15u; }
__host__ __device__ inline uint32_t sa_cwg_lds_bytes(uint32_t lslot, uint32_t waves) {
    return sa_cwg_rowres_bytes(lslot) + 256u + 2u * (64u + lslot + 128u) * 4u + waves * SA_COOP_RING * 256u + waves * 22u * 256u;
}

template <int W>
__global__ void __launch_bounds__(64 * W) sa_coop_wg_kernel(const SaPepParams P) {
    __shared__ int32_t tab[21 * 32];
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint32_t produced[W], consumed[W];
    __shared__ uint32_t item_slot;
    __shared__ int dfinal_slot, match_slot, gap_slot;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lslot = P.lslot;
    const uint32_t strips_cap = lslot >> 6;
    const uint32_t steps_cap = lslot + 128;
    const uint32_t blocks_cap = steps_cap >> 4;
    const uint32_t off_zeros = sa_cwg_rowres_bytes(lslot);
    const uint32_t off_wrap = off_zeros + 256;
    const uint32_t wrap_bytes = (64 + steps_cap) * 4;
    const uint32_t off_ring = off_wrap + 2 * wrap_bytes;
    const uint32_t off_ltab = off_ring + W * SA_COOP_RING * 256;
    // per-workgroup scratch as in sa_coop_kernel (the two border columns of that layout are not used here)
    uint8_t* const area = reinterpret_cast<uint8_t*>(P.spill) + (size_t)blockIdx.x * P.spill_rows * 1024;
    uint32_t* const dirs = reinterpret_cast<uint32_t*>(area);
    // the walk's notes: one dword per row (moves of the row, column it was left in), where sa_coop_kernel keeps two
    // 16-bit arrays of the same total size
    uint16_t* const note_l = reinterpret_cast<uint16_t*>(area + (size_t)strips_cap * blocks_cap * 256 + 2 * (size_t)steps_cap * 4);

    if (threadIdx.x < 64) reinterpret_cast<int32_t*>(smem + off_zeros)[threadIdx.x] = 0;
    for (int e = threadIdx.x; e < 21 * 32; e += 64 * W) {
        const int a = e >> 5, b = e & 31;
        int v = SA_TAG_TABLE(P.sprime[a * SA_SPRIME_STRIDE + (b <= 20 ? b : 20)]) + 1;
        if (b == SA_COOP_NEG_CODE) v = -(1 << 29);
        tab[e] = v;
    }

    for (;;) {
        if (threadIdx.x == 0) { item_slot = atomicAdd(P.counter, 1u); match_slot = 0; gap_slot = 0; }
        if (threadIdx.x < W) { produced[threadIdx.x] = 0; consumed[threadIdx.x] = 0; }
        __syncthreads();
        const uint32_t it = __builtin_amdgcn_readfirstlane(item_slot);
        if (it >= P.nitems) break;
        const SaItem item = P.items[it];
        const uint32_t s1 = item.p, s2 = item.v0;
        const int len1 = __builtin_amdgcn_readfirstlane((int)P.len_fo[s1]);
        const int len2 = __builtin_amdgcn_readfirstlane((int)P.len_fo[s2]);
        const uint8_t* const seq1 = P.codes_fo + (size_t)s1 * lslot;
        const uint8_t* const seq2 = P.codes_fo + (size_t)s2 * lslot;
        uint8_t* const rec = P.out + (((uint64_t)item.c0 << 32) | item.b0) * P.stride;
        const int nstrips = (len1 + 63) >> 6;
        const int nsuper = (len2 + 63 + 63) >> 6;
        const int lstar = (len1 - 1) & 63;

        // sequence 2 for every wave: code of row r at byte 64 + r; outside the matrix the code whose scores are hugely negative
        for (uint32_t o = threadIdx.x; o < off_zeros; o += 64 * W) {
            const int r = (int)o - 64;
            smem[o] = (r >= 0 && r < len2) ? seq2[r] : (uint8_t)SA_COOP_NEG_CODE;
        }
        __syncthreads();

        // ------------------------------------------------------------------ fill: this wave's strips
        // A step costs nine vector instructions (table address, DIAG sum, maximum, tag shifted in, clean value, UP
        // operand, LEFT operand by DPP, lane 63's value into the column register by v_readlane + v_writelane) and three
        // LDS reads, each with a literal offset off a per-block address: the row residue (a byte, four steps ahead), the
        // table entry (two steps ahead), and the border value of the row lane 0 comes to next (the same address for all
        // lanes; it lands in the register that the DPP move then fills for lanes 1 .. 63).  Nothing travels from lane to
        // lane but the LEFT operand.  Sixteen waves issue these nine at four cycles each: 9 cycles per step and CU, which
        // is what the fill of a 4096 x 4096 pair takes (270 000 steps, 1.17 ms).  With the last-column value stored to LDS
        // at every step instead (seven vector, four LDS instructions) the LDS is the limit: 1.38 ms.
        const int qin = (wv + W - 1) % W;                      // boundary on this wave's left
        const bool wrap_in = wv == 0, wrap_out = wv == W - 1;  // ... which keeps whole columns for wave 0, written by wave W - 1
        const uint32_t ltab_lane = off_ltab + (uint32_t)wv * 22 * 256 + (uint32_t)lane * 4;
        for (int s = wv; s < nstrips; s += W) {
            {   // this lane's row of the table
                const int a = (int)seq1[64 * s + lane];
#pragma unroll
                for (int b = 0; b < 22; b++) *reinterpret_cast<int32_t*>(smem + ltab_lane + b * 256) = tab[a * 32 + b];
            }
            const bool first_strip = s == 0, more = s + 1 < nstrips;
            const uint32_t gin0 = first_strip ? 0u : (uint32_t)((s - 1) / W) * (uint32_t)nsuper;     // block numbers of strip s - 1
            const uint32_t gout0 = (uint32_t)(s / W) * (uint32_t)nsuper;                              // ... and of this strip
            const uint32_t wrap_rd = off_wrap + (first_strip ? 0u : (uint32_t)(((s - 1) / W) & 1)) * wrap_bytes + 256;   // row 0
            const uint32_t wrap_wr = off_wrap + (uint32_t)((s / W) & 1) * wrap_bytes + 256;
            int up = 1;
            int left = 0;
            if (!first_strip) {
                // row 0 of the column on the left: entry 63 of its block 0
                while (SA_LDS_ACQUIRE(produced[qin]) < gin0 + 1) __builtin_amdgcn_s_sleep(1);
                const int row0 = *reinterpret_cast<const int32_t*>(smem + (wrap_in ? wrap_rd : off_ring + ((uint32_t)qin * SA_COOP_RING + gin0 % SA_COOP_RING) * 256 + 63 * 4));
                if (lane == 0) {
                    left = row0;
                    if (!wrap_in) SA_LDS_RELEASE(consumed[qin], gin0 + 1);
                }
            }
            int left_prev = 0;
            uint32_t acc = 0;
            int colbuf = 0;
            // residues of steps 2 and 3 and table entries of steps 0 and 1 are under way when the first step begins
            uint32_t res_off = 64u - (uint32_t)lane;            // byte of the residue of step 64 sb for this lane
            int Tq0, Tq1, Ba, Bb;
            {
                const int b0 = smem[res_off], b1 = smem[res_off + 1];
                Ba = smem[res_off + 2];
                Bb = smem[res_off + 3];
                Tq0 = *reinterpret_cast<const int32_t*>(smem + ltab_lane + (b0 << 8));
                Tq1 = *reinterpret_cast<const int32_t*>(smem + ltab_lane + (b1 << 8));
            }
            for (int sb = 0; sb < nsuper; sb++) {
                // border values of rows 64 sb + 1 .. 64 sb + 64: block sb + 1 of the column on the left (a block past its
                // last one would hold rows below the matrix only: zeros do there, as in the first strip)
                uint32_t bord_off = off_zeros, gin = 0;
                bool give_back = false;
                if (!first_strip && sb + 1 < nsuper) {
                    gin = gin0 + (uint32_t)sb + 1;
                    while (SA_LDS_ACQUIRE(produced[qin]) < gin + 1) __builtin_amdgcn_s_sleep(1);
                    bord_off = wrap_in ? wrap_rd + (64u * sb + 1) * 4 : off_ring + ((uint32_t)qin * SA_COOP_RING + gin % SA_COOP_RING) * 256;
                    give_back = !wrap_in;
                }
                const uint32_t gout = gout0 + (uint32_t)sb;
                int Xq0 = *reinterpret_cast<const int32_t*>(smem + bord_off);
                int Xq1 = *reinterpret_cast<const int32_t*>(smem + bord_off + 4);
#define SA_CWG_STEP(TL)                                                                                               \
    {                                                                                                                 \
        const int Bn = smem[res_off + (TL) + 4];                                                                      \
        const int Tn = *reinterpret_cast<const int32_t*>(smem + ltab_lane + ((uint32_t)Ba << 8));                     \
        int Xn = 0;                                                                                                   \
        if ((TL) + 2 < 64) Xn = *reinterpret_cast<const int32_t*>(smem + bord_off + 4 * ((TL) + 2));                  \
        const int m = left_prev + Tq0;                                                                                \
        const int h = max3i(m, up, left);                                                                             \
        acc = __builtin_amdgcn_alignbit((uint32_t)h, acc, 2);                                                         \
        const int zl = h & ~3;                                                                                        \
        up = zl | 1;                                                                                                  \
        /* the strip's last column, row t - 63, for the strip on the right */                                         \
        const int last = __builtin_amdgcn_readlane(zl, 63);                                                           \
        SA_WRITELANE_CONST(colbuf, last, (TL));                                                                       \
        left_prev = left;                                                                                             \
        /* next step's LEFT: from the lane on the left; lane 0 keeps the border value the register already holds */   \
        left = __builtin_amdgcn_update_dpp(Xq0, zl, SA_COOP_DPP_WAVE_SHR1, 0xf, 0xf, false);                          \
        Tq0 = Tq1; Tq1 = Tn; Ba = Bb; Bb = Bn; Xq0 = Xq1; Xq1 = Xn;                                                   \
    }
#define SA_CWG_BLOCK(B)                                                                                               \
    SA_CWG_STEP(16 * (B) + 0) SA_CWG_STEP(16 * (B) + 1) SA_CWG_STEP(16 * (B) + 2) SA_CWG_STEP(16 * (B) + 3)           \
    SA_CWG_STEP(16 * (B) + 4) SA_CWG_STEP(16 * (B) + 5) SA_CWG_STEP(16 * (B) + 6) SA_CWG_STEP(16 * (B) + 7)           \
    SA_CWG_STEP(16 * (B) + 8) SA_CWG_STEP(16 * (B) + 9) SA_CWG_STEP(16 * (B) + 10) SA_CWG_STEP(16 * (B) + 11)         \
    SA_CWG_STEP(16 * (B) + 12) SA_CWG_STEP(16 * (B) + 13) SA_CWG_STEP(16 * (B) + 14) SA_CWG_STEP(16 * (B) + 15)       \
    dirs[((size_t)s * blocks_cap + 4 * sb + (B)) * 64 + lane] = acc;
                SA_CWG_BLOCK(0) SA_CWG_BLOCK(1) SA_CWG_BLOCK(2) SA_CWG_BLOCK(3)
#undef SA_CWG_BLOCK
#undef SA_CWG_STEP
                res_off += 64;
                // colbuf lane i = last column of row 64 sb + i - 63: block sb of the boundary on the right
                if (more) {
                    uint32_t mine;
                    if (wrap_out) {
                        mine = wrap_wr + (uint32_t)((64 * sb - 63) * 4);
                    } else {
                        if (gout >= SA_COOP_RING)
                            while (SA_LDS_ACQUIRE(consumed[wv]) < gout - SA_COOP_RING + 1) __builtin_amdgcn_s_sleep(1);
                        mine = off_ring + ((uint32_t)wv * SA_COOP_RING + gout % SA_COOP_RING) * 256;
                    }
                    *reinterpret_cast<int32_t*>(smem + mine + (uint32_t)lane * 4) = colbuf;
                }
                // the block is written: the wave on the right may read it; the block read is free for its writer
                if (lane == 0) {
                    if (more) SA_LDS_RELEASE(produced[wv], gout + 1);
                    if (give_back) SA_LDS_RELEASE(consumed[qin], gin + 1);
                }
            }
            if (s == nstrips - 1 && lane == 0) dfinal_slot = __builtin_amdgcn_readlane(up, lstar) >> 2;
        }
        __syncthreads();      // every direction block is written

        // ------------------------------------------------------------- traceback: wave 0, as in sa_coop_kernel, except
        // that direction blocks are loaded eight at a time (blocks k, k - 1 .. k - 7 of the strip: within a strip t only
        // ever decreases, and a diagonal path crosses a strip in about eight blocks) -- one wait for memory per window,
        // not per block --, that a row's note goes to memory at once, as one dword (moves, column), and that runs of DIAG
        // moves are taken several moves at a time
        if (wv == 0) {
            int i = len2, j = len1, run = 0;
#ifdef SA_COOP_EXP_NO_WALK
            i = 0;       // experiment: the fill alone (records are wrong)
#endif
            int cur_s = -1, cur_k = -1, win_top = -1;
            uint32_t w = 0;                // block cur_k of strip cur_s
            uint32_t win[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // blocks win_top .. win_top - 7 of strip cur_s
            uint32_t* const notes = reinterpret_cast<uint32_t*>(note_l);
            while (i > 0) {
                int tag = SA_TAG_UP;
                int l = 0, t = 0;
                if (j > 0) {
                    const int c = j - 1, sidx = c >> 6, k = (i - 1 + (c & 63)) >> 4;
                    l = c & 63;
                    t = i - 1 + l;
                    if (sidx != cur_s || k != cur_k) {
                        if (sidx != cur_s || k > win_top || k < win_top - 7) {
                            const uint32_t* const strip_blocks = dirs + (size_t)sidx * blocks_cap * 64 + lane;
#pragma unroll
                            for (int q = 0; q < 8; q++) win[q] = strip_blocks[(size_t)max(k - q, 0) * 64];
                            win_top = k;
                        }
                        switch (win_top - k) {
                        case 0: w = win[0]; break;
                        case 1: w = win[1]; break;
                        case 2: w = win[2]; break;
                        case 3: w = win[3]; break;
                        case 4: w = win[4]; break;
                        case 5: w = win[5]; break;
                        case 6: w = win[6]; break;
                        default: w = win[7]; break;
                        }
                        cur_s = sidx; cur_k = k;
                    }
                    tag = (int)((__builtin_amdgcn_readlane(w, l) >> (2 * (t & 15))) & 3u);
                }
                if (tag == SA_TAG_LEFT) { run++; j--; continue; }
                if (tag == SA_TAG_DIAG) {
                    // A run of DIAG moves in one go, as far as it stays in this block and strip: move m of the run
                    // stands in lane l - m at step t - 2 m, so every lane looks at its own candidate and the run's
                    // length is the number of consecutive hits from lane l downwards.  Lane m notes row i - 1 - m.
                    const int mmax = min(min(l, (t & 15) >> 1), i - 1);
                    const int m = l - lane;
                    const uint32_t sh = (uint32_t)(2 * ((t - 2 * m) & 15));
                    const bool hit = m >= 0 && m <= mmax && ((w >> sh) & 3u) == (uint32_t)SA_TAG_DIAG;
                    const unsigned long long hits = __ballot(hit);                    // bit l is set: the move just read
                    const int cnt = __builtin_clzll(~(hits << (63 - l)));
                    if (lane < cnt) notes[i - 1 - lane] = (lane == 0 ? 2u * (uint32_t)run : 0u) | ((uint32_t)(j - 1 - lane) << 16);
                    i -= cnt;
                    j -= cnt;
                    run = 0;
                    continue;
                }
                if (lane == 0) notes[i - 1] = (uint32_t)(2 * run + 1) | ((uint32_t)(j > 0 ? j - 1 : 0) << 16);
                i--;
                run = 0;
            }
        }
        __syncthreads();

        // ------------------------------------------------------ record: 64 rows per wave at a time
        int matches = 0, gaps = 0;
        for (int r0 = 64 * wv; r0 < len2; r0 += 64 * W) {
            const int r = r0 + lane;
            if (r < len2) {
                const uint32_t note = reinterpret_cast<const uint32_t*>(note_l)[r];
                if (note & 1) gaps++;
                else if (seq1[note >> 16] == seq2[r]) matches++;
                *reinterpret_cast<uint16_t*>(rec + SA_LONG_HEADER_BYTES + 2 * (size_t)r) = (uint16_t)note;
            }
        }
        for (uint32_t o = SA_LONG_HEADER_BYTES + 2 * (uint32_t)len2 + 2 * threadIdx.x; o + 1 < P.stride; o += 128 * W)
            *reinterpret_cast<uint16_t*>(rec + o) = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            matches += __shfl_xor(matches, o, 64);
            gaps += __shfl_xor(gaps, o, 64);
        }
        if (lane == 0) { atomicAdd(&match_slot, matches); atomicAdd(&gap_slot, gaps); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int score = dfinal_slot + (len1 + len2) * P.gap;
            *reinterpret_cast<uint4*>(rec) = make_uint4((uint32_t)score, (uint32_t)match_slot, (uint32_t)gap_slot, 0u);
        }
        __syncthreads();          // the slots, notes and blocks are reused by the next pair
    }
}

extern "C" void sa_launch_coop(const SaPepParams* p, unsigned grid, hipStream_t stream) {
    hipLaunchKernelGGL(sa_coop_kernel, dim3(grid), dim3(64), 0, stream, *p);
}

// dynamic LDS of a workgroup of `waves` waves for sequences padded to `lslot` (the kernel has 3 KiB of its own on top)
extern "C" unsigned sa_coop_wg_lds_bytes(unsigned lslot, int waves) { return sa_cwg_lds_bytes(lslot, (uint32_t)waves); }

// waves = 2, 4, 8 or 16 waves per pair
extern "C" void sa_launch_coop_wg(int waves, const SaPepParams* p, unsigned grid, hipStream_t stream) {
    const unsigned lds = sa_cwg_lds_bytes(p->lslot, (uint32_t)waves);
    // dynamic LDS beyond the default ceiling (64 KiB) has to be asked for; asked from 48 KiB up to stay clear of it
    // (per device, so at every launch; a refusal shows at the launch)
    if (lds > 48u * 1024u) {
        const void* f = waves == 2 ? (const void*)sa_coop_wg_kernel<2> : waves == 4 ? (const void*)sa_coop_wg_kernel<4>
                      : waves == 8 ? (const void*)sa_coop_wg_kernel<8> : (const void*)sa_coop_wg_kernel<16>;
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    switch (waves) {
    case 2: hipLaunchKernelGGL(sa_coop_wg_kernel<2>, dim3(grid), dim3(128), lds, stream, *p); break;
    case 4: hipLaunchKernelGGL(sa_coop_wg_kernel<4>, dim3(grid), dim3(256), lds, stream, *p); break;
    case 8: hipLaunchKernelGGL(sa_coop_wg_kernel<8>, dim3(grid), dim3(512), lds, stream, *p); break;
    default: hipLaunchKernelGGL(sa_coop_wg_kernel<16>, dim3(grid), dim3(1024), lds, stream, *p); break;
    }
}
