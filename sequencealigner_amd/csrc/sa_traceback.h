// sa_traceback.h -- traceback over rows of direction tags, shared by the peptide
// kernels (sa_kernels.hip, sa_kernels_pair.hip).
#ifndef SA_TRACEBACK_H
#define SA_TRACEBACK_H

#include "sa_kernel_common.h"

// Buffer descriptors (base, byte count) for the arrays the row loops touch: a buffer
// access takes a wave-uniform byte offset in an SGPR and a per-lane 32-bit offset in a
// VGPR, so the row loops spend no vector instructions on addresses; accesses past the
// byte count are dropped by the hardware.
typedef uint32_t sa_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t sa_u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t sa_u32x4 __attribute__((ext_vector_type(4)));
#include <type_traits>
#define SA_BUFFER_FLAGS 0x00020000

// ---------------------------------------------------------------- traceback
// Bottom row first, four rows per chunk.  The direction words of a chunk are
// requested two chunks ahead of their use (global spill area, coalesced).  A row
// is W = ceil(ncols/16) dwords per lane: the tag of column c is bits 2(c%16), 2(c%16)+1 of dword c/16,
// tags 0 LEFT, 1 UP, 3 DIAG; a run of LEFT moves is found with count-leading-zeros, and the zero count
// is 2 * (run of LEFT) + (1 if an UP ends it, 0 if a DIAG does).  (The paired kernel stores lower columns
// higher up, tags 0 LEFT, 2 UP, 3 DIAG, and walks with count-trailing-zeros: traceback_pk below.)
// base_off is the byte offset of the pair's rows inside the spill buffer.
// Row residues (for counting matches): TRIANGULAR keeps the 64 residue codes of
// the lane's second sequence in 16 registers (bw), ADJACENT reads them four at a
// time from the sequence's slot.
struct SaChunk {
    uint4 d[4];
    uint32_t b4;    // ADJACENT: residues of the four rows
};

#define SA_WORD_CASE(K) case K: v = W[K]; break;
__device__ __forceinline__ uint32_t pick_word(const uint32_t (&W)[16], const int k) {   // k is wave-uniform
    uint32_t v = 0;
    switch (k) {
        SA_WORD_CASE(0) SA_WORD_CASE(1) SA_WORD_CASE(2) SA_WORD_CASE(3) SA_WORD_CASE(4) SA_WORD_CASE(5)
        SA_WORD_CASE(6) SA_WORD_CASE(7) SA_WORD_CASE(8) SA_WORD_CASE(9) SA_WORD_CASE(10) SA_WORD_CASE(11)
        SA_WORD_CASE(12) SA_WORD_CASE(13) SA_WORD_CASE(14) SA_WORD_CASE(15)
        default: break;
    }
    return v;
}

template <int MODE, int W>
__device__ __forceinline__ void traceback(const int lane, const int maxlen, const int lenq, const int lenp,
                                          const __amdgpu_buffer_rsrc_t spill, const int base_off,
                                          const uint32_t (&bw)[16], const uint8_t* __restrict__ brow,
                                          const uint8_t* __restrict__ acol, const uint64_t* __restrict__ emask,
                                          uint32_t* __restrict__ stage_rec, int& matches_out, int& gaps_out) {
    // j2 = twice the number of columns of sequence 1 still to the left of the walk: the tags are two bits
    // wide, so shift amounts, run lengths and the output byte (run * 2 + up) all come in this unit
    int j2 = 2 * lenp, matches = 0, gaps = 0;
    const int lane_off = lane * W * 4;

    // Nothing loaded here is used before process().  Row indices past the end of
    // the longest sequence are clamped, so the requests need no guard.
    auto load_chunk = [&](int ch, SaChunk& ck) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = min(4 * ch + k, maxlen - 1);   // wave-uniform
            const int row_off = base_off + r * 256 * W;
            if (W == 4) {
                const sa_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(spill, lane_off, row_off, 0);
                ck.d[k] = make_uint4(v.x, v.y, v.z, v.w);
            } else if (W == 3) {
                const sa_u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(spill, lane_off, row_off, 0);
                ck.d[k] = make_uint4(v.x, v.y, v.z, 0);
            } else if (W == 2) {
                const sa_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(spill, lane_off, row_off, 0);
                ck.d[k] = make_uint4(v.x, v.y, 0, 0);
            } else {
                ck.d[k] = make_uint4(__builtin_amdgcn_raw_buffer_load_b32(spill, lane_off, row_off, 0), 0, 0, 0);
            }
        }
        ck.b4 = (MODE == SA_MODE_ADJ) ? reinterpret_cast<const uint32_t*>(brow)[ch] : 0;
    };

    auto process = [&](int ch, const SaChunk& ck) {
        const uint32_t b4 = (MODE == SA_MODE_TRI) ? pick_word(bw, ch) : ck.b4;
        uint64_t e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = (MODE == SA_MODE_TRI) ? emask[(b4 >> (8 * k)) & 0xff] : 0;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const int r = 4 * ch + k;
            if (r < lenq) {
                // The tags of columns j-1, j-2, ... are moved to the top of a window, most recent
                // column first; L = leading zeros = 2 * (run of LEFT) + (1 if an UP ends it, 0 if a
                // DIAG does).  Branch-free per lane: with j == 0 the window is meaningless, the run
                // is cut to 0 and the move is UP (left border).
                int L;
                const uint32_t sh = 0u - (uint32_t)j2;       // (32 or 64 - j2) modulo the width of the shift
                if (W == 1) {
                    const uint32_t x = ck.d[k].x << (sh & 31);
                    L = x ? __builtin_clz(x) : 32;
                } else if (W == 2) {
                    const uint64_t v = ((uint64_t)ck.d[k].y << 32) | ck.d[k].x;
                    const uint64_t x = v << (sh & 63);
                    L = x ? __builtin_clzll(x) : 64;
                } else {
                    // columns 32..63 in one 64-bit half, 0..31 in the other; a run that starts in the
                    // upper half and finds nothing there goes on from column 31
                    const uint64_t v0 = ((uint64_t)ck.d[k].y << 32) | ck.d[k].x;
                    const uint64_t v1 = ((uint64_t)ck.d[k].w << 32) | ck.d[k].z;
                    const bool upper = j2 > 64;
                    const uint64_t x = (upper ? v1 : v0) << (sh & 63);
                    const int l0 = v0 ? __builtin_clzll(v0) : 64;
                    L = x ? __builtin_clzll(x) : (upper ? j2 - 64 + l0 : 64);
                }
                const int run2 = min(L & ~1, j2);                          // twice the run of LEFT moves
                j2 -= run2;
                const int t = (j2 > 0) ? (L & 1) : 1;                      // 0 = DIAG, 1 = UP (left border: UP)
                const int c1 = ((j2 >> 1) - 1) & 63;
                const int dg = t ^ 1;
                int eq;
                if (MODE == SA_MODE_TRI) eq = (W > 2) ? (int)((e[k] >> c1) & 1) : (int)(((uint32_t)e[k] >> (c1 & 31)) & 1);
                else eq = (t == 0 && (uint32_t)acol[c1] == ((b4 >> (8 * k)) & 0xff)) ? 1 : 0;
                matches += eq & dg;
                j2 -= 2 * dg;
                gaps += t;
                packed |= (uint32_t)(run2 + t) << (8 * k);
            }
        }
        stage_rec[2 + ch] = packed;
    };

    SaChunk c0, c1, c2;
    int ch = (maxlen - 1) >> 2;
    load_chunk(ch, c0);
    if (ch > 0) load_chunk(ch - 1, c1);
    for (; ch >= 0; ch -= 3) {
        if (ch > 1) load_chunk(ch - 2, c2);
        process(ch, c0);
        if (ch > 0) {
            if (ch > 2) load_chunk(ch - 3, c0);
            process(ch - 1, c1);
        }
        if (ch > 1) {
            if (ch > 3) load_chunk(ch - 4, c1);
            process(ch - 2, c2);
        }
    }
    // body bytes past the last chunk are zero, so that equal inputs give equal records
    for (int z = ((maxlen - 1) >> 2) + 1; z < SA_LDS_STAGE_DWORDS - 2; z++) stage_rec[2 + z] = 0;
    matches_out = matches;
    gaps_out = gaps;
}

// ---------------------------------------------------------------- traceback, paired peptide kernel
// The same walk for the paired kernel's layout (lower columns higher up, count-trailing-zeros), with fewer
// instructions per row.
//  * State s = 2 * (columns of sequence 1 still to the left) + 1.  With L = min(trailing zeros of the window, s)
//    the run of LEFT moves is L & ~1 and the move that ends it is UP when L is odd: a walk that reaches the left
//    border finds L = s, which is odd.  (On the border itself, s = 1, the shift amount 64 - 2 * columns wraps to
//    0 and the window holds the unshifted row: there L is forced to 1 by a lower bound, 2 - s, that is negative elsewhere.)  L is
//    also the output byte (2 * run + up).
//  * Matches: the first sequence's residue codes lie in LDS two bytes apart (column c at pcol[2c]), so the byte
//    address of the column a DIAG move consumes is s - L - 3, no shift; an UP move reads an odd address, one of the
//    bytes in between (or of the three bytes in front of column 0), which hold 0xff, no residue code: the
//    comparison with the row's residue fails without a look at the kind of move.  (Handing the value read to the next row's block, so
//    that nothing waits for the LDS, made no difference on the same machine and was dropped.)
//  * The dwords of a direction row are stored last first (dword m = columns 16(W-1-m) .. 16(W-1-m)+15), so that
//    the register pair a 64-bit load fills is the 64-bit window value, lower columns in the upper half.
//  * Gaps (the record counts the columns where sequence 1 has '-', the UP moves) are not counted per row but
//    per chunk: they are the odd bytes of the record.
__device__ __forceinline__ uint32_t sa_ffbl(uint32_t x) {      // trailing zeros; 0xffffffff for x = 0
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t sa_ctz64_sat(uint64_t x) { // trailing zeros of a 64-bit value; >= 64 for x = 0
    uint32_t fl, fh;
    asm("v_ffbl_b32 %0, %2\n\tv_ffbl_b32 %1, %3\n\tv_add_u32 %1, %1, 32 clamp"
        : "=&v"(fl), "=&v"(fh) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)));
    return min(fl, fh);
}

__device__ __forceinline__ uint32_t sa_ctz64_min(uint64_t x, uint32_t cap) { // min(trailing zeros of x, cap)
    uint32_t fl, fh;
    asm("v_ffbl_b32 %0, %2\n\tv_ffbl_b32 %1, %3\n\tv_add_u32 %1, %1, 32 clamp\n\tv_min3_u32 %0, %0, %1, %4"
        : "=&v"(fl), "=&v"(fh) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "v"(cap));
    return fl;
}

// PCOL_LDS_OFFSET: byte address of column 0's residue code inside the workgroup's LDS (at least 3);
// STAGE_DWORDS: dwords of the lane's staging record (2 of header, the rest one byte per row).
//  * The direction words of a chunk (four rows) are requested a chunk ahead of their use, two register sets in turn.
//    (Hand-written requests with counted waits, "at most 8 outstanding", changed nothing on the same machine with four
//    waves per SIMD, and were dropped.)
// Register sets of the traceback's row requests, by row width in dwords.  Three to six sets made no difference in time;
// with two, a chunk is requested one chunk (four rows) before its use instead of two, so the decision which areas of
// a row still have to be requested is made on a later state of the walk: 296 -> 286 GB read per launch of 50 000
// peptides, and 0.6 % less time.
#ifndef SA_TB_SETS_W1
#define SA_TB_SETS_W1 2
#define SA_TB_SETS_W2 2
#define SA_TB_SETS_W3 2
#define SA_TB_SETS_W4 2
#endif
// Dwords of a row that go to the area of the higher columns.  A row of two dwords is not split: as one 8-byte
// access per lane, sixteen lanes fill a 128-byte line, where two 4-byte accesses each return to a line whose
// first half is still on its way (102.7-103.1 -> 102.1-102.2 ms on one box, with 2 % more bytes read).
#ifndef SA_DIR_WH
#define SA_DIR_WH(WD) ((WD) >= 3 ? (WD) / 2 : 0)
#endif
template <int W> struct SaDirWords;
template <> struct SaDirWords<1> { typedef uint32_t type; };
template <> struct SaDirWords<2> { typedef sa_u32x2 type; };
template <> struct SaDirWords<3> { typedef sa_u32x3 type; };
template <> struct SaDirWords<4> { typedef sa_u32x4 type; };

// Scalar or vector of W dwords, element i
template <int W, typename WORDS>
__device__ __forceinline__ uint32_t sa_tb_elem(const WORDS& d, const int i) {
    if constexpr (W <= 1) return d;
    else return d[i];
}

// A direction row of one pair is 8-column pieces of 16 bits, N16 = ceil(columns / 8) of them, that is WD = N16 / 2
// dwords and, for odd N16 (ODD = 1), one more piece with the row's last eight columns.  In memory a row of the wave is
// three areas: the WH = SA_DIR_WH(WD) dwords of the higher columns of every lane (64 lanes side by side), the WL = WD - WH
// dwords of the lower columns, the pieces (128 bytes).  The walk moves towards column 0 and never looks at a column
// above the one it stands in: once every lane of the wave has left the higher columns their area is not requested any
// more (a third of the row requests' bytes on the bench set), and likewise the pieces.
// In registers the row is W = WD + ODD dwords, highest columns in dword 0: the piece shifted to the upper half of
// dword 0 (its lower half stands for columns past the row's end), then the higher area's dwords, then the lower's.
// PB: bytes of a lane's odd piece in memory, 2 or 1 -- when the row's last group of four columns stands alone in the
// piece, only the piece's upper byte is stored.
// CLAMP: the walk's lower bound on L at the left border (see above) is applied.  It is needed only when bit 0 of
// the window's lowest dword can be set: the window then holds a column in its last place (rows of 16, 32, 48 or 64
// columns, or a computed padding column there) or is wider than 64 bits.  Everywhere else that bit is 0, the
// unshifted window at the border has at least two trailing zeros and min(L, s = 1) is 1 by itself.
template <int WD, int ODD, int PB, bool CLAMP, int PCOL_LDS_OFFSET, int STAGE_DWORDS>
__device__ __forceinline__ void traceback_pk(const int lane, const int maxlen, const int lenq, const int lenp,
                                             const __amdgpu_buffer_rsrc_t spill, const int base_off,
                                             const uint32_t (&bw)[16],
                                             uint32_t* __restrict__ stage_rec, int& matches_out, int& gaps_out) {
    constexpr int W = WD + ODD;
    constexpr int WH = SA_DIR_WH(WD), WL = WD - WH;
    constexpr int NS = W <= 1 ? SA_TB_SETS_W1 : W == 2 ? SA_TB_SETS_W2 : W == 3 ? SA_TB_SETS_W3 : SA_TB_SETS_W4;   // register sets
    constexpr int ROW_BYTES = 64 * (4 * WD + PB * ODD);
    typedef typename SaDirWords<WL ? WL : 1>::type low_t;
    typedef typename SaDirWords<WH ? WH : 1>::type high_t;
    struct Chunk { low_t lo[4]; high_t hi[4]; uint32_t piece[4]; };
    // lanes without a pair keep state 0: they neither walk nor hold the wave in the higher columns
    uint32_t s = lenq > 0 ? 2u * (uint32_t)lenp + 1u : 0u, matches = 0, ups = 0;
    const int high_off = lane * WH * 4;
    const int low_off = 256 * WH + lane * WL * 4;
    const int piece_off = 256 * WD + lane * PB;

    // Rows of chunk ch; the state s = 2 * (column + 1) + 1 only falls, so what no lane needs now it will not need later.
    // Every call issues the same loads: an area that is not wanted (or a chunk above the first or below the last row)
    // is asked for at an offset beyond the buffer's size, where the hardware fetches nothing and returns 0.  With a
    // branch around a request instead, the compiler cannot count the requests that follow a chunk's own and waits for
    // all of them -- the chunk just requested included -- before it touches a row.
    constexpr int SA_TB_BEYOND = 0x7ffff000;
    auto load_chunk = [&](int ch, Chunk& ck) {
        const bool exists = ch >= 0;
        const bool need_high = WH > 0 && exists && __any(s >= 32u * WL + 3u);
        const bool need_piece = ODD == 1 && exists && __any(s >= 32u * WD + 3u);
        const int lo_at = exists ? low_off : SA_TB_BEYOND;
        const int hi_at = need_high ? high_off : SA_TB_BEYOND;
        const int piece_at = need_piece ? piece_off : SA_TB_BEYOND;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = max(min(4 * ch + k, maxlen - 1), 0);   // wave-uniform
            const int row_off = base_off + r * ROW_BYTES;
            if constexpr (WL == 4) ck.lo[k] = __builtin_amdgcn_raw_buffer_load_b128(spill, lo_at, row_off, 0);
            else if constexpr (WL == 3) ck.lo[k] = __builtin_amdgcn_raw_buffer_load_b96(spill, lo_at, row_off, 0);
            else if constexpr (WL == 2) ck.lo[k] = __builtin_amdgcn_raw_buffer_load_b64(spill, lo_at, row_off, 0);
            else if constexpr (WL == 1) ck.lo[k] = __builtin_amdgcn_raw_buffer_load_b32(spill, lo_at, row_off, 0);
            if constexpr (WH == 2) ck.hi[k] = __builtin_amdgcn_raw_buffer_load_b64(spill, hi_at, row_off, 0);
            else if constexpr (WH == 1) ck.hi[k] = __builtin_amdgcn_raw_buffer_load_b32(spill, hi_at, row_off, 0);
            if constexpr (ODD == 1) {
                if constexpr (PB == 2) ck.piece[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(spill, piece_at, row_off, 0);
                else ck.piece[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(spill, piece_at, row_off, 0);
            }
        }
    };
    // dword m of row k of a chunk in the register form described above
#define SA_TB_WORD(m) ((ODD == 1 && (m) == 0) ? ck.piece[k] << (32 - 8 * PB)                                                   \
                       : ((m) - ODD < WH) ? sa_tb_elem<WH>(ck.hi[k], ((m) - ODD < WH && (m) >= ODD) ? (m) - ODD : 0) \
                                          : sa_tb_elem<WL>(ck.lo[k], ((m) - ODD >= WH && (m) - ODD - WH < WL) ? (m) - ODD - WH : 0))

    auto process = [&](int ch, const Chunk& ck) {
        const uint32_t b4 = pick_word(bw, ch);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const int r = 4 * ch + k;
            if (r < lenq) {
                const uint32_t sh = 1u - s;                 // (32 or 64) - 2 * columns left, modulo the width of the shift
                uint32_t L;
                // lo = 2 - s as a signed number: 1 at the left border (s = 1), where the shift amount has wrapped
                // round and the window is meaningless, not above 0 elsewhere
                const int lo = (int)sh + 1;
                if constexpr (W == 1) {
                    L = min(sa_ffbl(SA_TB_WORD(0) >> (sh & 31)), s);
                } else if constexpr (W == 2) {
                    const uint64_t v = ((uint64_t)SA_TB_WORD(1) << 32) | SA_TB_WORD(0);
                    L = sa_ctz64_min(v >> (sh & 63), s);
                } else {
                    // columns 0..31 in one 64-bit half, 32..63 in the other; a run that starts in the upper
                    // half and finds nothing there goes on from column 31
                    uint64_t v0, v1;
                    if constexpr (W == 4) {
                        v0 = ((uint64_t)SA_TB_WORD(3) << 32) | SA_TB_WORD(2);
                        v1 = ((uint64_t)SA_TB_WORD(1) << 32) | SA_TB_WORD(0);
                    } else {
                        v0 = ((uint64_t)SA_TB_WORD(2) << 32) | SA_TB_WORD(1);
                        v1 = (uint64_t)SA_TB_WORD(0) << 32;
                    }
                    const bool upper = s > 65u;
                    const uint32_t lx = sa_ctz64_sat((upper ? v1 : v0) >> (sh & 63));
                    const uint32_t l0 = sa_ctz64_sat(v0);
                    L = min((upper && lx >= 64u) ? s - 65u + l0 : lx, s);
                }
                if constexpr (CLAMP) L = (uint32_t)max((int)L, lo);
                // t = L & 1 (1 = UP ends the run, 0 = DIAG); rest = s - L (DIAG: 2 * (column consumed + 1) + 1);
                // a = pcol[rest - 3] | t << 8; matches += (a == residue of row r); s = rest + 3t - 2, that is
                // s - run - (2 for a DIAG move); byte k of the chunk's word = L.  One block of instructions: the
                // LDS address is the state itself plus a constant that fits the instruction's offset field, and
                // the state is updated while the read is under way.  (Every output is written before the last
                // input is read: none may share a register with an input.)
                const uint32_t rest = s - L;
                uint32_t an_, tn_;
#define SA_TB_ROW(K)                                                                                         \
    asm volatile("ds_read_u8 %[an], %[rest] offset:%[off]\n\t"                                               \
                 "v_and_b32 %[tn], 1, %[L]\n\t"                                                              \
                 "v_lshl_or_b32 %[packed], %[L], " #K " * 8, %[packed]\n\t"                                  \
                 "v_mad_u32_u24 %[s], %[tn], 3, %[rest]\n\t"                                                 \
                 "v_add_u32 %[s], -2, %[s]\n\t"                                                              \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                  \
                 "v_cmp_eq_u32_sdwa vcc, %[an], %[bo] src0_sel:DWORD src1_sel:BYTE_" #K "\n\t"               \
                 "v_addc_co_u32 %[m], vcc, 0, %[m], vcc"                                                     \
                 : [an] "=&v"(an_), [tn] "=&v"(tn_), [packed] "+&v"(packed), [s] "=&v"(s), [m] "+&v"(matches) \
                 : [bo] "v"(b4), [rest] "v"(rest), [L] "v"(L), [off] "n"(PCOL_LDS_OFFSET - 3)                \
                 : "vcc")
                if (k == 3) SA_TB_ROW(3); else if (k == 2) SA_TB_ROW(2); else if (k == 1) SA_TB_ROW(1); else SA_TB_ROW(0);
#undef SA_TB_ROW
            }
        }
        ups += __builtin_popcount(packed & 0x01010101u);
        stage_rec[2 + ch] = packed;
    };

    // NS sets of registers take turns: the chunk worked on was requested NS - 1 chunks earlier.  (Without the row
    // requests -- every chunk reading the same four rows -- a launch of 50 000 peptides is 10 % shorter, without the
    // work on the rows 5 %; yet four or six sets instead of three changed nothing: what the requests cost is their
    // share of the memory system's throughput, not the time one of them takes.)
    Chunk c[NS];
    const int top = (maxlen - 1) >> 2;
#pragma unroll
    for (int j = 0; j < NS - 1; j++) load_chunk(top - j, c[j]);
    for (int t = top; t >= 0; t -= NS) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int ch = t - j;                 // the chunk in set j
            load_chunk(ch - (NS - 1), c[(j + NS - 1) % NS]);
            if (ch >= 0) process(ch, c[j]);
        }
    }
    // body bytes past the last chunk are zero, so that equal inputs give equal records
    for (int z = ((maxlen - 1) >> 2) + 1; z < STAGE_DWORDS - 2; z++) stage_rec[2 + z] = 0;
    matches_out = (int)matches;
    gaps_out = (int)ups;
#undef SA_TB_WORD
}

#endif
