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
#define SA_BUFFER_FLAGS 0x00020000

// ---------------------------------------------------------------- traceback
// Bottom row first, four rows per chunk.  The direction words of a chunk are
// requested two chunks ahead of their use (global spill area, coalesced).  A row
// is W = ceil(ncols/16) dwords per lane.  Two layouts:
//   REV = 0: the tag of column c is bits 2(c%16), 2(c%16)+1 of dword c/16, tags 0 LEFT,
//            1 UP, 3 DIAG; a run of LEFT moves is found with count-leading-zeros;
//   REV = 1: column c sits at bits 2(15 - c%16) of dword c/16 (lower columns higher up),
//            tags 0 LEFT, 2 UP, 3 DIAG; the same walk with count-trailing-zeros.
// In both, the zero count is 2 * (run of LEFT) + (1 if an UP ends it, 0 if a DIAG does).
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

template <int MODE, int W, int REV = 0>
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
                if (REV) {
                    if (W == 1) {
                        const uint32_t x = ck.d[k].x >> (sh & 31);
                        L = x ? __builtin_ctz(x) : 32;
                    } else if (W == 2) {
                        const uint64_t v = ((uint64_t)ck.d[k].x << 32) | ck.d[k].y;
                        const uint64_t x = v >> (sh & 63);
                        L = x ? __builtin_ctzll(x) : 64;
                    } else {
                        const uint64_t v0 = ((uint64_t)ck.d[k].x << 32) | ck.d[k].y;
                        const uint64_t v1 = ((uint64_t)ck.d[k].z << 32) | ck.d[k].w;
                        const bool upper = j2 > 64;
                        const uint64_t x = (upper ? v1 : v0) >> (sh & 63);
                        const int l0 = v0 ? __builtin_ctzll(v0) : 64;
                        L = x ? __builtin_ctzll(x) : (upper ? j2 - 64 + l0 : 64);
                    }
                } else if (W == 1) {
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

#endif
