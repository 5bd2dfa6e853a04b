// sa_long_walk.h -- the traceback walk of the packed long-sequence kernels (sa_kernels_long_ckpt.hip,
// sa_kernels_long_pair.hip) over direction rows of one 64-column strip: both halves of a lane side by side, eight rows
// per request.  Reference: the walk of include/seqalign.h:63-88, one row per step instead of one cell.
//
// Direction rows: row r of a strip is 1 KiB (16 bytes per lane: the tags of the strip's 64 columns, two bits each, as
// fill_row_pair leaves them and store_dirs splits them per half), rows 1 KiB apart, one area per half.
// The macros below work on these names of the code that uses them:
//   wi[2], wj[2]        rows and columns still to go per half (the path stands in cell (wi, wj))
//   wrun[2]             LEFT moves already taken in the current row (a run may go through several strips)
//   wmatch[2], wgaps[2] the record's counters;  wacc[2][4] the record rows being collected
//   tc0, tc1            the row (relative to r0) a half's path stands in within the rows on offer, -1: none
//   r0, cbase           first row of the rows on offer; first column of the strip
//   tile_a, tile_b      the two halves' direction rows (row r0 first), wave-uniform;  voff = lane * 16
//   resp                the lane's second sequence from row r0 on;  pseq, P.lslot: the first sequences in LDS
//   a0x .. a7x, b0x .. b7x (sa_u32x4v), rx (sa_u32x2v): the registers of a chunk;  rec0, rec1: the records
#ifndef SA_LONG_WALK_H
#define SA_LONG_WALK_H

typedef uint32_t sa_u32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t sa_u32x2v __attribute__((ext_vector_type(2)));

// position of the lowest set bit of a 64-bit value, 64 for 0
__device__ __forceinline__ int lck_ctz64(uint64_t v) { return v ? __builtin_ctzll(v) : 64; }

// A path finishes its rows from the last to the first, so a record is written eight rows (16 bytes) at a time: the
// rows' codes are shifted into a 128-bit register set, newest lowest, and the set is stored when the row that starts a
// group of eight is done.  (One 2-byte store per row and lane -- 64 partial lines per wave and row -- cost a third of
// the launch.)  Rows of the first group beyond the sequence's end stay zero.
#define SA_LCK_PUSH(H, REC, ROW, CODE)                                                                                     \
    {                                                                                                                      \
        wacc[H][3] = __builtin_amdgcn_alignbit(wacc[H][3], wacc[H][2], 16);                                                \
        wacc[H][2] = __builtin_amdgcn_alignbit(wacc[H][2], wacc[H][1], 16);                                                \
        wacc[H][1] = __builtin_amdgcn_alignbit(wacc[H][1], wacc[H][0], 16);                                                \
        wacc[H][0] = (wacc[H][0] << 16) | (uint32_t)(CODE);                                                                \
        if (((ROW) & 7) == 0) {                                                                                            \
            *reinterpret_cast<uint4*>((REC) + SA_LONG_HEADER_BYTES + 2 * (size_t)(ROW)) =                                  \
                make_uint4(wacc[H][0], wacc[H][1], wacc[H][2], wacc[H][3]);                                                \
            wacc[H][0] = wacc[H][1] = wacc[H][2] = wacc[H][3] = 0;                                                         \
        }                                                                                                                  \
    }

// A chunk's sixteen tag rows and its eight residues are requested and awaited by one block of instructions: left to the
// compiler a row's load moves into the branch that uses it and is waited for on the spot -- a trip to memory inside a
// row -- and requests kept in flight across compiler-generated code are not safe (the compiler takes their registers
// for written at the request, and copies or spills them before the data has arrived).
#define SA_LCK_FETCH(CH)                                                                                                 \
    {                                                                                                                    \
        /* a lane asks for the rows of a half only if its path of that half stands in or below this chunk: the other   \
           lanes' rows were not stored either */                                                                         \
        const unsigned long long needA_ = __ballot(tc0 >= 8 * (CH)), needB_ = __ballot(tc1 >= 8 * (CH));                 \
        unsigned long long all_;                                                                                         \
        asm volatile("s_mov_b64 %[all], exec\n\t"                                                                        \
                 "s_and_b64 exec, %[all], %[na]\n\t"                                                                     \
                 "global_load_dwordx4 %0, %[vo], %[a0]\n\tglobal_load_dwordx4 %1, %[vo], %[a0] offset:1024\n\t"          \
                 "global_load_dwordx4 %2, %[vo], %[a0] offset:2048\n\tglobal_load_dwordx4 %3, %[vo], %[a0] offset:3072\n\t" \
                 "global_load_dwordx4 %4, %[vo], %[a1]\n\tglobal_load_dwordx4 %5, %[vo], %[a1] offset:1024\n\t"          \
                 "global_load_dwordx4 %6, %[vo], %[a1] offset:2048\n\tglobal_load_dwordx4 %7, %[vo], %[a1] offset:3072\n\t" \
                 "s_and_b64 exec, %[all], %[nb]\n\t"                                                                     \
                 "global_load_dwordx4 %8, %[vo], %[b0]\n\tglobal_load_dwordx4 %9, %[vo], %[b0] offset:1024\n\t"          \
                 "global_load_dwordx4 %10, %[vo], %[b0] offset:2048\n\tglobal_load_dwordx4 %11, %[vo], %[b0] offset:3072\n\t" \
                 "global_load_dwordx4 %12, %[vo], %[b1]\n\tglobal_load_dwordx4 %13, %[vo], %[b1] offset:1024\n\t"        \
                 "global_load_dwordx4 %14, %[vo], %[b1] offset:2048\n\tglobal_load_dwordx4 %15, %[vo], %[b1] offset:3072\n\t" \
                 "s_mov_b64 exec, %[all]\n\t"                                                                            \
                 "global_load_dwordx2 %16, %[rp], off\n\t"                                                               \
                 "s_waitcnt vmcnt(0)"                                                                                    \
                 : "=&v"(a0x), "=&v"(a1x), "=&v"(a2x), "=&v"(a3x), "=&v"(a4x), "=&v"(a5x), "=&v"(a6x), "=&v"(a7x),        \
                   "=&v"(b0x), "=&v"(b1x), "=&v"(b2x), "=&v"(b3x), "=&v"(b4x), "=&v"(b5x), "=&v"(b6x), "=&v"(b7x), "=&v"(rx), \
                   [all] "=&s"(all_)                                                                                     \
                 : [vo] "v"(voff), [a0] "s"(tile_a + (size_t)(CH) * 8192), [a1] "s"(tile_a + (size_t)(CH) * 8192 + 4096), \
                   [b0] "s"(tile_b + (size_t)(CH) * 8192), [b1] "s"(tile_b + (size_t)(CH) * 8192 + 4096),                 \
                   [rp] "v"(resp + 8 * (CH)), [na] "s"(needA_), [nb] "s"(needB_)                                          \
                 : "memory", "scc");                                                                                     \
    }
#define SA_LCK_STEP(H, TC, D, REC)                                                                                         \
    if ((TC) == t) {                                                                                                       \
        const int jl = wj[H] - cbase;                         /* columns of this strip still to go: 1 .. 64 */             \
        const uint64_t v0 = ((uint64_t)(D).x << 32) | (D).y;                                                               \
        const uint64_t v1 = ((uint64_t)(D).z << 32) | (D).w;                                                               \
        const bool upper = jl > 32;                                                                                        \
        const int jj = upper ? jl - 32 : jl;                                                                               \
        const uint64_t x = (upper ? v1 : v0) >> ((64 - 2 * jj) & 63);                                                      \
        /* trailing zeros from column jl - 1 down = 2 * (LEFT moves) + (1 if an UP ends the run) */                        \
        const int L = x ? __builtin_ctzll(x) : (upper ? 2 * jj + lck_ctz64(v0) : 64);                                      \
        const int run = min(L >> 1, jl);                                                                                   \
        wj[H] -= run;                                                                                                      \
        wrun[H] += run;                                                                                                    \
        int up = 1;                                           /* at the matrix's left border: UP */                        \
        bool fin = wj[H] == 0;                                                                                             \
        if (run < jl) { up = L & 1; fin = true; }                                                                          \
        (TC) = -1;                                            /* the run goes on in the strip to the left */               \
        if (fin) {                                                                                                         \
            SA_LCK_PUSH(H, REC, r0 + t, wrun[H] * 2 + up)                                                                  \
            wrun[H] = 0;                                                                                                   \
            if (up == 0) {                                                                                                 \
                wmatch[H] += (pseq[(H) * P.lslot + wj[H] - 1] == b) ? 1 : 0;                                               \
                wj[H]--;                                                                                                   \
            }                                                                                                              \
            wgaps[H] += up;                                                                                                \
            wi[H]--;                                                                                                       \
            if (wj[H] > cbase) (TC) = t - 1;                  /* still in the strip: the row above (-1: the next tile) */  \
        }                                                                                                                  \
    }
#define SA_LCK_ROWSTEP(CH, K)                                                                                              \
    {                                                                                                                      \
        const int t = 8 * (CH) + (K);                                                                                      \
        const uint32_t b = (((K) < 4 ? rx.x : rx.y) >> (8 * ((K) & 3))) & 0xff;                                            \
        SA_LCK_STEP(0, tc0, a##K##x, rec0)                                                                                 \
        SA_LCK_STEP(1, tc1, b##K##x, rec1)                                                                                 \
    }

#endif
