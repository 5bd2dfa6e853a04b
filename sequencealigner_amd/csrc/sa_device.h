/*
 * sa_device.h -- data layout shared by the host side of the library and the
 * HIP kernels (plain C structs, no HIP types).
 *
 * Sequences live on the device twice:
 *   file order, one 64-byte slot per sequence (slot[0 .. len) = residue codes,
 *     the rest = SA_PAD_CODE).  A slot is the column operand of the DP:
 *     register column c holds slot[c].
 *   bucket order, position-major: the sequences are sorted by (length bucket,
 *     file index) and stored as codes_t[pos][sorted index], so the 64 lanes of a
 *     wave, which hold consecutive sorted indices, read row residue i of "their"
 *     sequences with one coalesced 64-byte load.  Inside a bucket the order is
 *     by file index, so for a given first sequence p the partners with a larger
 *     file index form a suffix of each bucket; cursor[p][b] is where that suffix
 *     starts.  The partners of p, bucket after bucket, are its "valid list";
 *     a wave takes 64 consecutive entries of it, whatever buckets they come from.
 */
#ifndef SA_DEVICE_H
#define SA_DEVICE_H

#include <stdint.h>

#define SA_LP 64                 /* DP columns held in registers by the peptide kernel */
#define SA_PAD_CODE 20           /* residue code of padding; its substitution scores are 0 */
#define SA_INVALID_FI 0xFFFFFFFFu
/* the 256-byte buffer of the work-queue head: dword 0 = head, dword SA_STATUS_WORD = status bits a kernel may raise */
#define SA_STATUS_WORD 8
#define SA_STATUS_LDS_BASE 1u    /* the kernel's dynamic LDS does not start at offset 0: it refused to run */
#define SA_SPRIME_STRIDE 32      /* dwords per row of the shifted substitution table */

/* LDS carve-up of the peptide kernel (bytes) */
/* Query-profile rows are 68 dwords apart: a row is read 16 bytes per lane with the
 * lanes of a wave in different rows (one per row residue), and a stride of 64 dwords
 * would put every row on the same four LDS banks. */
#define SA_PROF_STRIDE 68
#define SA_LDS_PROF_BYTES (21 * SA_PROF_STRIDE * 4)   /* 21 rows (20 residues + padding) */
#define SA_LDS_EMASK_BYTES 192   /* 21 x 8 B equality masks, padded */
#define SA_LDS_STAGE_DWORDS 20   /* per-lane staging record: 80 B */
#define SA_LDS_STAGE_BYTES (64 * SA_LDS_STAGE_DWORDS * 4)
#define SA_LDS_FIXED_BYTES (SA_LDS_PROF_BYTES + SA_LDS_EMASK_BYTES + SA_LDS_STAGE_BYTES)
/* paired kernels: profile rows 62 (52 columns, ng = 13) or 66 dwords apart, i.e. 2 banks modulo the 64 banks that
 * an 8-byte LDS read sees: a lane reads its four table entries as two 8-byte reads, the 32 lanes of a read group hold
 * up to 21 different row residues, and 21 windows of 2 banks fit side by side.  (With 16-byte reads and a stride of
 * 4 banks only 16 rows fit: rows b and b + 16 shared their banks and 41 % of the LDS cycles were conflict cycles.)
 * Both peptide variants stage 72 bytes per lane (8 of header, 64 of residues or record rows; the
 * last 8 bytes of an 80-byte record are padding and are not staged).  16 workgroups of the 52-column variant
 * share a CU's 160 KiB, 15 of the 64-column one. */
#define SA_PAIR_PROF_STRIDE(ng) ((ng) <= 13 ? 62 : 66)
#define SA_PAIR_STAGE_DWORDS(ng) 18
#define SA_PAIR_LDS_BYTES(ng) (21 * SA_PAIR_PROF_STRIDE(ng) * 4 + 2 * SA_LDS_EMASK_BYTES + 64 * SA_PAIR_STAGE_DWORDS(ng) * 4)

/* One unit of work pulled from the device-side queue (32 bytes).
 * TRIANGULAR: first sequence p (file index) against entries
 *             [v0, v0 + 64*count) of its valid list; b0 is the bucket that holds
 *             entry v0 and c0 the number of valid entries in the buckets before b0.
 * ADJACENT:   pairs (p+l, p+l+1) for l in [0, count), count <= 64. */
typedef struct {
    uint32_t p;
    uint32_t v0;
    uint32_t count;
    uint32_t b0;
    uint32_t c0;
    uint32_t reserved[3];
} SaItem;

/* Result record, `stride` bytes per pair (multiple of 16):
 *   int32  score
 *   uint16 matches
 *   uint16 gaps          (columns where sequence 1 has '-')
 *   uint8  body[stride-8]
 * body[r], r = 0..len2-1, describes how the traceback crosses DP row i = r+1
 * (the traceback itself runs from row len2 down to row 1):
 *   bits 7..1 = number of LEFT moves taken in that row before leaving it,
 *   bit 0     = how the row was left: 0 = DIAG, 1 = UP.
 * Whatever is left of sequence 1 after row 1 is consumed by LEFT moves. */
#define SA_REC_HEADER_BYTES 8
/* Long sequences (more than 64 residues): 16-byte header {int32 score, uint32 matches,
 * uint32 gaps, uint32 0}, then one uint16 per row of sequence 2 with the same meaning
 * (bits 15..1 = LEFT moves, bit 0 = DIAG/UP). */
#define SA_LONG_HEADER_BYTES 16
#define SA_MAX_LONG_LEN 4096

typedef struct {
    const uint8_t* codes_fo;     /* [N][64] residue codes padded with SA_PAD_CODE, file order */
    const uint16_t* len_fo;      /* [N] */
    const uint8_t* codes_t;      /* [64 pos][npad] residue codes in bucket order, position-major */
    const uint8_t* codes_s;      /* [4][npad][16] the same codes in 16-byte pieces, piece-major: the lanes of a wave (consecutive
                                    bucket-order slots) read a piece as 1 KiB in a row, and only the pieces their sequences reach */
    const uint32_t* fi_sorted;   /* [npad] file index of each bucket-order slot */
    const uint16_t* len_sorted;  /* [npad] */
    const uint32_t* bucket_base; /* [nb+1] first bucket-order slot of each bucket */
    const uint32_t* cursor;      /* [N][nb] entries of bucket b with file index <= p */
    uint32_t nb;
    uint32_t npad;               /* row pitch of codes_t */
    const int32_t* sprime;       /* [21][32] S[a][b] - 2*gap; row 20 and columns >= 20 are 0 */
    const SaItem* items;
    uint32_t nitems;
    uint32_t* counter;           /* work-queue head, zeroed before the launch */
    uint8_t* out;                /* records, canonical pair order */
    uint32_t stride;             /* bytes per record */
    uint64_t out_base_pair;      /* canonical pair index of out[0] */
    uint32_t* spill;             /* direction rows that do not fit in LDS: per workgroup
                                    spill_rows x 64 lanes x 4 dwords */
    uint32_t spill_rows;
    uint32_t nseq;
    int32_t gap;
    unsigned long long* phase_cycles;  /* optional [grid][8] shader-cycle totals per phase, or NULL */
    /* long-sequence kernel only */
    uint32_t lslot;              /* bytes per file-order slot = rows of codes_t = row capacity (multiple of 64) */
    int32_t* bcol;               /* per workgroup 2 x lslot x 64: last DP column of a strip, for the next strip */
    uint32_t rebase_mask;        /* packed long kernels, tagged rows: the lanes' bases move up every rebase_mask + 1 rows
                                    (15, 7 or 3: the larger the table's entries, the more often) */
} SaPepParams;


/* ---- rows of CSV text written on the device (sa_format.hip) */
enum { SA_COL_NONE = 0, SA_COL_SEQ1, SA_COL_SEQ2, SA_COL_SCORE, SA_COL_ALIGN, SA_COL_MATCHES, SA_COL_MISMATCHES,
       SA_COL_GAPS, SA_COL_SIMILARITY, SA_COL_OTHER1, SA_COL_OTHER2 };
#define SA_FMT_MAX_COLS 32
#define SA_FMT_PART_BYTES 128

typedef struct {
    const uint8_t* codes_fo;      /* file-order slots of residue codes */
    const uint16_t* len_fo;
    uint32_t lslot;
    const uint8_t* rec;           /* records of the submission, device memory */
    uint32_t stride, row_bytes;
    uint64_t first_pair, npairs, nseq;
    int32_t schedule;
    const char* other;            /* pass-through text of every input row: other[other_off[s] .. other_off[s+1]) */
    const uint32_t* other_off;
    uint32_t ncols;
    int8_t kind[SA_FMT_MAX_COLS]; /* SA_COL_* per output column */
    char part[3][SA_FMT_PART_BYTES];   /* the alignment format split at its two "%s" */
    uint32_t part_len[3];
    char alphabet[24];
    uint32_t* lens;               /* [npairs] row lengths */
    uint64_t* offs;               /* [npairs] row offsets in text */
    char* text;
    uint64_t* total;              /* bytes of text of the submission */
} SaFormatParams;

#endif
