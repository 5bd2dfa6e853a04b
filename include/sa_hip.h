/*
 * sa_hip.h -- C ABI of libsa_hip: Needleman-Wunsch score + traceback for many
 * sequence pairs on one AMD MI355X (gfx950).
 *
 * The library stands where the reference's worker pool stands: the reference
 * fills an array of AlignTask {seq1, seq2, len1, len2, scoring, result}
 * (include/thread.h:35-42, src/main.c:43-52), wakes its pthread workers
 * (src/main.c:54-60, include/thread.h:68-85) and, after they finish
 * (src/main.c:62), formats the Alignment results in pair order
 * (src/main.c:65-73).  Here the caller uploads the sequences once, submits a
 * range of rows of a pair schedule, and receives compact result records in
 * the same pair order; sa_expand_record() turns one record back into what the
 * reference's Alignment struct holds (include/common.h:19-29).
 *
 * Conventions: every function returns 0 on success and a negative code on
 * failure (sa_strerror() gives the text); the library owns all device and
 * pinned memory; the caller owns the sequences and whatever it does with the
 * records.  One context drives one GPU; a context may be used by one
 * submitting thread (sa_submit*) and one consuming thread (sa_next*, sa_release) at the same time.
 */
#ifndef SA_HIP_H
#define SA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sa_ctx sa_ctx;

/* Pair schedules.
 * ADJACENT:   pair k = (row k, row k+1), the reference's only schedule
 *             (src/main.c:43-52, include/user.h:31).  "Row r" of the schedule is pair r.
 * TRIANGULAR: all pairs (i, j), i < j, row-major, sequence 1 = row i, sequence 2 = row j.
 *             "Row r" of the schedule is the set of pairs with i = r. */
#define SA_SCHEDULE_ADJACENT 0
#define SA_SCHEDULE_TRIANGULAR 1

#define SA_ALPHABET "ARNDCQEGHILKMFPSTWYV"   /* include/scoring.h:6 */
#define SA_MAX_PEPTIDE_LEN 64      /* up to here the pair-per-lane register kernel runs */
#define SA_MAX_SEQ_LEN 4096         /* longest sequence the library takes (strip kernel above 64) */
/* sa_init device: a context without a GPU.  It answers schedule arithmetic and
 * builds work lists (for tests of the host logic); every compute call on it fails
 * with SA_ERR_STATE.  It is not a CPU implementation of anything. */
#define SA_DEVICE_NONE (-1)

enum {
    SA_OK = 0,
    SA_ERR_ARG = -1,          /* bad argument */
    SA_ERR_ALPHABET = -2,     /* residue outside SA_ALPHABET (undefined behaviour in the reference, include/scoring.h:42) */
    SA_ERR_LENGTH = -3,       /* empty sequence or longer than the context's max_seq_len */
    SA_ERR_NOMEM = -4,
    SA_ERR_STATE = -5,        /* call out of order (nothing uploaded, nothing submitted, ...) */
    SA_ERR_BUSY = -6,         /* both in-flight slots are in use: call sa_next/sa_release first */
    SA_ERR_RECORD = -7,       /* record does not describe an alignment of the given sequences */
    SA_ERR_KERNEL = -9,       /* a kernel found its launch configuration unusable and did nothing (reported by sa_next* / sa_run_resident) */
    SA_ERR_NOALLOC_SIZE = -8, /* sa_align_adjacent_noalloc: more pairs than its fixed scratch buffer serves */
    SA_ERR_HIP = -1000        /* -1000 - hipError_t */
};

/* What the reference's Alignment carries besides the two strings
 * (include/common.h:19-29); similarity is kept as the integer the reference
 * prints: (int)(((double)matches / columns) * 10000) (include/seqalign.h:109,
 * include/csv.h:125). */
typedef struct {
    int32_t score;
    int32_t columns;
    int32_t matches;
    int32_t mismatches;
    int32_t gaps;
    int32_t similarity_bp;
} sa_alignment_stats;

/* ---- life cycle: stands for init_thread_pool / destroy_thread_pool
 * (include/thread.h:87-114) and init_scoring_matrix (include/scoring.h:15-52).
 * S is the 20x20 substitution matrix indexed [sequence-1 residue][sequence-2
 * residue] in SA_ALPHABET order (include/seqalign.h:55); NULL selects the
 * reference's BLOSUM50.  gap is the linear gap penalty (include/user.h:13).
 * SA_ERR_ARG when max_seq_len * max|S - 2*gap| >= 2^29: the cells of the DP are 32-bit (the reference's are too,
 * include/seqalign.h:18) and keep two tag bits. */
int sa_init(sa_ctx** ctx, const int32_t* S, int32_t gap, uint32_t max_seq_len, int device);
void sa_destroy(sa_ctx* ctx);
const char* sa_strerror(int code);

/* ---- sequences: stands for the seqs[] / seq_lens[] batch arrays the reference
 * parses into (src/main.c:25-37).  codes are residue indices 0..19 (what
 * AMINO_LOOKUP yields, include/seqalign.h:41-44), sequence s occupying
 * codes[offs[s] .. offs[s]+lens[s]).  Rows keep their file order. */
int sa_encode(const char* seq, uint32_t len, uint8_t* codes);
int sa_upload(sa_ctx* ctx, const uint8_t* codes, const uint64_t* offs, const uint32_t* lens, uint64_t nseq);

/* One set of sequences on several GPUs: sa_clone_sequences gives `ctx` everything `donor` holds after its
 * sa_upload -- the sequences, the tables of sa_prepare_schedule when they are built, the pass-through text and
 * the text layout -- by device-to-device copies, with no second validation or bucketing on the host.  `ctx`
 * must allow sequences as long as the donor's and have nothing in flight.
 * sa_prepare_schedule builds what the first submission of a schedule would build (TRIANGULAR: the length-bucket
 * order and the table of where each row's partners start), so that it is built once, before cloning. */
int sa_prepare_schedule(sa_ctx* ctx, int schedule);
int sa_clone_sequences(sa_ctx* ctx, sa_ctx* donor);

/* ---- schedule arithmetic (no device work) */
uint64_t sa_schedule_rows(const sa_ctx* ctx, int schedule);                 /* number of rows */
uint64_t sa_pair_index(const sa_ctx* ctx, int schedule, uint64_t row);      /* canonical index of the row's first pair */
uint64_t sa_pair_count(const sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end);
uint64_t sa_cell_count(const sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end); /* sum len1*len2 */
uint32_t sa_record_stride(const sa_ctx* ctx);                               /* bytes per result record */
uint32_t sa_record_row_bytes(const sa_ctx* ctx);                            /* 1, or 2 when a sequence is longer than 64 */
/* split [0, rows) into `parts` contiguous row ranges of near-equal cell count;
 * bounds receives parts+1 row indices */
int sa_shard_rows(const sa_ctx* ctx, int schedule, uint32_t parts, uint64_t* bounds);

/* ---- dispatch: stands for filling g_thread_work[] + sem_post(work_ready)
 * (src/main.c:54-60) and sem_wait(work_done) (src/main.c:62).
 * sa_submit starts the kernels for rows [row_begin,row_end) and the copy of
 * their records to pinned host memory, and returns at once.  sa_next blocks
 * until the oldest submission is complete and hands out its records, which
 * stay valid until sa_release.  Two submissions may be in flight. */
int sa_submit(sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end);
/* Optional, part of the set-up (the reference allocates its batch arrays before its timing window opens,
 * src/main.c:25-30): device and pinned host buffers of both in-flight slots for submissions of up to `npairs` pairs of
 * sequences of up to `longest_sequence` residues.  Without it the buffers are allocated, and grown, by the submissions
 * themselves (a pinned allocation of 100 MB takes tens of milliseconds). */
int sa_reserve_submission(sa_ctx* ctx, uint64_t npairs, uint32_t longest_sequence);
int sa_next(sa_ctx* ctx, const uint8_t** records, uint64_t* first_pair, uint64_t* npairs);
int sa_release(sa_ctx* ctx);

/* ---- records without their padding.  A record has the context's stride, sized for the longest uploaded sequence; most
 * pairs need less (one code per row of THEIR second sequence).  With sa_set_compact_records(ctx, 1) a submission's records
 * are packed on the device before they travel to the host: pair after pair in canonical order, each cut to
 * sa_compact_record_size(second sequence) = header + codes, rounded up to 4 bytes; sa_next then hands out that packed
 * stream (sa_compact_bytes of the submission's rows long).  Pair (i, j) of a submission that starts at row r starts at
 * sa_compact_pair_offset(schedule, i, j) - sa_compact_pair_offset(schedule, r, r + 1); sa_expand_record reads a packed
 * record when it is given the record's own size as its stride.  The reference moves whole 280-byte Alignment structs
 * from workers to writer (include/common.h:19-29, include/thread.h:35-42); over PCIe the bytes are the cost.
 * Default: off (records of sa_record_stride bytes each). */
int sa_set_compact_records(sa_ctx* ctx, int enable);
uint32_t sa_compact_record_size(const sa_ctx* ctx, uint64_t second_sequence);
uint64_t sa_compact_pair_offset(const sa_ctx* ctx, int schedule, uint64_t i, uint64_t j);
uint64_t sa_compact_bytes(const sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end);

/* ---- rows of CSV text written on the GPU.  Stands for the formatting the reference does on its main
 * thread after a batch (buffer_output, include/csv.h:84-141; init_format, include/csv.h:38-82): a
 * submission hands back the finished text of its pairs, in pair order, instead of records.
 * sa_text_layout: what each output column shows (SA_TEXT_*; OTHER1/OTHER2 = the pass-through text of
 * the pair's first / second input row) and the alignment format with its two "%s".
 * sa_upload_passthrough: the pass-through text of every input row, row s at bytes[offs[s] .. offs[s+1]).
 * sa_submit_text / sa_next_text mirror sa_submit / sa_next (same two slots, same sa_release); the text
 * is copied to one of two pinned buffers of the context in turn and stays valid until the next-but-one
 * sa_next_text on that context, also after sa_release. */
enum { SA_TEXT_NONE = 0, SA_TEXT_SEQ1, SA_TEXT_SEQ2, SA_TEXT_SCORE, SA_TEXT_ALIGN, SA_TEXT_MATCHES, SA_TEXT_MISMATCHES,
       SA_TEXT_GAPS, SA_TEXT_SIMILARITY, SA_TEXT_OTHER1, SA_TEXT_OTHER2 };
typedef struct {
    uint32_t ncols;
    int8_t kind[32];
    char align_fmt[128];
} sa_text_layout;
int sa_set_text_layout(sa_ctx* ctx, const sa_text_layout* layout);
int sa_upload_passthrough(sa_ctx* ctx, const char* bytes, const uint64_t* offs, uint64_t nseq);
int sa_submit_text(sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end);
/* Optional, part of the set-up like sa_reserve_submission: the buffers of text submissions of up to `npairs` pairs and
 * `text_bytes` bytes of text (device side of both slots, the two pinned buffers finished text is handed out in). */
int sa_reserve_text(sa_ctx* ctx, uint64_t npairs, uint64_t text_bytes);
int sa_next_text(sa_ctx* ctx, const char** text, uint64_t* nbytes, uint64_t* first_pair, uint64_t* npairs);

/* ---- measurement: the same kernels with the records left in device memory.
 * sa_prepare_resident builds the work list and buffers for the row range once;
 * sa_run_resident launches the kernels over it (kernel_ms: HIP-event time on the
 * launch stream); sa_fetch_resident copies records [first, first+n) of the
 * range (relative to its first pair) to host memory. */
int sa_prepare_resident(sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end);
int sa_run_resident(sa_ctx* ctx, float* kernel_ms);
int sa_fetch_resident(sa_ctx* ctx, uint64_t first, uint64_t n, uint8_t* dst);
/* 128-bit digest of records [first, first+n) of the resident range, computed on the device (position-dependent hash
 * of every 16 bytes, summed): equal for two contexts exactly when their records are the same bytes in the same
 * places, up to hash collisions.  For comparing kernels over ranges too large to copy out. */
int sa_digest_resident(sa_ctx* ctx, uint64_t first, uint64_t n, uint64_t digest[2]);
/* tuning knobs of the peptide kernel: 64-pair tiles per work item, workgroups (waves)
 * per CU in a launch; 0 keeps the current value.  Until they are set the library takes 1 .. 16
 * tiles per item from the size of the row range (at least 32 items per workgroup) and launches
 * as many workgroups per CU as registers and LDS let reside there, plus one. */
int sa_set_tuning(sa_ctx* ctx, uint32_t tiles_per_item, uint32_t wg_per_cu);
/* width of the TRIANGULAR length buckets in residues (a wave's 64 second sequences, and the two first sequences of a
 * packed wave, come from one bucket: wide buckets pad short sequences up to the longest of the wave, narrow ones leave
 * first sequences without a partner); 0 = automatic: exact lengths when every length has 256 sequences or more, wider
 * otherwise (sequences longer than 64 residues: about 48 sequences per bucket, 8 to 64 residues wide).  Results do not depend on it. */
int sa_set_bucket_width(sa_ctx* ctx, uint32_t width);
/* TRIANGULAR work normally runs on kernels that handle two first sequences per wave in packed
 * 16-bit cells; they are used only when the substitution table allows it (S - 2*gap >= 1 everywhere
 * and longest sequence * max(S - 2*gap) <= 1983 when no sequence exceeds 64 residues; 1 .. 24 otherwise).  enable = 0 forces one first
 * sequence per wave, 32-bit cells (same results; for comparisons and tests).  Default: enabled. */
int sa_set_pairing(sa_ctx* ctx, int enable);
/* A set with sequences of more than 64 residues runs on the long-sequence kernels and its records have the long form.
 * Where such a set also holds peptides, the TRIANGULAR tiles whose sequences all have at most 64 residues are handed to
 * the paired peptide kernel (writing the same long-form records, byte for byte): the reference treats every pair alike
 * (include/seqalign.h:13-113), here a few long sequences among peptides must not cost the peptides their kernel.
 * enable = 0 keeps every tile on the long-sequence kernel (same results; for comparisons and tests).  Default: enabled;
 * without effect when pairing is off or the table does not fit the packed peptide cells. */
int sa_set_peptide_tiles(sa_ctx* ctx, int enable);
/* Loads the device code a job of this kind will use (slot packing and the peptide kernels; the paired peptide kernel
 * for TRIANGULAR work of a context whose sequences cannot exceed 64 residues; the row formatter; the record packer)
 * on the context's device now, instead of at the first launch of each kernel.  Touches no state of the context and may
 * run on a thread of its own while the caller reads its input -- the CSV driver does that: a fresh process spent 23 ms
 * of its timed window loading code at first launches.  No counterpart in the reference (its code is loaded with the
 * program, src/main.c); results do not depend on it. */
int sa_preload_kernels(sa_ctx* ctx, int schedule, int device_text, int compact_records);
/* Sequences longer than 64 residues on the packed TRIANGULAR kernel: how the traceback gets its directions.  The
 * reference keeps the whole score matrix of a pair and walks it (include/seqalign.h:18,63-88).  rows = 64 (default) or
 * 32: the fill keeps values only and leaves a checkpoint (one matrix row per strip) every `rows` rows plus the last
 * column of every strip; the traceback fills again, with direction tags, only the tiles its path goes through.
 * rows = 0: the fill writes two direction bits for every cell and the traceback reads them back (the earlier kernel;
 * four times the scratch memory).  SA_LONG_CHECKPOINT_AUTO (the default): 64 when the longest uploaded sequence has
 * 512 residues or more, else 0 -- for short sequences a 64-row tile is most of the matrix, and filling it twice costs
 * more than keeping the direction bits.  Same records either way. */
#define SA_LONG_CHECKPOINT_AUTO 0xFFFFFFFFu
int sa_set_long_checkpoint_rows(sa_ctx* ctx, uint32_t rows);
/* Sequences longer than 64 residues, few pairs: a submission of at most 384 pairs per CU runs on a kernel that gives
 * every pair a whole wave (the 64 lanes sweep anti-diagonals of one matrix and hand their operands to the neighbour
 * lane) instead of a lane: a few hundred protein pairs then use a few hundred waves, not a handful.  The reference
 * handles one pair per call of align_sequences (include/seqalign.h:13-113) whatever the count.  mode 0 (default)
 * chooses by the pair count of the submission, 1 = for every submission of long sequences of up to 4096 pairs per CU
 * (beyond that the per-pair work list would not be worth its memory), -1 = never.  Same records either way. */
int sa_set_cooperative(sa_ctx* ctx, int mode);
/* Waves per pair in that kernel.  0 (default): from the pair count -- with fewer pairs than the chip has wave slots
 * (16 per CU) a pair gets a workgroup of 2, 4, 8 or 16 waves, which take its 64-column strips in turn and hand a
 * strip's last column to the wave on their right through LDS, 64 rows at a time, while both are being swept; 64 pairs
 * of 4096 residues then keep 1024 waves busy, not 64.  1, 2, 4, 8, 16: that many, whatever the count (measurements).
 * The reference has no counterpart: it aligns a pair on one thread (include/seqalign.h:13-113).  Same records. */
int sa_set_cooperative_waves(sa_ctx* ctx, int waves);

/* name of the fill + traceback kernel the context's most recent launch ran ("" before the first): the library picks
 * between packed 16-bit and 32-bit cells, peptide and strip kernels from the table, the lengths and the row range */
const char* sa_last_kernel(const sa_ctx* ctx);
/* workgroups of that launch, and how many it would have had with memory to spare: the long-sequence kernels keep a
 * scratch area per workgroup (up to 31 MiB for 1000-residue proteins), sized at the first launch after an upload from
 * what is free on the device then (three quarters of it at most, never the last 16 GiB); a context that shares its
 * card with others gets fewer workgroups, i.e. fewer waves per SIMD, and this call says so.  No counterpart in the
 * reference. */
int sa_last_grid(const sa_ctx* ctx, uint32_t* workgroups, uint32_t* wanted);

/* diagnostic: with enable != 0 the next launches accumulate shader cycles per kernel
 * phase (0 queue fetch, 1 profile build, 2 operand set-up, 3 fill, 4 traceback,
 * 5 record store), summed over workgroups; totals (may be NULL) receives the sums of
 * the previous launch.  Timed launches run a few percent slower. */
int sa_debug_phase_cycles(sa_ctx* ctx, int enable, uint64_t totals[8]);
/* diagnostic: the work list the kernels would pull for rows [row_begin,row_end), eight
 * uint32 per item {p, v0, count, b0, c0, p2, 0, 0}.  TRIANGULAR: first sequence p
 * against entries [v0, v0 + 64*count) of its valid list, i.e. of the sequences with a
 * larger file index taken in bucket order (sa_debug_bucket_order); with pairing, p2 is
 * either 0xFFFFFFFF or a second first sequence (p2 > p) aligned against those of the
 * same entries whose file index is above p2.  ADJACENT: pairs p .. p+count-1 (p2 = 0). */
int sa_debug_work_items(sa_ctx* ctx, int schedule, uint64_t row_begin, uint64_t row_end,
                        uint32_t* items8, uint64_t cap, uint64_t* count);
/* diagnostic: file index of every sequence in bucket order (length bucket, then file index) */
int sa_debug_bucket_order(sa_ctx* ctx, uint32_t* file_index, uint64_t cap, uint64_t* count);

/* ---- one entry point that allocates nothing.  No counterpart in the reference: it is for callers that
 * cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc; not stream
 * capture: the call synchronises with the host).  It is the ADJACENT peptide path of sa_submit / sa_next with every buffer passed in:
 *   d_slots    device memory, nseq slots of 64 bytes: residue codes 0..19 of sequence s, then padding (any
 *              value; 20 is what sa_upload writes)
 *   d_lens     device memory, nseq lengths (uint16), each 1 .. 64
 *   d_records  device memory, (nseq - 1) * stride bytes; receives the record of pair k = (s_k, s_k+1) at
 *              k * stride, in the layout sa_expand_record reads (row_bytes 1)
 *   stride     multiple of 16, 16 .. 80, at least 8 + longest sequence
 *   S, gap     host memory / value, as for sa_init (NULL = BLOSUM50)
 * The kernels' scratch (work-queue head, work list, shifted table, direction rows) is a fixed array of
 * SA_NOALLOC_SCRATCH_BYTES in the library's device image, one copy per GPU, held until the process ends.  At
 * most SA_NOALLOC_MAX_PAIRS pairs per call (SA_ERR_NOALLOC_SIZE beyond).  The call runs on the null stream
 * and returns when the records are complete; it calls no hipMalloc / hipFree and creates no stream or event.
 * All calls on one GPU share that GPU's scratch array: calls that overlap on one GPU race.  The call makes
 * `device` current while it runs and puts the caller's current device back before it returns. */
#define SA_NOALLOC_MAX_PAIRS (1u << 22)
#define SA_NOALLOC_SCRATCH_BYTES (131u << 20)
int sa_align_adjacent_noalloc(int device, const uint8_t* d_slots, const uint16_t* d_lens, uint64_t nseq,
                              const int32_t* S, int32_t gap, uint8_t* d_records, uint32_t stride);

/* ---- records -> what the reference's Alignment holds (host only, no GPU).
 * seq1/seq2 are the ASCII sequences of the pair; a1/a2 receive the aligned
 * strings (columns+1 bytes each, NUL-terminated); stride and row_bytes are the
 * context's sa_record_stride / sa_record_row_bytes. */
int sa_expand_record(const uint8_t* record, uint32_t stride, uint32_t row_bytes, const char* seq1, uint32_t len1,
                     const char* seq2, uint32_t len2, char* a1, char* a2, sa_alignment_stats* st);
int sa_similarity_bp(int32_t matches, int32_t columns);
void sa_default_matrix(int32_t S[400]);

#ifdef __cplusplus
}
#endif
#endif
