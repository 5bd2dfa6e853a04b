/*
 * sa_host.h -- the CSV-in / CSV-out driver above sa_hip.h, as a C ABI.
 *
 * sa_settings carries the reference's compile-time settings (include/user.h:7-75)
 * as run-time values under the same names, in lower case; sa_settings_load reads
 * them from a file of "#define NAME value" lines, i.e. from a user.h the
 * reference's GUI has written (scripts/config_schema.py:201-241).  sa_run_csv is
 * the reference's main() (src/main.c:7-134) with the worker pool replaced by the
 * GPU library: parse -> dispatch -> format in pair order -> write.
 */
#ifndef SA_HOST_H
#define SA_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    char input_file[4096];            /* INPUT_FILE  (include/user.h:59) */
    char output_file[4096];           /* OUTPUT_FILE (include/user.h:60) */
    uint32_t max_csv_line;            /* MAX_CSV_LINE, informational: rows are not copied into fixed buffers */
    uint32_t max_seq_len;             /* MAX_SEQ_LEN: sequences must be shorter than this (include/user.h:10) */
    int32_t gap_penalty;              /* GAP_PENALTY (include/user.h:13) */
    uint32_t read_csv_seq_pos;        /* READ_CSV_SEQ_POS */
    uint32_t read_csv_cols;           /* READ_CSV_COLS */
    char write_csv_header[1024];      /* WRITE_CSV_HEADER */
    uint32_t write_csv_seq1_pos;      /* WRITE_CSV_SEQ1_POS (sequence 2 follows it) */
    uint32_t write_csv_score_pos;
    uint32_t write_csv_align_pos;
    uint32_t write_csv_matches_pos;
    uint32_t write_csv_mismatches_pos;
    uint32_t write_csv_gaps_pos;
    uint32_t write_csv_similarity_pos;
    char write_csv_align_fmt[128];    /* WRITE_CSV_ALIGN_FMT: two %s */
    int32_t mode_write;               /* MODE_WRITE: 0 = align only (include/user.h:65, src/main.c:111-116) */
    int32_t similarity_analysis;      /* SIMILARITY_ANALYSIS */
    uint32_t batch_size;              /* BATCH_SIZE: rows per batch; one GPU submission covers 4 batches unless pairs_per_submit is set */
    /* settings the reference does not have */
    int32_t schedule;                 /* SA_SCHEDULE_ADJACENT (the reference's) or SA_SCHEDULE_TRIANGULAR */
    int32_t num_devices;              /* GPUs to use at most, chunks are dealt to them in turn; a job too small to keep them busy uses
                                         fewer (one per 30 ms of expected kernel time: a further GPU costs a copy of the sequences
                                         and takes the same turns at the one output file) */
    int32_t devices[8];
    int32_t format_threads;           /* host threads that expand and format records; 0 = min(16, cores) */
    uint64_t pairs_per_submit;        /* 0 = batch_size * 4 (ADJACENT), batch_size * 64 (TRIANGULAR) */
    int32_t pairing;                  /* sa_set_pairing() of every context; default 1 */
    int32_t device_format;            /* 1 (default): rows of CSV text are written on the GPU (sa_submit_text); 0: records come back and host threads format them */
    char scoring_matrix_file[4096];   /* SA_SCORING_MATRIX: file with the 20x20 substitution matrix (sa_load_matrix_file); empty = the
                                         reference's BLOSUM50 literal (include/scoring.h:16-36) */
    int32_t cooperative;              /* SA_COOPERATIVE: sa_set_cooperative() of every context; default 0 (by pair count) */
    int32_t compact_records;          /* SA_COMPACT_RECORDS: 1 (default): records that travel to the host (host formatting, MODE_WRITE 0)
                                         are packed on the device first (sa_set_compact_records); 0: records of the full stride */
    int32_t force_devices;            /* SA_FORCE_DEVICES: 1 = use all num_devices contexts whatever the size of the job (tests, measurements) */
} sa_settings;

typedef struct {
    uint64_t rows;                    /* sequences read */
    uint64_t pairs;
    uint64_t cells;                   /* sum of len1*len2 */
    uint64_t bytes_written;
    uint64_t chunks_done;
    double seconds;                   /* first byte parsed -> last byte written, set-up excluded (src/main.c:30,129) */
    double seconds_ingest;            /* of which: map, parse, validate, encode */
    double seconds_upload;            /*           sequences to the device(s) */
    double seconds_gpu_wait;          /*           waiting for records (kernels + device-to-host copies not hidden) */
    double seconds_format;            /*           expanding and formatting records */
    double seconds_write;             /*           write() calls */
    char message[256];                /* error text when the return code is not 0 */
    /* per device (index = position in sa_settings.devices): seconds its submit thread spent in sa_submit* (work lists
     * included), and its drain thread waiting for the GPU, formatting records and waiting for its writer */
    double device_submit[8];
    double device_wait[8];
    double device_format[8];
    double device_write[8];
    uint64_t record_bytes;            /* bytes of records copied from the devices to the host (0 when the GPUs wrote the text) */
    int32_t devices_used;             /* of sa_settings.num_devices: as many as the job could feed (see sa_run_csv) */
} sa_run_report;

void sa_settings_default(sa_settings* st);
/* reads "#define NAME value" lines; unknown names are ignored; returns 0 or a negative code */
int sa_settings_load(sa_settings* st, const char* path);
/* the reference's checks on a settings file (scripts/config_schema.py:100-198); message receives the first problem */
int sa_settings_validate(const sa_settings* st, char* message, uint32_t message_cap);
int sa_run_csv(const sa_settings* st, sa_run_report* report);
/* The scoring matrix as a setting: stands for the literal in init_scoring_matrix (include/scoring.h:15-52).  The file
 * holds 400 integers, row-major, S[a][b] with a = residue of sequence 1 and b = residue of sequence 2 in the order
 * ARNDCQEGHILKMFPSTWYV (include/scoring.h:6); integers are separated by blanks, commas, semicolons or line ends; lines
 * that start with '#' and tokens that are single letters (a header row, a leading letter per row) are skipped.
 * message (may be NULL) receives the problem when the result is not 0. */
int sa_load_matrix_file(const char* path, int32_t S[400], char* message, uint32_t message_cap);

#ifdef __cplusplus
}
#endif
#endif
