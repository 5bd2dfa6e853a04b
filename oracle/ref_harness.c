/*
 * ref_harness.c -- thin driver around the REFERENCE's own kernel and CSV code,
 * compiled against the reference headers where they lie (oracle/Makefile).
 * It adds what the reference program lacks for testing: a single-pair entry
 * point callable through ctypes, and an all-pairs (TRIANGULAR) run, which the
 * reference's main() does not have (it pairs adjacent rows only).
 *
 * Test infrastructure only.  Nothing here restates the algorithm: alignment,
 * parsing and row formatting are the reference's functions, called unchanged.
 */
#include "csv.h"

#include <errno.h>

/* one pair through the reference kernel; aligned strings are copied out */
int ref_align_pair(const char* seq1, int len1, const char* seq2, int len2,
                   char* a1, char* a2, int* score, int* matches, int* mismatches, int* gaps,
                   double* similarity) {
    static ScoringMatrix scoring;
    static int ready = 0;
    if (!ready) { init_scoring_matrix(&scoring); ready = 1; }
    if (len1 <= 0 || len2 <= 0 || len1 >= MAX_SEQ_LEN || len2 >= MAX_SEQ_LEN) return -1;
    char s1[MAX_SEQ_LEN] = {0}, s2[MAX_SEQ_LEN] = {0};
    memcpy(s1, seq1, (size_t)len1);
    memcpy(s2, seq2, (size_t)len2);
    Alignment r = align_sequences(s1, (size_t)len1, s2, (size_t)len2, &scoring);
    strcpy(a1, r.seq1_aligned);
    strcpy(a2, r.seq2_aligned);
    *score = r.score; *matches = r.matches; *mismatches = r.mismatches; *gaps = r.gaps;
    *similarity = r.similarity;
    return 0;
}

int ref_max_seq_len(void) { return MAX_SEQ_LEN; }

/* all pairs i<j of the input file, row-major, seq1 = row i, seq2 = row j,
 * written with the reference's formatter */
int ref_run_triangular(const char* in_path, const char* out_path, long max_rows) {
    setenv("SA_REF_INPUT", in_path, 1);
    setenv("SA_REF_OUTPUT", out_path, 1);
    Files files = get_files();
    if (files.fd < 0 || files.writer.fd < 0) return -1;
    char* cur = files.file_data;
    char* end = files.file_data + files.data_size;
    cur = skip_header(cur, end);
    init_format();
    ScoringMatrix scoring;
    init_scoring_matrix(&scoring);

    size_t cap = 1024, n = 0;
    char (*seqs)[MAX_SEQ_LEN] = malloc(cap * MAX_SEQ_LEN);
    char (*other)[MAX_CSV_LINE - MAX_SEQ_LEN] = malloc(cap * (MAX_CSV_LINE - MAX_SEQ_LEN));
    size_t* lens = malloc(cap * sizeof(size_t));
    while (cur < end && *cur && (max_rows <= 0 || (long)n < max_rows)) {
        if (n == cap) {
            cap *= 2;
            seqs = realloc(seqs, cap * MAX_SEQ_LEN);
            other = realloc(other, cap * (MAX_CSV_LINE - MAX_SEQ_LEN));
            lens = realloc(lens, cap * sizeof(size_t));
        }
        lens[n] = parse_csv_line(&cur, seqs[n], other[n]);
        n++;
    }
    for (size_t i = 0; i + 1 < n; i++) {
        for (size_t j = i + 1; j < n; j++) {
            Alignment r = align_sequences(seqs[i], lens[i], seqs[j], lens[j], &scoring);
            if (files.writer.pos >= WRITE_BUF - MAX_CSV_LINE * 2) flush_buffer(&files.writer);
            Data prev = {seqs[i], other[i], lens[i]};
            Data curr = {seqs[j], other[j], lens[j]};
            files.writer.pos += buffer_output(files.writer.buffer, files.writer.pos, &prev, &curr, &r);
        }
    }
    free_files(&files);
    free(seqs); free(other); free(lens);
    return 0;
}

#ifdef REF_HARNESS_MAIN
int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.csv out.csv [max_rows]\n", argv[0]); return 2; }
    return ref_run_triangular(argv[1], argv[2], argc > 3 ? atol(argv[3]) : 0) ? 1 : 0;
}
#endif
