/*
 * nw_oracle.h -- interface of the CPU restatement in nw_oracle.c.
 * Test infrastructure only (the checker and the timed CPU baseline); the
 * product library never includes or links it.
 */
#ifndef SA_NW_ORACLE_H
#define SA_NW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SA_ORACLE_ADJACENT = 0, SA_ORACLE_TRIANGULAR = 1 };

typedef struct {
    int32_t score;
    int32_t columns;        /* aligned length */
    int32_t matches;
    int32_t mismatches;
    int32_t gaps;
    int32_t similarity_bp;  /* percentage * 100, truncated the way the reference prints it */
} sa_oracle_result;

typedef struct {
    double seconds;         /* first pair aligned -> last buffer flushed */
    uint64_t pairs;
    uint64_t cells;         /* sum of len1*len2 */
    uint64_t rows;
} sa_oracle_run_stats;

void sa_oracle_default_matrix(int32_t out[400]);
const char* sa_oracle_alphabet(void);
int sa_oracle_residue_index(int c);
int sa_oracle_similarity_bp(int matches, int pos);
int sa_oracle_align_ws(const char* seq1, int len1, const char* seq2, int len2,
                       const int32_t* S, int gap, int32_t* work,
                       char* a1, char* a2, sa_oracle_result* out);
int sa_oracle_align(const char* seq1, int len1, const char* seq2, int len2,
                    const int32_t* S, int gap, char* a1, char* a2, sa_oracle_result* out);
size_t sa_oracle_format_row(char* buf, const char* seq1, int len1, const char* seq2, int len2,
                            const char* extra1, const char* extra2,
                            const char* a1, const char* a2, const sa_oracle_result* r);
int sa_oracle_run_file(const char* in_path, const char* out_path, int schedule, int threads,
                       int gap, const int32_t* S, uint64_t max_pairs, sa_oracle_run_stats* st);

#ifdef __cplusplus
}
#endif
#endif
