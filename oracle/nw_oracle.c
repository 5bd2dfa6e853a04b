/*
 * nw_oracle.c -- CPU restatement of the reference's Needleman-Wunsch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against and the CPU baseline that bench.py times; nothing in the product
 * library links, loads or calls it.
 *
 * Parity status: PINNED.  The restatement is checked byte-for-byte against the
 * reference's own golden file results/results.csv (ADJACENT schedule) and
 * against binaries compiled from the reference sources in place (oracle/_ref,
 * see oracle/Makefile) for the ADJACENT and TRIANGULAR schedules.
 *
 * Every function names the reference lines it restates (paths relative to the
 * reference checkout).  The code is written from the behaviour, not copied:
 * the reference is a header-only program with compile-time settings; here the
 * settings are run-time arguments and the matrix lives on the heap so any
 * sequence length works.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "nw_oracle.h"

/* ---------------------------------------------------------------- scoring */

/* Residue alphabet order of the substitution table: include/scoring.h:6. */
static const char ORACLE_ALPHABET[21] = "ARNDCQEGHILKMFPSTWYV";

/* BLOSUM50 restricted to the 20 standard residues in the order above
 * (values as in include/scoring.h:16-36; the table is the published matrix). */
static const int8_t ORACLE_BLOSUM50[400] = {
    5,-2,-1,-2,-1,-1,-1,0,-2,-1,-2,-1,-1,-3,-1,1,0,-3,-2,0,
    -2,7,-1,-2,-4,1,0,-3,0,-4,-3,3,-2,-3,-3,-1,-1,-3,-1,-3,
    -1,-1,7,2,-2,0,0,0,1,-3,-4,0,-2,-4,-2,1,0,-4,-2,-3,
    -2,-2,2,8,-4,0,2,-1,-1,-4,-4,-1,-4,-5,-1,0,-1,-5,-3,-4,
    -1,-4,-2,-4,13,-3,-3,-3,-3,-2,-2,-3,-2,-2,-4,-1,-1,-5,-3,-1,
    -1,1,0,0,-3,7,2,-2,1,-3,-2,2,0,-4,-1,0,-1,-1,-1,-3,
    -1,0,0,2,-3,2,6,-3,0,-4,-3,1,-2,-3,-1,-1,-1,-3,-2,-3,
    0,-3,0,-1,-3,-2,-3,8,-2,-4,-4,-2,-3,-4,-2,0,-2,-3,-3,-4,
    -2,0,1,-1,-3,1,0,-2,10,-4,-3,0,-1,-1,-2,-1,-2,-3,2,-4,
    -1,-4,-3,-4,-2,-3,-4,-4,-4,5,2,-3,2,0,-3,-3,-1,-3,-1,4,
    -2,-3,-4,-4,-2,-2,-3,-4,-3,2,5,-3,3,1,-4,-3,-1,-2,-1,1,
    -1,3,0,-1,-3,2,1,-2,0,-3,-3,6,-2,-4,-1,0,-1,-3,-2,-3,
    -1,-2,-2,-4,-2,0,-2,-3,-1,2,3,-2,7,0,-3,-2,-1,-1,0,1,
    -3,-3,-4,-5,-2,-4,-3,-4,-1,0,1,-4,0,8,-4,-3,-2,1,4,-1,
    -1,-3,-2,-1,-4,-1,-1,-2,-2,-3,-4,-1,-3,-4,10,-1,-1,-4,-3,-3,
    1,-1,1,0,-1,0,-1,0,-1,-3,-3,0,-2,-3,-1,5,2,-4,-2,-2,
    0,-1,0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1,2,5,-3,-2,0,
    -3,-3,-4,-5,-5,-1,-3,-3,-3,-3,-2,-3,-1,1,-4,-4,-3,15,2,-3,
    -2,-1,-2,-3,-3,-1,-2,-3,2,-1,-1,-2,0,4,-3,-2,-2,2,8,-1,
    0,-3,-3,-4,-1,-3,-3,-4,-4,4,1,-3,1,-1,-3,-2,0,-3,-1,5,
};

void sa_oracle_default_matrix(int32_t out[400]) {
    for (int k = 0; k < 400; k++) out[k] = ORACLE_BLOSUM50[k];
}

const char* sa_oracle_alphabet(void) { return ORACLE_ALPHABET; }

/* char -> 0..19, everything else -1: include/scoring.h:13,42-45.
 * Stored +1 so that the zero-initialised entries mean "not a residue". */
static const int8_t ORACLE_LOOKUP1[128] = {
    ['A'] = 1,  ['R'] = 2,  ['N'] = 3,  ['D'] = 4,  ['C'] = 5,  ['Q'] = 6,  ['E'] = 7,
    ['G'] = 8,  ['H'] = 9,  ['I'] = 10, ['L'] = 11, ['K'] = 12, ['M'] = 13, ['F'] = 14,
    ['P'] = 15, ['S'] = 16, ['T'] = 17, ['W'] = 18, ['Y'] = 19, ['V'] = 20,
};

int sa_oracle_residue_index(int c) {
    if (c < 0 || c > 127) return -1;
    return (int)ORACLE_LOOKUP1[c] - 1;
}

/* ------------------------------------------------------------- similarity */

/* The percentage the reference prints: the fp64 quotient is formed first
 * (include/seqalign.h:109), multiplied by 10000 and truncated afterwards
 * (include/csv.h:125).  Not the same as floor(matches*10000/pos). */
int sa_oracle_similarity_bp(int matches, int pos) {
    volatile double q = (double)matches / (double)pos;
    volatile double scaled = q * 10000.0;
    return (int)scaled;
}

/* -------------------------------------------------------------- alignment */

/* One pair: fill (include/seqalign.h:22-38,47-60), traceback (:63-88),
 * reversal and score (:90-96), statistics (:98-110).
 * work must hold (len1+1)*(len2+1) + len1 + len2 ints. a1/a2 must hold len1+len2+1 chars. */
int sa_oracle_align_ws(const char* seq1, int len1, const char* seq2, int len2,
                       const int32_t* S, int gap, int32_t* work,
                       char* a1, char* a2, sa_oracle_result* out) {
    const int cols = len1 + 1;
    int32_t* H = work;
    for (int j = 0; j <= len1; j++) H[j] = j * gap;

    /* residue indices are looked up once per sequence (include/seqalign.h:41-44,52);
     * residues outside the alphabet are undefined behaviour in the reference
     * (include/scoring.h:42, include/seqalign.h:43,55); here they are an error */
    int32_t* idx1 = work + (size_t)(len1 + 1) * (size_t)(len2 + 1);
    int32_t* idx2 = idx1 + len1;
    for (int j = 0; j < len1; j++) if ((idx1[j] = sa_oracle_residue_index(seq1[j])) < 0) return -1;
    for (int i = 0; i < len2; i++) if ((idx2[i] = sa_oracle_residue_index(seq2[i])) < 0) return -1;

    for (int i = 1; i <= len2; i++) {
        const int32_t* up = H + (size_t)(i - 1) * cols;
        int32_t* cur = H + (size_t)i * cols;
        const int r2 = idx2[i - 1];
        cur[0] = i * gap;
        for (int j = 1; j <= len1; j++) {
            int best = up[j - 1] + S[idx1[j - 1] * 20 + r2];   /* [seq1 residue][seq2 residue] */
            const int del = up[j] + gap;
            const int ins = cur[j - 1] + gap;
            if (del > best) best = del;
            if (ins > best) best = ins;
            cur[j] = best;
        }
    }

    /* traceback: DIAG if the cell equals diagonal+substitution, else UP if it
     * equals upper+gap, else LEFT; on the borders UP while i>0, LEFT while j>0 */
    int i = len2, j = len1, n = 0;
    while (i > 0 || j > 0) {
        int move;
        if (i > 0 && j > 0) {
            const int here = H[(size_t)i * cols + j];
            if (here == H[(size_t)(i - 1) * cols + (j - 1)] + S[idx1[j - 1] * 20 + idx2[i - 1]]) move = 0;
            else if (here == H[(size_t)(i - 1) * cols + j] + gap) move = 1;
            else move = 2;
        } else {
            move = (i > 0) ? 1 : 2;
        }
        a1[n] = (move != 1) ? seq1[j - 1] : '-';
        a2[n] = (move != 2) ? seq2[i - 1] : '-';
        n++;
        if (move != 2) i--;
        if (move != 1) j--;
    }
    for (int k = 0; k < n / 2; k++) {
        char t = a1[k]; a1[k] = a1[n - 1 - k]; a1[n - 1 - k] = t;
        t = a2[k]; a2[k] = a2[n - 1 - k]; a2[n - 1 - k] = t;
    }
    a1[n] = a2[n] = '\0';

    /* statistics: a column with equal characters is a match; otherwise a '-'
     * in the FIRST string is a gap; a '-' in the second string counts as a
     * mismatch (include/seqalign.h:100-108) */
    int matches = 0, gaps = 0;
    for (int k = 0; k < n; k++) {
        if (a1[k] == a2[k]) matches++;
        else if (a1[k] == '-') gaps++;
    }
    out->score = H[(size_t)len2 * cols + len1];
    out->columns = n;
    out->matches = matches;
    out->gaps = gaps;
    out->mismatches = n - matches - gaps;
    out->similarity_bp = n > 0 ? sa_oracle_similarity_bp(matches, n) : 0;
    return 0;
}

int sa_oracle_align(const char* seq1, int len1, const char* seq2, int len2,
                    const int32_t* S, int gap, char* a1, char* a2, sa_oracle_result* out) {
    int32_t blosum[400];
    if (!S) { sa_oracle_default_matrix(blosum); S = blosum; }
    int32_t* work = (int32_t*)malloc(sizeof(int32_t) * ((size_t)(len1 + 1) * (size_t)(len2 + 1) + (size_t)len1 + (size_t)len2));
    if (!work) return -2;
    int rc = sa_oracle_align_ws(seq1, len1, seq2, len2, S, gap, work, a1, a2, out);
    free(work);
    return rc;
}

/* ------------------------------------------------------------- formatting */

/* decimal text of a signed int: include/fstring.h:25-55 */
static char* put_int(char* p, int v) {
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    if (v < 0) *p++ = '-';
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

static char* put_mem(char* p, const char* s, size_t n) { memcpy(p, s, n); return p + n; }

/* One output row in the default column layout (include/user.h:44-56):
 * seq1,seq2,extra1,extra2,score,"('A1', 'A2')",matches,mismatches,gaps,P.DD%
 * restating include/csv.h:85-140 for WRITE_CSV_*_POS = 0,4,5,6,7,8,9. */
size_t sa_oracle_format_row(char* buf, const char* seq1, int len1, const char* seq2, int len2,
                            const char* extra1, const char* extra2,
                            const char* a1, const char* a2, const sa_oracle_result* r) {
    char* p = buf;
    p = put_mem(p, seq1, (size_t)len1); *p++ = ',';
    p = put_mem(p, seq2, (size_t)len2); *p++ = ',';
    p = put_mem(p, extra1, strlen(extra1)); *p++ = ',';
    p = put_mem(p, extra2, strlen(extra2)); *p++ = ',';
    p = put_int(p, r->score); *p++ = ',';
    p = put_mem(p, "\"('", 3);
    p = put_mem(p, a1, (size_t)r->columns);
    p = put_mem(p, "', '", 4);
    p = put_mem(p, a2, (size_t)r->columns);
    p = put_mem(p, "')\"", 3); *p++ = ',';
    p = put_int(p, r->matches); *p++ = ',';
    p = put_int(p, r->mismatches); *p++ = ',';
    p = put_int(p, r->gaps); *p++ = ',';
    p = put_int(p, r->similarity_bp / 100);
    *p++ = '.';
    if (r->similarity_bp % 100 < 10) *p++ = '0';
    p = put_int(p, r->similarity_bp % 100);
    *p++ = '%';
    *p++ = '\n';
    return (size_t)(p - buf);
}

/* ---------------------------------------------------------------- ingest */

typedef struct {
    char* seq;        /* NUL-terminated residues */
    char* extra;      /* first non-sequence column, NUL-terminated */
    int len;
} oracle_row;

typedef struct {
    oracle_row* rows;
    size_t n;
    char* arena;
} oracle_table;

/* CSV rows -> sequences + first pass-through column, restating
 * include/csv.h:28-36 (header skipped unconditionally) and :143-209
 * (leading blanks/CR/LF skipped, split on ',', CRLF tolerated, missing final
 * newline tolerated).  seq_pos is READ_CSV_SEQ_POS (include/user.h:25). */
static int load_table(const char* path, int seq_pos, oracle_table* t) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return -1; }
    size_t size = (size_t)sb.st_size;
    char* data = size ? (char*)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0) : NULL;
    if (size && data == MAP_FAILED) { close(fd); return -1; }
    const char* p = data;
    const char* end = data + size;
    while (p < end && *p != '\n') p++;
    if (p < end) p++;

    size_t cap = 1024, n = 0;
    oracle_row* rows = (oracle_row*)malloc(cap * sizeof(oracle_row));
    char* arena = (char*)malloc(size + 16);
    char* w = arena;
    while (p < end && *p) {
        while (p < end && (*p == ' ' || *p == '\r' || *p == '\n')) p++;
        if (p >= end || !*p) break;
        if (n == cap) { cap *= 2; rows = (oracle_row*)realloc(rows, cap * sizeof(oracle_row)); }
        oracle_row* r = &rows[n];
        r->seq = NULL; r->extra = NULL; r->len = 0;
        int col = 0;
        while (p < end && *p && *p != '\n' && *p != '\r') {
            char* field = w;
            while (p < end && *p && *p != ',' && *p != '\n' && *p != '\r') *w++ = *p++;
            *w++ = '\0';
            if (col == seq_pos) { r->seq = field; r->len = (int)(w - field - 1); }
            else if (!r->extra) r->extra = field;
            if (p < end && *p == ',') { p++; col++; }
        }
        if (!r->seq) { r->seq = w; *w++ = '\0'; }
        if (!r->extra) { r->extra = w; *w++ = '\0'; }
        while (p < end && (*p == '\n' || *p == '\r')) p++;
        n++;
    }
    if (data) munmap(data, size);
    close(fd);
    t->rows = rows; t->n = n; t->arena = arena;
    return 0;
}

static void free_table(oracle_table* t) { free(t->rows); free(t->arena); }

/* --------------------------------------------------------------- drivers */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    char* a1; char* a2;         /* aligned strings, stride = 2*maxlen+2 each */
    sa_oracle_result* res;
} batch_out;

typedef struct {
    const oracle_table* t;
    const int32_t* S;
    int gap, maxlen;
    int schedule;
    /* batch description */
    uint64_t first_pair, npairs;   /* canonical pair range of the batch */
    uint64_t lo, hi;               /* this worker's slice of the batch */
    batch_out out;
    int32_t* work;
    int rc;
    pthread_t th;
    pthread_barrier_t* go; pthread_barrier_t* done;
    volatile int* stop;
} worker;

/* canonical TRIANGULAR index <-> (i,j), row-major over i<j */
static void tri_unrank(uint64_t n, uint64_t p, uint64_t* pi, uint64_t* pj) {
    /* row i starts at i*(2n-i-1)/2 */
    uint64_t lo = 0, hi = n - 1;
    while (lo + 1 < hi) {
        uint64_t mid = (lo + hi) / 2;
        uint64_t start = mid * (2 * n - mid - 1) / 2;
        if (start <= p) lo = mid; else hi = mid;
    }
    uint64_t start = lo * (2 * n - lo - 1) / 2;
    *pi = lo; *pj = lo + 1 + (p - start);
}

static void pair_of(const worker* w, uint64_t p, uint64_t* i, uint64_t* j) {
    if (w->schedule == SA_ORACLE_ADJACENT) { *i = p; *j = p + 1; }
    else tri_unrank(w->t->n, p, i, j);
}

static void run_slice(worker* w) {
    const size_t stride = 2 * (size_t)w->maxlen + 2;
    uint64_t i = 0, j = 0;
    if (w->lo < w->hi) pair_of(w, w->first_pair + w->lo, &i, &j);
    for (uint64_t k = w->lo; k < w->hi; k++) {
        const oracle_row* r1 = &w->t->rows[i];
        const oracle_row* r2 = &w->t->rows[j];
        int rc = sa_oracle_align_ws(r1->seq, r1->len, r2->seq, r2->len, w->S, w->gap, w->work,
                                    w->out.a1 + k * stride, w->out.a2 + k * stride, &w->out.res[k]);
        if (rc) w->rc = rc;
        if (w->schedule == SA_ORACLE_ADJACENT) { i++; j++; }
        else if (++j == w->t->n) { i++; j = i + 1; }
    }
}

/* pool worker in the shape of include/thread.h:68-85: wait, run the static
 * slice [lo,hi) of the batch, signal */
static void* worker_main(void* arg) {
    worker* w = (worker*)arg;
    for (;;) {
        pthread_barrier_wait(w->go);
        if (*w->stop) break;
        run_slice(w);
        pthread_barrier_wait(w->done);
    }
    return NULL;
}

/* Whole-file run in the shape of the multithreaded driver src/main.c:25-89:
 * batches of pairs, static even split over the pool with the remainder on the
 * last worker (:54-60), barrier, then serial in-order formatting through one
 * write buffer (:65-73, include/files.h:20-27).  threads<=1 runs inline.
 * TRIANGULAR has no counterpart in the reference (it pairs adjacent rows only,
 * src/main.c:43-52); its pair order is row-major i<j with seq1=row i. */
int sa_oracle_run_file(const char* in_path, const char* out_path, int schedule, int threads,
                       int gap, const int32_t* S_in, uint64_t max_pairs, sa_oracle_run_stats* st) {
    int32_t blosum[400];
    const int32_t* S = S_in;
    if (!S) { sa_oracle_default_matrix(blosum); S = blosum; }
    oracle_table t;
    if (load_table(in_path, 0, &t) != 0) return -1;
    int maxlen = 1;
    size_t maxextra = 0;
    for (size_t k = 0; k < t.n; k++) {
        if (t.rows[k].len > maxlen) maxlen = t.rows[k].len;
        if (strlen(t.rows[k].extra) > maxextra) maxextra = strlen(t.rows[k].extra);
        if (t.rows[k].len == 0) { free_table(&t); return -3; }
    }
    uint64_t total = 0;
    if (t.n >= 2) total = schedule == SA_ORACLE_ADJACENT ? t.n - 1 : (uint64_t)t.n * (t.n - 1) / 2;
    if (max_pairs && total > max_pairs) total = max_pairs;

    int fd = -1;
    const size_t WBUF = 1u << 20;
    char* wbuf = NULL; size_t wpos = 0;
    static const char header[] =
        "sequence1,sequence2,label1,label2,score,alignment,matches,mismatches,gaps,similarity\n";
    if (out_path) {
        fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { free_table(&t); return -4; }
        /* room for one more row past the flush mark: two sequences, two aligned strings of up to
         * len1 + len2 columns, two pass-through fields, numbers and punctuation */
        wbuf = (char*)malloc(WBUF + 6 * (size_t)maxlen + 2 * maxextra + 128);
        memcpy(wbuf, header, sizeof(header) - 1);
        wpos = sizeof(header) - 1;
    }

    if (threads < 1) threads = 1;
    const uint64_t BATCH = 32767;               /* BATCH_SIZE-1 pairs: include/user.h:68 */
    const size_t stride = 2 * (size_t)maxlen + 2;
    batch_out out;
    out.a1 = (char*)malloc(BATCH * stride);
    out.a2 = (char*)malloc(BATCH * stride);
    out.res = (sa_oracle_result*)malloc(BATCH * sizeof(sa_oracle_result));
    worker* ws = (worker*)calloc((size_t)threads, sizeof(worker));
    pthread_barrier_t go, done;
    volatile int stop = 0;
    pthread_barrier_init(&go, NULL, (unsigned)threads + 1);
    pthread_barrier_init(&done, NULL, (unsigned)threads + 1);
    for (int k = 0; k < threads; k++) {
        ws[k].t = &t; ws[k].S = S; ws[k].gap = gap; ws[k].maxlen = maxlen; ws[k].schedule = schedule;
        ws[k].out = out; ws[k].go = &go; ws[k].done = &done; ws[k].stop = &stop;
        ws[k].work = (int32_t*)malloc(sizeof(int32_t) * ((size_t)(maxlen + 1) * (size_t)(maxlen + 1) + 2 * (size_t)maxlen));
        if (threads > 1) pthread_create(&ws[k].th, NULL, worker_main, &ws[k]);
    }

    int rc = 0;
    uint64_t cells = 0;
    const double t0 = now_s();
    for (uint64_t first = 0; first < total; first += BATCH) {
        const uint64_t np = total - first < BATCH ? total - first : BATCH;
        const uint64_t per = np / (uint64_t)threads;
        for (int k = 0; k < threads; k++) {
            ws[k].first_pair = first; ws[k].npairs = np;
            ws[k].lo = (uint64_t)k * per;
            ws[k].hi = k == threads - 1 ? np : (uint64_t)(k + 1) * per;
        }
        if (threads > 1) { pthread_barrier_wait(&go); pthread_barrier_wait(&done); }
        else run_slice(&ws[0]);
        for (int k = 0; k < threads; k++) if (ws[k].rc) rc = ws[k].rc;

        uint64_t i, j;
        pair_of(&ws[0], first, &i, &j);
        for (uint64_t k = 0; k < np; k++) {
            const oracle_row* r1 = &t.rows[i];
            const oracle_row* r2 = &t.rows[j];
            cells += (uint64_t)r1->len * (uint64_t)r2->len;
            if (fd >= 0) {
                if (wpos >= WBUF) { if (write(fd, wbuf, wpos) < 0) rc = -5; wpos = 0; }
                wpos += sa_oracle_format_row(wbuf + wpos, r1->seq, r1->len, r2->seq, r2->len,
                                             r1->extra, r2->extra, out.a1 + k * stride,
                                             out.a2 + k * stride, &out.res[k]);
            }
            if (schedule == SA_ORACLE_ADJACENT) { i++; j++; }
            else if (++j == t.n) { i++; j = i + 1; }
        }
    }
    if (fd >= 0 && wpos) { if (write(fd, wbuf, wpos) < 0) rc = -5; }
    const double t1 = now_s();

    stop = 1;
    if (threads > 1) {
        pthread_barrier_wait(&go);
        for (int k = 0; k < threads; k++) pthread_join(ws[k].th, NULL);
    }
    for (int k = 0; k < threads; k++) free(ws[k].work);
    pthread_barrier_destroy(&go); pthread_barrier_destroy(&done);
    free(ws); free(out.a1); free(out.a2); free(out.res);
    if (fd >= 0) close(fd);
    free(wbuf);
    if (st) { st->seconds = t1 - t0; st->pairs = total; st->cells = cells; st->rows = t.n; }
    free_table(&t);
    return rc;
}
