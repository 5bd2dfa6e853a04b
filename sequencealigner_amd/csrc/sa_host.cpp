// sa_host.cpp -- the CSV-in / CSV-out program around the GPU library: settings,
// ingest, chunked dispatch to one or more GPUs, record expansion, row
// formatting and the ordered writer.  It restates the host side of the
// reference's driver (src/main.c:7-134) with the worker pool replaced by
// sa_submit / sa_next (include/sa_hip.h).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sa_hip.h"
#include "../../include/sa_host.h"
#include "sa_parallel.h"

namespace {

double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct Row {
    const char* seq;
    uint32_t len;
    const char* other;      // first column that is not the sequence (what the reference prints, include/csv.h:91-95)
    uint32_t other_len;
};

struct Table {
    char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    Row* rows_data = nullptr;     // rows in file order (filled by several threads, never value-initialised)
    size_t nrows = 0;
    struct RowView {
        const Row* p; size_t n;
        size_t size() const { return n; }
        const Row& operator[](size_t k) const { return p[k]; }
    } rows{nullptr, 0};
    ~Table() {
        free(rows_data);
        if (data && data != MAP_FAILED) munmap(data, size);
        if (fd >= 0) close(fd);
    }
};

// Rows of the input file.  Follows include/csv.h:28-36 (the first line is a
// header and is skipped unconditionally) and include/csv.h:143-209 (leading
// blanks / CR / LF skipped, fields split on ',', a row ends at CR, LF or NUL,
// a missing final newline is fine).
int load_table(const sa_settings* st, Table& t, std::string& err) {
    t.fd = open(st->input_file, O_RDONLY);
    if (t.fd < 0) { err = std::string("cannot open input file: ") + strerror(errno); return SA_ERR_ARG; }
    struct stat sb;
    if (fstat(t.fd, &sb) != 0) { err = "cannot stat input file"; return SA_ERR_ARG; }
    t.size = (size_t)sb.st_size;
    if (t.size == 0) return SA_OK;
    t.data = (char*)mmap(nullptr, t.size, PROT_READ, MAP_PRIVATE, t.fd, 0);
    if (t.data == MAP_FAILED) { err = "cannot map input file"; return SA_ERR_ARG; }
    madvise(t.data, t.size, MADV_SEQUENTIAL);
    const char* const end = t.data + t.size;
    const char* body = t.data;
    while (body < end && *body != '\n') body++;
    if (body < end) body++;
    // A NUL ends the table (include/csv.h:150); lines are then parsed in slices that start after a newline.
    const char* stop = (const char*)memchr(body, 0, (size_t)(end - body));
    if (!stop) stop = end;
    const uint32_t seq_pos = st->read_csv_seq_pos;
    // rows == nullptr: count only
    auto parse = [seq_pos](const char* p, const char* lim, Row* rows) -> size_t {
        size_t count = 0;
        while (p < lim) {
            while (p < lim && (*p == ' ' || *p == '\r' || *p == '\n')) p++;
            if (p >= lim) break;
            Row r = {nullptr, 0, "", 0};
            uint32_t col = 0;
            bool have_other = false;
            while (p < lim && *p != '\n' && *p != '\r') {
                const char* f = p;
                while (p < lim && *p != ',' && *p != '\n' && *p != '\r') p++;
                if (col == seq_pos) { r.seq = f; r.len = (uint32_t)(p - f); }
                else if (!have_other) { r.other = f; r.other_len = (uint32_t)(p - f); have_other = true; }
                if (p < lim && *p == ',') { p++; col++; }
            }
            if (!r.seq) { r.seq = p; r.len = 0; }
            if (rows) rows[count] = r;
            count++;
            while (p < lim && (*p == '\n' || *p == '\r')) p++;
        }
        return count;
    };
    // a slice owns the lines that begin in it: both ends move forward to the next line start.  Two passes
    // over the slices: count the rows, then write them to their place in one array.
    const uint64_t bytes = (uint64_t)(stop - body);
    const unsigned maxT = sa_host_threads();
    std::vector<const char*> from(maxT + 1, stop);
    std::vector<size_t> count(maxT + 1, 0);
    const unsigned T = sa_parallel_for(bytes, 1 << 22, [&](uint64_t lo, uint64_t hi, unsigned th) {
        const char* a = body + lo;
        const char* b = body + hi;
        if (lo) { while (a < stop && a[-1] != '\n' && a[-1] != '\r') a++; }
        if (hi < bytes) { while (b < stop && b[-1] != '\n' && b[-1] != '\r') b++; }
        from[th] = a;
        count[th] = parse(a, b, nullptr);
    });
    from[T] = stop;
    std::vector<size_t> base(T + 1, 0);
    for (unsigned k = 0; k < T; k++) base[k + 1] = base[k] + count[k];
    t.nrows = base[T];
    t.rows_data = (Row*)malloc(std::max<size_t>(1, t.nrows) * sizeof(Row));
    if (!t.rows_data) { err = "out of memory for the row table"; return SA_ERR_NOMEM; }
    sa_parallel_for(T, 1, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t k = lo; k < hi; k++) parse(from[k], from[k + 1], t.rows_data + base[k]);
    });
    t.rows = Table::RowView{t.rows_data, t.nrows};
    return SA_OK;
}

// Output layout: which output column shows what.  Restates init_format
// (include/csv.h:38-82): the columns named by the WRITE_CSV_*_POS settings are
// taken, the free ones are handed out in adjacent pairs to the pass-through
// columns in order.
struct Layout {
    uint32_t ncols = 0;
    std::vector<int> kind;      // per output column: K_* below
    std::string part[3];        // alignment format split at its two "%s"
};
enum { K_NONE, K_SEQ1, K_SEQ2, K_SCORE, K_ALIGN, K_MATCHES, K_MISMATCHES, K_GAPS, K_SIMILARITY, K_OTHER1, K_OTHER2 };

int make_layout(const sa_settings* st, Layout& L, std::string& err) {
    const uint32_t sim = st->similarity_analysis ? 1 : 0;
    L.ncols = 4 + (st->read_csv_cols - 1) * 2 + sim * 4;
    L.kind.assign(L.ncols + 2, K_NONE);
    auto take = [&](uint32_t pos, int kind) -> bool {
        if (pos >= L.ncols || L.kind[pos] != K_NONE) return false;
        L.kind[pos] = kind;
        return true;
    };
    bool ok = take(st->write_csv_seq1_pos, K_SEQ1) && take(st->write_csv_seq1_pos + 1, K_SEQ2) &&
              take(st->write_csv_score_pos, K_SCORE) && take(st->write_csv_align_pos, K_ALIGN);
    if (sim)
        ok = ok && take(st->write_csv_matches_pos, K_MATCHES) && take(st->write_csv_mismatches_pos, K_MISMATCHES) &&
             take(st->write_csv_gaps_pos, K_GAPS) && take(st->write_csv_similarity_pos, K_SIMILARITY);
    if (!ok) { err = "output column positions overlap or exceed the column count"; return SA_ERR_ARG; }
    uint32_t idx = 0;
    for (uint32_t i = 0; i + 1 < L.ncols && idx < st->read_csv_cols - 1; i++) {
        if (L.kind[i] == K_NONE) {
            L.kind[i] = K_OTHER1;
            L.kind[i + 1] = K_OTHER2;
            idx++;
            i++;
        }
    }
    const char* f = st->write_csv_align_fmt;
    int part = 0;
    const char* start = f;
    while (*f) {
        if (f[0] == '%' && f[1] == 's' && part < 2) {
            L.part[part++] = std::string(start, f);
            f += 2;
            start = f;
        } else {
            f++;
        }
    }
    L.part[part] = std::string(start, f);
    if (part != 2) { err = "WRITE_CSV_ALIGN_FMT needs exactly two %s"; return SA_ERR_ARG; }
    return SA_OK;
}

inline char* put_int(char* p, int v) {   // decimal text as include/fstring.h:25-55 produces it
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    if (v < 0) *p++ = '-';
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}
inline char* put_mem(char* p, const char* s, size_t n) { memcpy(p, s, n); return p + n; }

// One output row; restates buffer_output (include/csv.h:85-140).
size_t format_row(char* buf, const Layout& L, const Row& r1, const Row& r2, const char* a1, const char* a2,
                  const sa_alignment_stats& s) {
    char* p = buf;
    for (uint32_t col = 0; col < L.ncols; col++) {
        if (col) *p++ = ',';
        switch (L.kind[col]) {
            case K_OTHER1: p = put_mem(p, r1.other, r1.other_len); break;
            case K_OTHER2: p = put_mem(p, r2.other, r2.other_len); break;
            case K_SEQ1: p = put_mem(p, r1.seq, r1.len); break;
            case K_SEQ2: p = put_mem(p, r2.seq, r2.len); break;
            case K_SCORE: p = put_int(p, s.score); break;
            case K_ALIGN:
                p = put_mem(p, L.part[0].data(), L.part[0].size());
                p = put_mem(p, a1, (size_t)s.columns);
                p = put_mem(p, L.part[1].data(), L.part[1].size());
                p = put_mem(p, a2, (size_t)s.columns);
                p = put_mem(p, L.part[2].data(), L.part[2].size());
                break;
            case K_MATCHES: p = put_int(p, s.matches); break;
            case K_MISMATCHES: p = put_int(p, s.mismatches); break;
            case K_GAPS: p = put_int(p, s.gaps); break;
            case K_SIMILARITY:
                p = put_int(p, s.similarity_bp / 100);
                *p++ = '.';
                if (s.similarity_bp % 100 < 10) *p++ = '0';
                p = put_int(p, s.similarity_bp % 100);
                *p++ = '%';
                break;
            default: break;
        }
    }
    *p++ = '\n';
    return (size_t)(p - buf);
}

bool write_all(int fd, const char* p, size_t n) {
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w;
    }
    return true;
}

// Blocks for a chunk's part of the output file, taken in one call before the chunk is written: on a disk file system
// the write otherwise allocates them piece by piece (500 MB: 0.048 s without, 0.041 s with, tools/write_bench.cpp);
// where the file system cannot do it the call fails and nothing is lost.
bool reserve_blocks(int fd, uint64_t at, uint64_t nbytes) {
    return nbytes && fallocate(fd, 0, (off_t)at, (off_t)nbytes) == 0;
}

bool pwrite_all(int fd, const char* p, size_t n, uint64_t off) {
    while (n) {
        ssize_t w = pwrite(fd, p, n, (off_t)off);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w; off += (uint64_t)w;
    }
    return true;
}

// A chunk's bytes to their place in the output file.  write() into one file is one copy at a time however many threads
// call it (the inode's lock); a shared mapping of the chunk's pages is filled by several threads at once.  Used for
// files in memory (tmpfs) whose blocks were reserved first: the 500 MB of the 4 001 280-row data set 0.069 -> 0.042 s,
// the whole run 0.119 -> 0.090 s (tools/e2e_write_ab.py, same box, alternating).  Not on a disk file system: reserved
// but unwritten extents are converted page by page at the faults there, 0.041 -> 0.16-0.21 s on the same box; and
// not over a hole, where a full disk would show as a signal instead of an error code.
bool file_in_memory(int fd) {
    struct statfs fs;
    return fstatfs(fd, &fs) == 0 && (unsigned long)fs.f_type == 0x01021994ul;      // TMPFS_MAGIC
}
bool write_region(int fd, const char* p, size_t n, uint64_t off, bool mappable, unsigned threads) {
    constexpr size_t kMapFrom = 1u << 20;
    if (!mappable || n < kMapFrom || threads < 2) return pwrite_all(fd, p, n, off);
    const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE);
    const uint64_t start = off & ~(page - 1);
    const size_t lead = (size_t)(off - start);
    void* m = mmap(nullptr, n + lead, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)start);
    if (m == MAP_FAILED) return pwrite_all(fd, p, n, off);
    char* const dst = (char*)m + lead;
    const unsigned t_count = (unsigned)std::min<size_t>(threads, n / kMapFrom + 1);
    std::vector<std::thread> th;
    auto part = [&](unsigned t) {
        const size_t lo = n * t / t_count, hi = n * (t + 1) / t_count;
        memcpy(dst + lo, p + lo, hi - lo);
    };
    for (unsigned t = 1; t < t_count; t++) th.emplace_back(part, t);
    part(0);
    for (auto& x : th) x.join();
    munmap(m, n + lead);
    return true;
}

struct Chunk { uint64_t r0, r1; };

}  // namespace

extern "C" {

void sa_settings_default(sa_settings* st) {
    memset(st, 0, sizeof(*st));
    st->max_csv_line = 256;
    st->max_seq_len = 64;
    st->gap_penalty = -4;
    st->read_csv_seq_pos = 0;
    st->read_csv_cols = 2;
    snprintf(st->write_csv_header, sizeof(st->write_csv_header), "%s",
             "sequence1,sequence2,label1,label2,score,alignment,matches,mismatches,gaps,similarity\n");
    st->write_csv_seq1_pos = 0;
    st->write_csv_score_pos = 4;
    st->write_csv_align_pos = 5;
    st->write_csv_matches_pos = 6;
    st->write_csv_mismatches_pos = 7;
    st->write_csv_gaps_pos = 8;
    st->write_csv_similarity_pos = 9;
    snprintf(st->write_csv_align_fmt, sizeof(st->write_csv_align_fmt), "%s", "\"('%s', '%s')\"");
    st->mode_write = 1;
    st->similarity_analysis = 1;
    st->batch_size = 32768;
    st->schedule = SA_SCHEDULE_ADJACENT;
    st->num_devices = 1;
    for (int k = 0; k < 8; k++) st->devices[k] = k;
    st->format_threads = 0;
    st->pairs_per_submit = 0;
    st->pairing = 1;
    st->device_format = 1;
    st->compact_records = 1;
}

static bool parse_c_string(const char* p, char* out, size_t cap) {
    while (*p == ' ' || *p == '\t') p++;
    if (*p != '"') return false;
    p++;
    size_t n = 0;
    while (*p && *p != '"') {
        char ch = *p++;
        if (ch == '\\' && *p) {
            const char e = *p++;
            ch = e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e;
        }
        if (n + 1 >= cap) return false;
        out[n++] = ch;
    }
    out[n] = '\0';
    return *p == '"';
}

int sa_settings_load(sa_settings* st, const char* path) {
    if (!st || !path) return SA_ERR_ARG;
    FILE* f = fopen(path, "r");
    if (!f) return SA_ERR_ARG;
    char line[8192];
    int rc = SA_OK;
    while (fgets(line, sizeof(line), f)) {
        char name[64];
        int off = 0;
        if (sscanf(line, " #define %63s %n", name, &off) < 1 || off == 0) continue;
        const char* v = line + off;
        const std::string k(name);
        auto num = [&](long& out) { char* e; out = strtol(v, &e, 0); return e != v; };
        long x;
        if (k == "INPUT_FILE") { if (!parse_c_string(v, st->input_file, sizeof(st->input_file))) st->input_file[0] = 0; }
        else if (k == "OUTPUT_FILE") { if (!parse_c_string(v, st->output_file, sizeof(st->output_file))) st->output_file[0] = 0; }
        else if (k == "WRITE_CSV_HEADER") { if (!parse_c_string(v, st->write_csv_header, sizeof(st->write_csv_header))) rc = SA_ERR_ARG; }
        else if (k == "WRITE_CSV_ALIGN_FMT") { if (!parse_c_string(v, st->write_csv_align_fmt, sizeof(st->write_csv_align_fmt))) rc = SA_ERR_ARG; }
        else if (k == "MAX_CSV_LINE") { if (num(x)) st->max_csv_line = (uint32_t)x; }
        else if (k == "MAX_SEQ_LEN") { if (num(x)) st->max_seq_len = (uint32_t)x; }
        else if (k == "GAP_PENALTY") { if (num(x)) st->gap_penalty = (int32_t)x; }
        else if (k == "READ_CSV_SEQ_POS") { if (num(x)) st->read_csv_seq_pos = (uint32_t)x; }
        else if (k == "READ_CSV_COLS") { if (num(x)) st->read_csv_cols = (uint32_t)x; }
        else if (k == "WRITE_CSV_SEQ1_POS") { if (num(x)) st->write_csv_seq1_pos = (uint32_t)x; }
        else if (k == "WRITE_CSV_SCORE_POS") { if (num(x)) st->write_csv_score_pos = (uint32_t)x; }
        else if (k == "WRITE_CSV_ALIGN_POS") { if (num(x)) st->write_csv_align_pos = (uint32_t)x; }
        else if (k == "WRITE_CSV_MATCHES_POS") { if (num(x)) st->write_csv_matches_pos = (uint32_t)x; }
        else if (k == "WRITE_CSV_MISMATCHES_POS") { if (num(x)) st->write_csv_mismatches_pos = (uint32_t)x; }
        else if (k == "WRITE_CSV_GAPS_POS") { if (num(x)) st->write_csv_gaps_pos = (uint32_t)x; }
        else if (k == "WRITE_CSV_SIMILARITY_POS") { if (num(x)) st->write_csv_similarity_pos = (uint32_t)x; }
        else if (k == "SIMILARITY_ANALYSIS") { if (num(x)) st->similarity_analysis = (int32_t)x; }
        else if (k == "MODE_WRITE") { if (num(x)) st->mode_write = (int32_t)x; }
        else if (k == "BATCH_SIZE") { if (num(x)) st->batch_size = (uint32_t)x; }
        else if (k == "SA_SCHEDULE") { if (num(x)) st->schedule = (int32_t)x; }
        else if (k == "SA_NUM_DEVICES") { if (num(x)) st->num_devices = (int32_t)x; }
        else if (k == "SA_COOPERATIVE") { if (num(x)) st->cooperative = (int32_t)x; }
        else if (k == "SA_COMPACT_RECORDS") { if (num(x)) st->compact_records = (int32_t)x; }
        else if (k == "SA_FORCE_DEVICES") { if (num(x)) st->force_devices = (int32_t)x; }
        else if (k == "SA_SCORING_MATRIX") { if (!parse_c_string(v, st->scoring_matrix_file, sizeof(st->scoring_matrix_file))) st->scoring_matrix_file[0] = 0; }
    }
    fclose(f);
    return rc;
}

int sa_load_matrix_file(const char* path, int32_t S[400], char* message, uint32_t cap) {
    auto say = [&](const std::string& m) { if (message && cap) snprintf(message, cap, "%s", m.c_str()); return SA_ERR_ARG; };
    if (!path || !S) return SA_ERR_ARG;
    FILE* f = fopen(path, "r");
    if (!f) return say(std::string("cannot open scoring matrix file: ") + strerror(errno));
    char line[8192];
    int count = 0;
    bool bad = false;
    std::string bad_token;
    while (!bad && fgets(line, sizeof(line), f)) {
        const char* p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (*p == '#') continue;
        while (*p) {
            while (*p == ' ' || *p == '\t' || *p == ',' || *p == ';' || *p == '\r' || *p == '\n') p++;
            if (!*p) break;
            const char* tok = p;
            while (*p && *p != ' ' && *p != '\t' && *p != ',' && *p != ';' && *p != '\r' && *p != '\n') p++;
            const size_t len = (size_t)(p - tok);
            if (len == 1 && ((*tok >= 'A' && *tok <= 'Z') || (*tok >= 'a' && *tok <= 'z') || *tok == '*')) continue;
            char* e = nullptr;
            const long v = strtol(tok, &e, 10);
            if (e != p || v < -32768 || v > 32767) { bad = true; bad_token.assign(tok, len); break; }
            if (count < 400) S[count] = (int32_t)v;
            count++;
        }
    }
    fclose(f);
    if (bad) return say("scoring matrix file: '" + bad_token + "' is not an integer");
    if (count != 400) return say("scoring matrix file holds " + std::to_string(count) + " integers, 400 (20 x 20) are needed");
    if (message && cap) message[0] = 0;
    return SA_OK;
}

int sa_settings_validate(const sa_settings* st, char* message, uint32_t cap) {
    auto say = [&](const char* m) { if (message && cap) snprintf(message, cap, "%s", m); return SA_ERR_ARG; };
    if (!st) return SA_ERR_ARG;
    if (st->gap_penalty >= 0) return say("GAP_PENALTY must be negative");
    if (st->max_csv_line < 32) return say("MAX_CSV_LINE must be at least 32");
    if (st->max_seq_len < 2) return say("MAX_SEQ_LEN must be at least 2");
    if (st->read_csv_cols < 1) return say("READ_CSV_COLS must be at least 1");
    if (st->read_csv_seq_pos >= st->read_csv_cols) return say("READ_CSV_SEQ_POS must be below READ_CSV_COLS");
    if (st->batch_size < 2) return say("BATCH_SIZE must be at least 2");
    const uint32_t want = 2 * st->read_csv_cols + 2 + (st->similarity_analysis ? 4 : 0);
    uint32_t have = 1;
    for (const char* p = st->write_csv_header; *p; p++) have += *p == ',';
    if (have != want) return say("WRITE_CSV_HEADER must have 2*READ_CSV_COLS+2 columns (+4 with SIMILARITY_ANALYSIS)");
    Layout L;
    std::string err;
    if (make_layout(st, L, err) != SA_OK) return say(err.c_str());
    if (st->scoring_matrix_file[0]) {
        int32_t S[400];
        const int rc = sa_load_matrix_file(st->scoring_matrix_file, S, message, cap);
        if (rc) return rc;
    }
    if (message && cap) message[0] = 0;
    return SA_OK;
}

int sa_run_csv(const sa_settings* st, sa_run_report* rep) {
    if (!st || !rep) return SA_ERR_ARG;
    memset(rep, 0, sizeof(*rep));
    std::string err;
    auto fail = [&](int code) {
        snprintf(rep->message, sizeof(rep->message), "%s", err.empty() ? sa_strerror(code) : err.c_str());
        return code;
    };
    if (!*st->input_file) { err = "INPUT_FILE is not set"; return fail(SA_ERR_ARG); }
    if (st->mode_write && !*st->output_file) { err = "OUTPUT_FILE is not set"; return fail(SA_ERR_ARG); }
    if (st->read_csv_cols < 1 || st->read_csv_seq_pos >= st->read_csv_cols) { err = "READ_CSV_SEQ_POS must be below READ_CSV_COLS"; return fail(SA_ERR_ARG); }
    if (st->num_devices < 1 || st->num_devices > 8) { err = "1 to 8 devices"; return fail(SA_ERR_ARG); }
    Layout L;
    int rc = make_layout(st, L, err);
    if (rc) return fail(rc);

    // contexts first: like the reference's pool, they are set up outside the timed window (src/main.c:10,30)
    const int ndev = st->num_devices;
    std::vector<sa_ctx*> ctx((size_t)ndev, nullptr);
    std::thread preload;          // loads device code while the input is read (below); joined before anything is destroyed
    auto cleanup = [&]() {
        if (preload.joinable()) preload.join();
        for (auto c : ctx) sa_destroy(c);
    };
    // the reference's buffers hold sequences shorter than MAX_SEQ_LEN (NUL included, include/thread.h:27-29)
    const uint32_t max_len = st->max_seq_len ? st->max_seq_len - 1 : 0;
    if (max_len == 0 || max_len > SA_MAX_SEQ_LEN) { err = "MAX_SEQ_LEN must be between 2 and 4097"; return fail(SA_ERR_LENGTH); }
    int32_t matrix[400];
    const int32_t* S = nullptr;          // NULL: the library's BLOSUM50
    if (st->scoring_matrix_file[0]) {
        char msg[200];
        rc = sa_load_matrix_file(st->scoring_matrix_file, matrix, msg, sizeof(msg));
        if (rc) { err = msg; return fail(rc); }
        S = matrix;
    }
    // rows of more columns than the device formatter's layout holds are formatted on the host (any column count)
    const bool text_mode = st->mode_write && st->device_format && L.ncols <= 32;
    const bool compact = !text_mode && st->compact_records != 0;       // records travel to the host: without their padding
    for (int d = 0; d < ndev; d++) {
        rc = sa_init(&ctx[(size_t)d], S, st->gap_penalty, max_len, st->devices[d]);
        if (rc == SA_OK) rc = sa_set_pairing(ctx[(size_t)d], st->pairing);
        if (rc == SA_OK) rc = sa_set_cooperative(ctx[(size_t)d], st->cooperative);
        if (rc == SA_OK) rc = sa_set_compact_records(ctx[(size_t)d], compact ? 1 : 0);
        if (rc == SA_OK && !text_mode) {
            // the record buffers of both slots, as part of the set-up (at most 256 MiB each here; larger submissions
            // grow them when they come)
            const uint64_t cap0 = st->pairs_per_submit ? st->pairs_per_submit
                                                       : (uint64_t)st->batch_size * (st->schedule == SA_SCHEDULE_TRIANGULAR ? 64 : 4);
            const uint64_t per_pair = max_len > 64 ? 16 + 2 * (uint64_t)max_len + 16 : 8 + (uint64_t)max_len + 16;
            const uint64_t pairs0 = std::max<uint64_t>(1, std::min<uint64_t>(cap0, (256ull << 20) / per_pair));
            rc = sa_reserve_submission(ctx[(size_t)d], pairs0, max_len);
        }
        if (rc == SA_OK && text_mode) {
            // text submissions: the same cap as below, the worst-case row of a sequence of MAX_SEQ_LEN - 1 residues
            uint64_t cap0 = st->pairs_per_submit ? st->pairs_per_submit
                                                 : (uint64_t)st->batch_size * (st->schedule == SA_SCHEDULE_TRIANGULAR ? 64 : 4);
            cap0 = std::max<uint64_t>(1, std::min<uint64_t>(cap0, std::max<uint64_t>(1024, (1ull << 30) / (6ull * st->max_seq_len + 600))));
            const uint64_t text0 = std::min<uint64_t>(cap0 * (6ull * max_len + 256), 256ull << 20);
            rc = sa_reserve_text(ctx[(size_t)d], cap0, text0);
        }
        if (rc) { cleanup(); return fail(rc); }
    }

    int out_fd = -1;
    if (st->mode_write) {
        // read access too: chunks are written through shared mappings; a file that only allows writing is written with pwrite
        out_fd = open(st->output_file, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (out_fd < 0 && errno == EACCES) out_fd = open(st->output_file, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (out_fd < 0) { err = std::string("cannot open output file: ") + strerror(errno); cleanup(); return fail(SA_ERR_ARG); }
    }

    const double t0 = now_s();
    double t_format = 0, t_write = 0, t_wait = 0;
    // The device code this job will launch is loaded while the input is read (a thread of its own: the runtime would
    // otherwise load each code object at the first launch of one of its kernels, 23 ms of a fresh process's 120 on the
    // 4 001 280-row data set).  A failure here is not the job's: the launches would report it.
    preload = std::thread([&]() { (void)sa_preload_kernels(ctx[0], st->schedule, text_mode ? 1 : 0, compact ? 1 : 0); });
    Table tab;
    rc = load_table(st, tab, err);
    if (rc) { cleanup(); if (out_fd >= 0) close(out_fd); return fail(rc); }
    const uint64_t n = tab.rows.size();
    rep->rows = n;

    // encode + validate (the reference has undefined behaviour here; see include/scoring.h:42, include/csv.h:174,181)
    // plain arrays: a vector would clear 150 MB on one thread before the threads below fill them
    std::unique_ptr<uint64_t[]> offs_mem(new (std::nothrow) uint64_t[n + 1]);
    std::unique_ptr<uint32_t[]> lens_mem(new (std::nothrow) uint32_t[n + 1]);
    if (!offs_mem || !lens_mem) { err = "out of memory"; cleanup(); if (out_fd >= 0) close(out_fd); return fail(SA_ERR_NOMEM); }
    uint64_t* const offs = offs_mem.get();
    uint32_t* const lens = lens_mem.get();
    uint32_t longest_seq = 0, longest_other = 0;     // what the row buffers of the host formatter are sized by
    {
        // lengths, their range and the first row outside it, on the host threads; offsets as prefix sums
        struct Part { uint32_t seq = 0, other = 0; uint64_t bad = UINT64_MAX; };
        std::vector<Part> part(sa_host_threads());
        sa_parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned t) {
            Part q;
            for (uint64_t s = lo; s < hi; s++) {
                const Row& r = tab.rows[s];
                q.seq = std::max(q.seq, r.len);
                q.other = std::max(q.other, r.other_len);
                if ((r.len == 0 || r.len > max_len) && q.bad == UINT64_MAX) q.bad = s;
                lens[s] = r.len;
            }
            part[t] = q;
        });
        uint64_t bad = UINT64_MAX;
        for (const Part& q : part) {
            longest_seq = std::max(longest_seq, q.seq);
            longest_other = std::max(longest_other, q.other);
            bad = std::min(bad, q.bad);
        }
        if (bad != UINT64_MAX) {
            err = "row " + std::to_string(bad + 2) + ": sequence length " + std::to_string(tab.rows[bad].len) + " outside 1.." + std::to_string(max_len);
            cleanup(); if (out_fd >= 0) close(out_fd);
            return fail(SA_ERR_LENGTH);
        }
        prefix_sums(n, offs, [&](uint64_t s) { return (uint64_t)lens[s]; });
    }
    const uint64_t total = n ? offs[n] : 0;
    std::unique_ptr<uint8_t[]> codes_mem(new (std::nothrow) uint8_t[total + 1]);
    if (!codes_mem) { err = "out of memory"; cleanup(); if (out_fd >= 0) close(out_fd); return fail(SA_ERR_NOMEM); }
    uint8_t* const codes = codes_mem.get();
    std::atomic<uint64_t> bad_row(UINT64_MAX);
    sa_parallel_for(n, 1 << 15, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t s = lo; s < hi; s++)
            if (sa_encode(tab.rows[s].seq, lens[s], codes + offs[s]) != SA_OK) {
                uint64_t seen = bad_row.load();
                while (s < seen && !bad_row.compare_exchange_weak(seen, s)) {}
                return;
            }
    });
    if (bad_row.load() != UINT64_MAX) {
        err = "row " + std::to_string(bad_row.load() + 2) + ": residue outside " SA_ALPHABET;
        cleanup(); if (out_fd >= 0) close(out_fd);
        return fail(SA_ERR_ALPHABET);
    }
    rep->seconds_ingest = now_s() - t0;
    bool io_ok = true;
    if (out_fd >= 0) io_ok = write_all(out_fd, st->write_csv_header, strlen(st->write_csv_header));

    if (n >= 2) {
        const double tu0 = now_s();
        const int schedule = st->schedule;
        // How many of the contexts this job can keep busy: one per 30 ms of expected kernel time (a pair-per-lane
        // kernel does about 2e12 cells a second on adjacent pairs, 8e12 on all-vs-all).  A further context costs a copy
        // of the sequences and its share of the host's turns at the output file: the 4 001 280-row data set (2.2e9
        // cells, a few milliseconds of kernel time) took 0.16 s on one context and 0.21 s on three.
        int ndev = st->num_devices;
        {
            double cells = 0;
            if (schedule == SA_SCHEDULE_ADJACENT) {
                std::vector<double> part(sa_host_threads(), 0.0);
                sa_parallel_for(n - 1, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned t) {
                    double a = 0;
                    for (uint64_t s = lo; s < hi; s++) a += (double)lens[s] * lens[s + 1];
                    part[t] = a;
                });
                for (double a : part) cells += a;
            } else {
                // sum over i < j of len[i] len[j] = ((sum len)^2 - sum len^2) / 2
                const double sum = (double)total;
                double sq = 0;
                for (uint64_t s = 0; s < n; s++) sq += (double)lens[s] * lens[s];
                cells = (sum * sum - sq) / 2;
            }
            const double est_ms = cells / (schedule == SA_SCHEDULE_ADJACENT ? 2e12 : 8e12) * 1e3;
            if (!st->force_devices) ndev = std::max(1, std::min(ndev, (int)(est_ms / 30.0)));
        }
        rep->devices_used = ndev;
        // The first context takes the sequences from the host (validation, slots, length tables; for TRIANGULAR the
        // bucket order and cursor table; for rows written on the GPU the layout and the pass-through text); the
        // others clone it device to device, all at once.
        if (preload.joinable()) preload.join();
        rc = sa_upload(ctx[0], codes, offs, lens, n);
        if (rc == SA_OK && text_mode) {
            sa_text_layout tl;
            memset(&tl, 0, sizeof(tl));
            tl.ncols = L.ncols;
            for (uint32_t k = 0; k < L.ncols && k < 32; k++) tl.kind[k] = (int8_t)L.kind[k];
            snprintf(tl.align_fmt, sizeof(tl.align_fmt), "%s", st->write_csv_align_fmt);
            // plain arrays again: nothing cleared on one thread before the threads fill it
            std::unique_ptr<uint64_t[]> ooff(new (std::nothrow) uint64_t[n + 1]);
            std::unique_ptr<char[]> obytes;
            if (ooff) {
                prefix_sums(n, ooff.get(), [&](uint64_t s) { return (uint64_t)tab.rows[s].other_len; });
                obytes.reset(new (std::nothrow) char[ooff[n] + 1]);
            }
            if (!ooff || !obytes) rc = SA_ERR_NOMEM;
            if (rc == SA_OK) {
                sa_parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned) {
                    for (uint64_t s = lo; s < hi; s++) memcpy(obytes.get() + ooff[s], tab.rows[s].other, tab.rows[s].other_len);
                });
                rc = sa_set_text_layout(ctx[0], &tl);
            }
            if (rc == SA_OK) rc = sa_upload_passthrough(ctx[0], obytes.get(), ooff.get(), n);
        }
        if (rc == SA_OK && ndev > 1) {
            rc = sa_prepare_schedule(ctx[0], schedule);
            std::vector<int> crc((size_t)ndev, SA_OK);
            std::vector<std::thread> cl;
            if (rc == SA_OK)
                for (int d = 1; d < ndev; d++)
                    cl.emplace_back([&, d]() { crc[(size_t)d] = sa_clone_sequences(ctx[(size_t)d], ctx[0]); });
            for (auto& t : cl) t.join();
            for (int d = 1; d < ndev && rc == SA_OK; d++) rc = crc[(size_t)d];
        }
        if (rc) { cleanup(); if (out_fd >= 0) close(out_fd); return fail(rc); }
        rep->seconds_upload = now_s() - tu0;
        const uint64_t rows = sa_schedule_rows(ctx[0], schedule);
        rep->pairs = sa_pair_count(ctx[0], schedule, 0, rows);
        rep->cells = sa_cell_count(ctx[0], schedule, 0, rows);
        const uint32_t stride = sa_record_stride(ctx[0]);
        const uint32_t row_bytes = sa_record_row_bytes(ctx[0]);

        // chunks of whole schedule rows; chunk k goes to device k % ndev, the text of chunk k follows that of chunk k-1
        // 262144 pairs at the default BATCH_SIZE: the best of the tuner's sweep (ADJACENT, writing on).  TRIANGULAR
        // submissions are whole rows of the schedule and find partners for the paired kernel inside a
        // submission: sixteen times as many pairs.
        uint64_t cap = st->pairs_per_submit ? st->pairs_per_submit
                                            : (uint64_t)st->batch_size * (st->schedule == SA_SCHEDULE_TRIANGULAR ? 64 : 4);
        // text submissions keep a worst-case text buffer per pair on the device: at most about 1 GiB per submission
        if (text_mode) cap = std::min<uint64_t>(cap, std::max<uint64_t>(1024, (1ull << 30) / (6ull * st->max_seq_len + 600)));
        if (cap < 1) cap = 1;
        std::vector<Chunk> chunks;
        for (uint64_t r0 = 0; r0 < rows;) {
            uint64_t r1 = r0 + 1;
            if (schedule == SA_SCHEDULE_ADJACENT) r1 = std::min(rows, r0 + cap);
            else while (r1 < rows && sa_pair_count(ctx[0], schedule, r0, r1 + 1) <= cap) r1++;
            chunks.push_back({r0, r1});
            r0 = r1;
        }
        const size_t nchunks = chunks.size();
        const int total_threads = st->format_threads > 0 ? st->format_threads : (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
        const int fthreads = std::max(1, total_threads / ndev);      // format threads of one device's drain thread
        // longest possible row: both sequences, both aligned strings (at most len1 + len2 columns each), the two
        // pass-through fields, the format parts, five numbers of at most 12 characters, separators and the newline
        const size_t row_bound = 6 * (size_t)longest_seq + 2 * (size_t)longest_other + L.part[0].size() + L.part[1].size() +
                                 L.part[2].size() + 5 * 12 + L.ncols + 1;

        // ---- where a chunk's text goes in the file: known once every earlier chunk has reported its size
        struct Order {
            std::mutex m;
            std::condition_variable cv;
            std::vector<int64_t> size;          // -1: not known yet
            std::vector<uint64_t> offset;       // offset[k] valid for k <= front
            size_t front = 0;
        } order;
        order.size.assign(nchunks, -1);
        order.offset.assign(nchunks + 1, strlen(st->write_csv_header));
        std::atomic<int> err(SA_OK);
        auto set_err = [&](int code) {
            int none = SA_OK;
            err.compare_exchange_strong(none, code);
            { std::lock_guard<std::mutex> lk(order.m); }
            order.cv.notify_all();
        };
        auto publish = [&](size_t k, uint64_t nbytes) {
            { std::lock_guard<std::mutex> lk(order.m);
              order.size[k] = (int64_t)nbytes;
              while (order.front < nchunks && order.size[order.front] >= 0) {
                  order.offset[order.front + 1] = order.offset[order.front] + (uint64_t)order.size[order.front];
                  order.front++;
              } }
            order.cv.notify_all();
        };
        auto offset_of = [&](size_t k, uint64_t& at) -> bool {     // false: the run has failed
            std::unique_lock<std::mutex> lk(order.m);
            order.cv.wait(lk, [&] { return order.front >= k || err.load() != SA_OK; });
            if (order.front < k) return false;
            at = order.offset[k];
            return true;
        };
        // One writer at a time: writes into one file take turns inside the kernel anyway, and writers of several
        // devices that wait there for each other cost more than they hide (4 001 279 rows, 500 MB, three contexts on
        // one card: 0.24 s with the writers side by side, 0.20 s one after the other).
        std::mutex file_mu;
        // threads that fill a chunk's mapping: half the host threads, eight at most (rows are formatted on the device in
        // this mode, the host has little else to do; more than eight no longer add)
        const unsigned map_threads = std::max(1u, std::min(8u, sa_host_threads() / 2));
        const bool out_in_memory = out_fd >= 0 && file_in_memory(out_fd);
        std::atomic<bool> write_failed(false);
        std::atomic<uint64_t> written(0);
        std::atomic<uint64_t> chunks_done(0);
        std::atomic<uint64_t> record_bytes(0);

        // ---- per device: one thread submits its chunks (work lists are built there), one drains them in order
        struct Dev {
            std::mutex m;
            std::condition_variable cv;
            size_t submitted = 0, consumed = 0;
            bool submit_done = false;
            double t_submit = 0, t_wait = 0, t_format = 0, t_write = 0;
        };
        std::vector<Dev> devs((size_t)ndev);
        auto submit_loop = [&](int d) {
            Dev& D = devs[(size_t)d];
            sa_ctx* c = ctx[(size_t)d];
            for (size_t k = (size_t)d; k < nchunks; k += (size_t)ndev) {
                {   // at most two chunks of a context in flight
                    std::unique_lock<std::mutex> lk(D.m);
                    D.cv.wait(lk, [&] { return D.submitted - D.consumed < 2 || err.load() != SA_OK; });
                }
                if (err.load() != SA_OK) break;
                const double ts = now_s();
                const int r = text_mode ? sa_submit_text(c, schedule, chunks[k].r0, chunks[k].r1)
                                        : sa_submit(c, schedule, chunks[k].r0, chunks[k].r1);
                D.t_submit += now_s() - ts;
                if (r) { set_err(r); break; }
                { std::lock_guard<std::mutex> lk(D.m); D.submitted++; }
                D.cv.notify_all();
            }
            { std::lock_guard<std::mutex> lk(D.m); D.submit_done = true; }
            D.cv.notify_all();
        };
        auto drain_loop = [&](int d) {
            Dev& D = devs[(size_t)d];
            sa_ctx* c = ctx[(size_t)d];
            // host formatting: two sets of buffers, one is written to the file while the other is filled
            std::vector<std::vector<char>> bufset[2] = {std::vector<std::vector<char>>((size_t)fthreads), std::vector<std::vector<char>>((size_t)fthreads)};
            std::vector<size_t> usedset[2] = {std::vector<size_t>((size_t)fthreads), std::vector<size_t>((size_t)fthreads)};
            int cur_set = 0;
            std::thread writer;            // writes the previous chunk of this device while the next one is awaited
            auto join_writer = [&]() {
                if (!writer.joinable()) return;
                const double t = now_s();
                writer.join();
                D.t_write += now_s() - t;
            };
            size_t idx = 0;
            for (size_t k = (size_t)d; k < nchunks; k += (size_t)ndev, idx++) {
                {
                    std::unique_lock<std::mutex> lk(D.m);
                    D.cv.wait(lk, [&] { return D.submitted > idx || D.submit_done || err.load() != SA_OK; });
                    if (D.submitted <= idx) break;
                }
                if (err.load() != SA_OK) break;
                const uint8_t* recs = nullptr;
                const char* text = nullptr;
                uint64_t first = 0, np = 0, nbytes = 0;
                const double tw0 = now_s();
                int r = text_mode ? sa_next_text(c, &text, &nbytes, &first, &np) : sa_next(c, &recs, &first, &np);
                D.t_wait += now_s() - tw0;
                if (r) { set_err(r); break; }
                if (text_mode) {
                    // Finished rows: the slot is free again at once (the text sits in a buffer of the context that
                    // stays valid over the next sa_next_text); a writer thread puts the slices of the text at their
                    // place in the file while the next submission is awaited.
                    if ((r = sa_release(c))) { set_err(r); break; }
                    { std::lock_guard<std::mutex> lk(D.m); D.consumed++; }
                    D.cv.notify_all();
                    publish(k, nbytes);    // first: the writers of later chunks only wait for this chunk's size
                    join_writer();         // the buffer the next-but-one text will overwrite is on disk
                    writer = std::thread([&, k, text, nbytes]() {
                        uint64_t at = 0;
                        if (!offset_of(k, at)) return;
                        std::lock_guard<std::mutex> one_writer(file_mu);
                        const bool reserved = reserve_blocks(out_fd, at, nbytes);
                        // pwrite calls side by side gain nothing: writes into one file take the inode's lock in turn
                        // (500 MB: 1 thread 0.087 s, 16 threads 0.101 s on tmpfs; 0.048 / 0.064 s on a disk file system);
                        // in a tmpfs a shared mapping of the reserved blocks is filled by several threads (write_region)
                        if (!write_region(out_fd, text, nbytes, at, reserved && out_in_memory, map_threads)) write_failed.store(true);
                        written.fetch_add(nbytes);
                    });
                    chunks_done.fetch_add(1);
                    continue;
                }
                record_bytes.fetch_add(compact ? sa_compact_bytes(c, schedule, chunks[k].r0, chunks[k].r1) : np * (uint64_t)stride);
                if (st->mode_write && np) {
                    // the previous chunk's writer works on the other buffer set: it is joined only when this chunk's
                    // rows are ready, so writing overlaps formatting
                    std::vector<std::vector<char>>& bufs = bufset[cur_set];
                    std::vector<size_t>& used = usedset[cur_set];
                    std::atomic<int> bad(0);
                    auto work = [&](int t) {
                        const uint64_t lo = np * (uint64_t)t / (uint64_t)fthreads, hi = np * (uint64_t)(t + 1) / (uint64_t)fthreads;
                        std::vector<char>& b = bufs[(size_t)t];
                        if (b.size() < (hi - lo) * row_bound + 16) b.resize((hi - lo) * row_bound + 16);
                        std::vector<char> a1(2 * (size_t)longest_seq + 2), a2(2 * (size_t)longest_seq + 2);
                        size_t pos = 0;
                        // pair (i, j) of the first record of this slice
                        uint64_t i, j;
                        if (schedule == SA_SCHEDULE_ADJACENT) { i = first + lo; j = i + 1; }
                        else {
                            i = chunks[k].r0;
                            uint64_t ix = first + lo;
                            while (sa_pair_index(c, schedule, i + 1) <= ix) i++;
                            j = i + 1 + (ix - sa_pair_index(c, schedule, i));
                        }
                        // packed records: where this slice's first record starts inside the submission
                        uint64_t at = 0;
                        if (compact) at = sa_compact_pair_offset(c, schedule, i, j) - sa_compact_pair_offset(c, schedule, chunks[k].r0, chunks[k].r0 + 1);
                        for (uint64_t q = lo; q < hi; q++) {
                            const Row& r1 = tab.rows[i];
                            const Row& r2 = tab.rows[j];
                            sa_alignment_stats s;
                            const uint32_t rec_bytes = compact ? sa_compact_record_size(c, j) : stride;
                            const uint8_t* rec = compact ? recs + at : recs + q * (size_t)stride;
                            at += rec_bytes;
                            if (sa_expand_record(rec, rec_bytes, row_bytes, r1.seq, r1.len, r2.seq, r2.len, a1.data(), a2.data(), &s) != SA_OK) {
                                bad.store(1);
                                break;
                            }
                            pos += format_row(b.data() + pos, L, r1, r2, a1.data(), a2.data(), s);
                            if (schedule == SA_SCHEDULE_ADJACENT) { i++; j++; }
                            else if (++j == n) { i++; j = i + 1; }
                        }
                        used[(size_t)t] = pos;
                    };
                    const double tf0 = now_s();
                    std::vector<std::thread> th;
                    for (int t = 1; t < fthreads; t++) th.emplace_back(work, t);
                    work(0);
                    for (auto& x : th) x.join();
                    D.t_format += now_s() - tf0;
                    if (bad.load()) { set_err(SA_ERR_RECORD); break; }
                    uint64_t total_bytes = 0;
                    for (int t = 0; t < fthreads; t++) total_bytes += used[(size_t)t];
                    publish(k, total_bytes);
                    join_writer();
                    // every format slice goes to its own place in the file
                    writer = std::thread([&, k, total_bytes, cur_set]() {
                        uint64_t chunk_off = 0;
                        if (!offset_of(k, chunk_off)) return;
                        std::lock_guard<std::mutex> one_writer(file_mu);
                        (void)reserve_blocks(out_fd, chunk_off, total_bytes);
                        std::vector<std::vector<char>>& wb = bufset[cur_set];
                        std::vector<size_t>& wu = usedset[cur_set];
                        std::vector<uint64_t> at((size_t)fthreads + 1, chunk_off);
                        for (int t = 0; t < fthreads; t++) at[(size_t)t + 1] = at[(size_t)t] + wu[(size_t)t];
                        for (int t = 0; t < fthreads; t++)
                            if (!pwrite_all(out_fd, wb[(size_t)t].data(), wu[(size_t)t], at[(size_t)t])) write_failed.store(true);
                        written.fetch_add(total_bytes);
                    });
                    cur_set ^= 1;
                } else {
                    publish(k, 0);
                }
                if ((r = sa_release(c))) { set_err(r); break; }
                { std::lock_guard<std::mutex> lk(D.m); D.consumed++; }
                D.cv.notify_all();
                chunks_done.fetch_add(1);
            }
            join_writer();
            // a drain thread that stops early must not leave its submit thread waiting for a free slot
            if (err.load() != SA_OK) { { std::lock_guard<std::mutex> lk(D.m); D.consumed = D.submitted; } D.cv.notify_all(); }
        };
        std::vector<std::thread> workers;
        for (int d = 0; d < ndev; d++) {
            workers.emplace_back(submit_loop, d);
            workers.emplace_back(drain_loop, d);
        }
        for (auto& w : workers) w.join();
        rc = err.load();
        rep->chunks_done = chunks_done.load();
        rep->record_bytes = record_bytes.load();
        for (int d = 0; d < ndev; d++) {
            const Dev& D = devs[(size_t)d];
            t_wait += D.t_wait; t_format += D.t_format; t_write += D.t_write;
            rep->device_submit[d] = D.t_submit; rep->device_wait[d] = D.t_wait;
            rep->device_format[d] = D.t_format; rep->device_write[d] = D.t_write;
        }
        rep->bytes_written += written.load();
        if (write_failed.load()) io_ok = false;
    }
    rep->bytes_written += (out_fd >= 0) ? strlen(st->write_csv_header) : 0;
    const double t1 = now_s();
    rep->seconds = t1 - t0;
    rep->seconds_gpu_wait = t_wait;
    rep->seconds_format = t_format;
    rep->seconds_write = t_write;
    if (out_fd >= 0) close(out_fd);
    cleanup();
    if (rc) return fail(rc);
    if (!io_ok) { err = "write to output file failed"; return fail(SA_ERR_ARG); }
    return SA_OK;
}

}  // extern "C"
