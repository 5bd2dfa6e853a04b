// sa_api.cpp -- host side of libsa_hip: context, sequence upload (length
// bucketing and tiling), work-list construction, launches, result streaming.
// The C ABI is declared in include/sa_hip.h.  There is no CPU fallback: every
// compute entry point needs a working HIP device and fails with a HIP error
// code otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include <time.h>

#include "../../include/sa_hip.h"
#include "sa_device.h"
#include "sa_parallel.h"

extern "C" void sa_launch_peptide(int mode, int waves_per_simd, const SaPepParams* p, unsigned grid,
                                  unsigned lds_bytes, hipStream_t stream);
extern "C" void sa_launch_pack_slots(const uint8_t* codes, const uint64_t* offs, const uint32_t* lens, uint64_t nseq,
                                     uint64_t nslots, uint32_t lslot, uint8_t* slots, uint16_t* len16, hipStream_t stream);
extern "C" void sa_launch_digest(const void* records, uint64_t bytes, unsigned long long* acc, hipStream_t stream);
extern "C" void sa_launch_peptide_pair(int ng, int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream);
extern "C" size_t sa_format_scan_bytes(uint64_t n);
extern "C" hipError_t sa_launch_format(const SaFormatParams* F, void* scan_tmp, size_t scan_bytes, hipStream_t stream);
extern "C" void sa_launch_long_pair(int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream);
extern "C" void sa_launch_long_ckpt(int ck_rows, int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream);
extern "C" void sa_launch_coop(const SaPepParams* p, unsigned grid, hipStream_t stream);
extern "C" void sa_launch_coop_wg(int waves, const SaPepParams* p, unsigned grid, hipStream_t stream);
extern "C" int sa_load_code_peptide(hipStream_t stream);
extern "C" int sa_load_code_peptide_pair(void);
extern "C" int sa_load_code_format(void);
extern "C" int sa_load_code_compact(void);
extern "C" void sa_launch_peptide_pair_longrec(int ng, int wide, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream);
extern "C" unsigned sa_coop_wg_lds_bytes(unsigned lslot, int waves);
extern "C" void sa_launch_long(int mode, const SaPepParams* p, unsigned grid, unsigned lds_bytes, hipStream_t stream);
extern "C" void sa_launch_compact(int schedule, const uint8_t* out, uint8_t* packed, const uint64_t* cp, const uint64_t* cq,
                                  uint64_t nseq, uint64_t r0, uint64_t r1, uint32_t stride, hipStream_t stream);

namespace {

// BLOSUM50 over SA_ALPHABET (the matrix of include/scoring.h:16-36)
const int8_t kBlosum50[400] = {
     5,-2,-1,-2,-1,-1,-1, 0,-2,-1,-2,-1,-1,-3,-1, 1, 0,-3,-2, 0,
    -2, 7,-1,-2,-4, 1, 0,-3, 0,-4,-3, 3,-2,-3,-3,-1,-1,-3,-1,-3,
    -1,-1, 7, 2,-2, 0, 0, 0, 1,-3,-4, 0,-2,-4,-2, 1, 0,-4,-2,-3,
    -2,-2, 2, 8,-4, 0, 2,-1,-1,-4,-4,-1,-4,-5,-1, 0,-1,-5,-3,-4,
    -1,-4,-2,-4,13,-3,-3,-3,-3,-2,-2,-3,-2,-2,-4,-1,-1,-5,-3,-1,
    -1, 1, 0, 0,-3, 7, 2,-2, 1,-3,-2, 2, 0,-4,-1, 0,-1,-1,-1,-3,
    -1, 0, 0, 2,-3, 2, 6,-3, 0,-4,-3, 1,-2,-3,-1,-1,-1,-3,-2,-3,
     0,-3, 0,-1,-3,-2,-3, 8,-2,-4,-4,-2,-3,-4,-2, 0,-2,-3,-3,-4,
    -2, 0, 1,-1,-3, 1, 0,-2,10,-4,-3, 0,-1,-1,-2,-1,-2,-3, 2,-4,
    -1,-4,-3,-4,-2,-3,-4,-4,-4, 5, 2,-3, 2, 0,-3,-3,-1,-3,-1, 4,
    -2,-3,-4,-4,-2,-2,-3,-4,-3, 2, 5,-3, 3, 1,-4,-3,-1,-2,-1, 1,
    -1, 3, 0,-1,-3, 2, 1,-2, 0,-3,-3, 6,-2,-4,-1, 0,-1,-3,-2,-3,
    -1,-2,-2,-4,-2, 0,-2,-3,-1, 2, 3,-2, 7, 0,-3,-2,-1,-1, 0, 1,
    -3,-3,-4,-5,-2,-4,-3,-4,-1, 0, 1,-4, 0, 8,-4,-3,-2, 1, 4,-1,
    -1,-3,-2,-1,-4,-1,-1,-2,-2,-3,-4,-1,-3,-4,10,-1,-1,-4,-3,-3,
     1,-1, 1, 0,-1, 0,-1, 0,-1,-3,-3, 0,-2,-3,-1, 5, 2,-4,-2,-2,
     0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 2, 5,-3,-2, 0,
    -3,-3,-4,-5,-5,-1,-3,-3,-3,-3,-2,-3,-1, 1,-4,-4,-3,15, 2,-3,
    -2,-1,-2,-3,-3,-1,-2,-3, 2,-1,-1,-2, 0, 4,-3,-2,-2, 2, 8,-1,
     0,-3,-3,-4,-1,-3,-3,-4,-4, 4, 1,-3, 1,-1,-3,-2, 0,-3,-1, 5,
};

inline int hip_code(hipError_t e) { return e == hipSuccess ? SA_OK : SA_ERR_HIP - (int)e; }

#define SA_HIP_TRY(expr)                          \
    do {                                          \
        hipError_t e_ = (expr);                   \
        if (e_ != hipSuccess) return hip_code(e_); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SA_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return hip_code(e); }
        hipError_t e = hipMalloc(&p, bytes);
        // a refused allocation is an answer, not a fault: the runtime's sticky error is taken back, or the next
        // launch's hipGetLastError would report it
        if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); return hip_code(e); }
        cap = bytes;
        return SA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SA_OK;
        if (p) { hipError_t e = hipHostFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return hip_code(e); }
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); return hip_code(e); }
        cap = bytes;
        return SA_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Work {                 // one submitted (or prepared) row range
    DevBuf items, out, counter;
    PinBuf host;
    hipEvent_t kernel_done = nullptr, copied = nullptr, t0 = nullptr, t1 = nullptr;
    uint32_t nitems = 0;
    uint32_t nshort = 0;      // the last nshort items are peptide tiles of a long-sequence context (build_items)
    int schedule = 0;
    uint64_t first_pair = 0, npairs = 0;
    bool busy = false;
    bool paired = false;      // items were built for the paired kernel
    bool coop = false;        // items are single pairs for the wave-per-pair kernel
    // text submissions: row lengths, row offsets, text, total bytes, scan scratch; pinned total and text
    DevBuf lens, offs, text, total, scan_tmp;
    PinBuf host_total;
    PinBuf items_pin;          // the work list on its way to the device (pinned: the copy needs no wait)
    PinBuf status_pin;         // the kernels' status word, copied back behind every launch
    bool text_mode = false;
    DevBuf packed;             // compact submissions: the records without their padding, as they travel to the host
    bool compact = false;
    uint64_t packed_bytes = 0;
    uint64_t r0 = 0, r1 = 0;
};

}  // namespace

struct sa_ctx {
    int device = 0;          // -1: host-only context (schedule arithmetic and work lists, no compute)
    int ncu = 0;
    int32_t S[400];
    int32_t gap = -4;
    uint32_t max_len = 64;
    hipStream_t compute = nullptr, copy = nullptr;
    // tuning
    uint32_t tiles_per_item = 0;   // 64-pair tiles per work item; 0 = from the size of the row range (see build_items)
    uint32_t wg_per_cu = 0;      // workgroups per CU in a launch; 0 = as many as are resident plus one that waits and fills the tail
    bool pairing = true;               // TRIANGULAR peptides: two first sequences per wave where the table allows it
    int64_t sprime_min = 0, sprime_max = 0;   // range of S' = S - 2*gap over the table: decides whether packed cells fit
    uint32_t bucket_width = 1;
    int coop_mode = 0;                 // wave-per-pair kernel for long sequences: 0 = when a submission has few pairs, 1 always, -1 never
    int coop_waves = 0;                // waves per pair in that kernel: 0 = from the number of pairs, else 1, 2, 4, 8 or 16
    bool compact = false;              // sa_set_compact_records: submissions hand out packed records
    std::vector<uint64_t> cp, cq;      // prefix arrays of the packed record sizes (sa_compact.hip), built with the upload
    DevBuf d_cp, d_cq;
    bool compact_on_device = false;
    uint32_t ckpt_rows = 64;           // paired long-sequence kernel in use: rows between checkpoints (32 or 64); 0 = the kernel that stores a whole direction matrix
    bool ckpt_auto = true;             // ckpt_rows follows the longest uploaded sequence (sa_upload) until sa_set_long_checkpoint_rows names a value
    uint32_t bucket_width_forced = 0;  // sa_set_bucket_width: 0 = from the number of sequences per length
    bool peptide_tiles = true;         // sa_set_peptide_tiles: in a set with long sequences, tiles of peptides alone run on the peptide kernel
    uint32_t short_buckets = 0;        // long-sequence contexts with peptides in them: the first short_buckets length buckets are the peptides'
    uint32_t short_width = 1;          // ... their width in residues (bucket_width: that of the buckets from 65 residues up)
    uint32_t short_max = 0;            // ... and the longest sequence in them
    // sequences (host mirror)
    uint64_t nseq = 0;
    std::vector<uint16_t> len;         // file order
    std::vector<uint64_t> len_suffix;  // len_suffix[i] = sum of len[i..]; built on first TRIANGULAR use (ensure_suffix)
    std::mutex tables_mu;              // the lazily built host tables (len_suffix, cp, cq): a const context may be asked from any thread
    std::vector<uint8_t> slots_host;   // file-order slots: host-only contexts, or read back from the device for the bucket order
    uint32_t min_seq = 0, max_seq = 0; // shortest / longest uploaded sequence
    uint32_t stride = 80;
    uint32_t lslot = SA_LP;            // bytes per file-order slot; > 64 selects the long-sequence kernel
    uint32_t row_bytes = 1;            // bytes per row of sequence 2 in a record
    // bucket order (built on first TRIANGULAR use)
    bool tri_ready = false;
    uint32_t nb = 0, npad = 0;
    std::vector<uint32_t> bucket_base;   // [nb+1]
    std::vector<uint32_t> fi_sorted;     // [nseq] file index of each bucket-order slot
    std::vector<uint32_t> cursor;        // [nseq][nb] entries of bucket b with file index <= p
    // device
    DevBuf d_coop;                     // scratch of the wave-per-pair kernel
    DevBuf d_codes, d_len, d_codes_t, d_fi_sorted, d_len_sorted, d_bucket_base, d_cursor, d_sprime, d_spill, d_phase, d_bcol, d_codes_s;
    bool phase_timers = false;
    // device-side text
    DevBuf d_other, d_other_off;
    SaFormatParams fmt;
    bool fmt_set = false, other_set = false;
    uint32_t max_other = 0;
    PinBuf text_ring[2];               // finished text goes to these in turn (valid until the next-but-one sa_next_text)
    int text_ring_pos = 0;
    uint32_t spill_grid = 0;
    uint64_t spill_budget = 96ull << 30;   // bytes of direction scratch a launch of the long kernels may use (see launch_work)
    Work slots[2];
    std::mutex slot_mu;                // head / tail / inflight: one submitting and one consuming thread may overlap
    int head = 0, tail = 0, inflight = 0;
    Work resident;
    bool resident_ready = false;
    const char* last_kernel = "";      // name of the fill + traceback kernel of the most recent launch
    std::string last_kernel_both;      // ... when there were two (long sequences + peptide tiles): "a + b"
    uint32_t last_grid = 0, last_grid_wanted = 0;    // its workgroups, and how many it would have had with memory to spare
};

namespace {

uint64_t tri_row_start(uint64_t n, uint64_t i) { return i * (2 * n - i - 1) / 2; }

// The paired kernels keep 16*D + 15 below 0x7c00 (1983 for D) in 16 bits and take a carry back through their
// table entries (S' >= 1, S' = S - 2*gap).  Peptides: D <= longest sequence * max S'.  Long sequences: D minus a
// per-lane base that follows the strip's border column and is moved every 16 rows, at most (17 + 64) * max S'.
void table_range(const int32_t* S, int32_t gap, int64_t& low, int64_t& top) {
    low = INT64_MAX; top = INT64_MIN;
    for (int k = 0; k < 400; k++) {
        const int64_t sp = (int64_t)S[k] - 2 * (int64_t)gap;
        low = std::min(low, sp); top = std::max(top, sp);
    }
}
// Packed 16-bit cells hold 16 (D - base) + two tag fields.  With the three-way maximum as one v_pk_maximum3_f16 the
// patterns must stay below 0x7c00: D - base <= 1983 ("narrow").  With two v_pk_max_u16 per maximum (one instruction
// more per column) the whole 16-bit range is usable: D - base <= 4095 ("wide").
// Peptides: base 0, D grows by at most max S' per row and column, so longest sequence x max S' must fit.  Long
// sequences: a strip is 64 columns wide and the lanes' bases move up to the row above every R rows, so
// (64 + R + 1) x max S' must fit; R = 16 where the table allows it, else 8 or 4 -- a move of the bases is 64
// subtractions per lane, 2.6 % of the tagged rows' instructions at R = 8.  Narrow: max S' <= 24 (BLOSUM50 at gap -4:
// 23), 27, 28; wide: <= 50, 56, 59.  0: the table does not fit.  Same table, same box, 2 000 proteins: narrow 68.0 ms,
// wide 76.4 ms (checkpointing kernel), 88.7 / 95.9 ms (direction rows); the 32-bit kernel takes 146 ms.
uint32_t long_rebase_rows(const sa_ctx* c, int64_t limit) {
    if (c->sprime_min < 1) return 0;
    for (uint32_t rows : {16u, 8u, 4u})
        if ((int64_t)(64 + rows + 1) * c->sprime_max <= limit) return rows;
    return 0;
}
enum { kCellsNone = 0, kCellsNarrow = 1, kCellsWide = 2 };
int packed_cells_mode(const sa_ctx* c, bool lng) {
    if (c->sprime_min < 1) return kCellsNone;
    if (lng) return long_rebase_rows(c, 1983) ? kCellsNarrow : long_rebase_rows(c, 4095) ? kCellsWide : kCellsNone;
    // (the peptide tiles of a set with long sequences: the longest of its peptides)
    const int64_t span = std::max<int64_t>(1, c->lslot > SA_LP ? c->short_max : std::min<int64_t>(c->max_seq, SA_LP));
    return span * c->sprime_max <= 1983 ? kCellsNarrow : span * c->sprime_max <= 4095 ? kCellsWide : kCellsNone;
}
bool packed_cells_fit(const sa_ctx* c, bool lng) { return packed_cells_mode(c, lng) != kCellsNone; }

// Tables with one 8-byte entry per sequence are built when something first asks for them, under a lock (queries take
// a const context and may come from any thread): the 4 001 280-row ADJACENT run with rows written on the device needs
// none of them, and their 96 MB of fresh pages were 20 ms of its upload.
// len_suffix[i] = sum of len[i ..]
void ensure_suffix(const sa_ctx* cc) {
    sa_ctx* c = const_cast<sa_ctx*>(cc);
    std::lock_guard<std::mutex> lock(c->tables_mu);
    const uint64_t n = c->nseq;
    if (c->len_suffix.size() == n + 1) return;
    c->len_suffix.resize(n + 1);
    uint64_t* const v = c->len_suffix.data();
    prefix_sums(n, v, [&](uint64_t s) { return (uint64_t)c->len[s]; });
    const uint64_t total = v[n];
    sa_parallel_for(n + 1, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t s = lo; s < hi; s++) v[s] = total - v[s];
    });
}

// packed record of a pair whose second sequence has `len` residues: header + one code per row, rounded up to 4 bytes
inline uint32_t packed_record_bytes(const sa_ctx* c, uint32_t len) {
    const uint32_t header = c->row_bytes == 1 ? SA_REC_HEADER_BYTES : SA_LONG_HEADER_BYTES;
    return (header + c->row_bytes * len + 3u) & ~3u;
}
// cp[s] = packed bytes of the records whose second sequence is one of 0 .. s-1; cq[i] = packed bytes of the
// TRIANGULAR rows of first sequences 0 .. i-1 (row i holds second sequences i+1 .. n-1)
void ensure_compact_tables(const sa_ctx* cc) {
    sa_ctx* c = const_cast<sa_ctx*>(cc);
    std::lock_guard<std::mutex> lock(c->tables_mu);
    const uint64_t n = c->nseq;
    if (c->cp.size() == n + 2 && c->cq.size() == n + 1) return;
    c->cp.resize(n + 2);
    prefix_sums(n, c->cp.data(), [&](uint64_t s) { return (uint64_t)packed_record_bytes(c, c->len[s]); });
    c->cp[n + 1] = c->cp[n];
    c->cq.resize(n + 1);
    const uint64_t all = c->cp[n];
    prefix_sums(n, c->cq.data(), [&](uint64_t i) { return all - c->cp[i + 1]; });
    c->compact_on_device = false;
}
int ensure_compact_on_device(sa_ctx* c) {
    if (c->compact_on_device) return SA_OK;
    ensure_compact_tables(c);
    int rc;
    if ((rc = c->d_cp.reserve(c->cp.size() * 8)) || (rc = c->d_cq.reserve(c->cq.size() * 8))) return rc;
    SA_HIP_TRY(hipMemcpy(c->d_cp.p, c->cp.data(), c->cp.size() * 8, hipMemcpyHostToDevice));
    SA_HIP_TRY(hipMemcpy(c->d_cq.p, c->cq.data(), c->cq.size() * 8, hipMemcpyHostToDevice));
    c->compact_on_device = true;
    return SA_OK;
}
// byte offset of pair (i, j) in the packed stream of the whole schedule
inline uint64_t compact_offset(const sa_ctx* c, int schedule, uint64_t i, uint64_t j) {
    if (schedule == SA_SCHEDULE_ADJACENT) return c->cp[j] - c->cp[1];          // pair (i, i + 1): second sequences 1 .. i come before
    return c->cq[i] + c->cp[j] - c->cp[i + 1];
}

// length bucket of a sequence (see ensure_triangular)
uint32_t bucket_of_length(const sa_ctx* c, uint32_t len) {
    if (c->short_buckets == 0) return (len - c->min_seq) / c->bucket_width;
    if (len <= SA_LP) return (len - c->min_seq) / c->short_width;
    return c->short_buckets + (len - (SA_LP + 1)) / c->bucket_width;
}

int ensure_triangular(sa_ctx* c) {
    if (c->tri_ready) return SA_OK;
    const uint64_t n = c->nseq;
    ensure_suffix(c);
    if (c->slots_host.empty()) {
        // the residue codes live on the device only (the upload keeps no second host copy: the 4 M-row
        // ADJACENT run never needs one); the bucket order is built from the slots read back
        c->slots_host.resize((size_t)(n + 2) * c->lslot);
        SA_HIP_TRY(hipMemcpy(c->slots_host.data(), c->d_codes.p, c->slots_host.size(), hipMemcpyDeviceToHost));
    }
    // length buckets: exact lengths when every length has plenty of sequences, wider otherwise
    // long sequences: about 48 sequences per bucket, at least 8 residues wide (2 000 proteins of 300-1000 residues: 97.2 ms
    // at 16 residues against 100.8 at 64; 10 000: flat between 8 and 18) -- a wave's rows run to its longest second
    // sequence, and a wave takes its 64 lanes across bucket borders anyway
    auto peptide_width = [&](uint32_t range, uint64_t count) {
        return std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>((256ull * range + count - 1) / count, 64u));
    };
    auto long_width = [&](uint32_t range, uint64_t count) {
        return std::max<uint32_t>(8u, (uint32_t)std::min<uint64_t>((48ull * range + count - 1) / count, 64u));
    };
    auto forced = [&](uint32_t width, uint32_t range) {
        return c->bucket_width_forced ? std::max(1u, std::min(c->bucket_width_forced, std::max(range, 1u))) : width;
    };
    // A set with sequences on both sides of 64 residues has two runs of buckets: the peptides' (widths as in a peptide
    // context) and, from 65 residues up, the long sequences'.  No bucket mixes the two kinds (see build_items).
    uint64_t npep = 0;
    if (c->lslot > SA_LP && c->min_seq <= SA_LP)
        for (uint64_t s = 0; s < n; s++) npep += c->len[s] <= SA_LP;
    uint32_t width, nb;
    if (npep) {
        const uint32_t range1 = SA_LP - c->min_seq + 1, range2 = c->max_seq - SA_LP;
        c->short_width = forced(peptide_width(range1, npep), range1);
        c->short_buckets = (range1 + c->short_width - 1) / c->short_width;
        width = forced(long_width(range2, n - npep), range2);
        nb = c->short_buckets + (range2 + width - 1) / width;
    } else {
        const uint32_t range = c->max_seq - c->min_seq + 1;
        width = forced(c->lslot > SA_LP ? long_width(range, n) : peptide_width(range, n), range);
        c->short_width = width;
        c->short_buckets = 0;
        nb = (range + width - 1) / width;
    }
    if ((uint64_t)nb * n > (1ull << 28)) return SA_ERR_NOMEM;   // cursor table of more than 1 GiB
    c->nb = nb;
    c->bucket_width = width;
    std::vector<uint32_t> count(nb + 1, 0);
    std::vector<uint16_t> bucket_of(n);
    for (uint64_t s = 0; s < n; s++) {
        bucket_of[s] = (uint16_t)bucket_of_length(c, c->len[s]);
        count[bucket_of[s] + 1]++;
    }
    c->bucket_base.assign(nb + 1, 0);
    for (uint32_t b = 0; b < nb; b++) c->bucket_base[b + 1] = c->bucket_base[b] + count[b + 1];
    c->npad = (uint32_t)((n + 63) / 64 * 64 + 64);
    c->fi_sorted.assign(n, 0);
    std::vector<uint16_t> len_sorted(c->npad, 0);
    std::vector<uint8_t> codes_t((size_t)c->lslot * c->npad, SA_PAD_CODE);
    const bool lng = c->lslot > SA_LP;
    // Long-sequence contexts: the buckets whose sequences all have at most 64 residues ("peptide buckets") keep the
    // piece-major copy as well, for the tiles that run on the peptide kernel (build_items).
    c->short_max = 0;
    const bool pieces = !lng || c->short_buckets > 0;
    std::vector<uint8_t> codes_s(pieces ? (size_t)c->npad * SA_LP : 16, SA_PAD_CODE);
    std::vector<uint32_t> fill(c->bucket_base.begin(), c->bucket_base.end() - 1);
    c->cursor.assign((size_t)n * nb, 0);
    std::vector<uint32_t> running(nb, 0);
    for (uint64_t s = 0; s < n; s++) {          // ascending file index: every bucket ends up ordered by it
        const uint32_t b = bucket_of[s];
        const uint32_t slot = fill[b]++;
        c->fi_sorted[slot] = (uint32_t)s;
        len_sorted[slot] = c->len[s];
        const uint8_t* src = c->slots_host.data() + (size_t)s * c->lslot;
        for (uint32_t i = 0; i < c->len[s]; i++) codes_t[(size_t)i * c->npad + slot] = src[i];
        if (pieces && c->len[s] <= SA_LP) {      // 16-byte pieces, piece-major: piece j of bucket-order slot q at (j * npad + q) * 16
            for (uint32_t i = 0; i < c->len[s]; i += 16)
                memcpy(&codes_s[((size_t)(i >> 4) * c->npad + slot) * 16], src + i, std::min<uint32_t>(16, c->len[s] - i));
            if (b < c->short_buckets) c->short_max = std::max<uint32_t>(c->short_max, c->len[s]);
        }
        running[b]++;
        memcpy(&c->cursor[(size_t)s * nb], running.data(), nb * sizeof(uint32_t));
    }
    if (c->device != SA_DEVICE_NONE) {
        int rc;
        std::vector<uint32_t> fi_pad(c->npad, SA_INVALID_FI);
        memcpy(fi_pad.data(), c->fi_sorted.data(), n * 4);
        if ((rc = c->d_codes_t.reserve(codes_t.size()))) return rc;
        if ((rc = c->d_codes_s.reserve(codes_s.size()))) return rc;
        SA_HIP_TRY(hipMemcpy(c->d_codes_s.p, codes_s.data(), codes_s.size(), hipMemcpyHostToDevice));
        if ((rc = c->d_fi_sorted.reserve(fi_pad.size() * 4))) return rc;
        if ((rc = c->d_len_sorted.reserve(len_sorted.size() * 2))) return rc;
        if ((rc = c->d_bucket_base.reserve(c->bucket_base.size() * 4))) return rc;
        if ((rc = c->d_cursor.reserve(c->cursor.size() * 4))) return rc;
        SA_HIP_TRY(hipMemcpy(c->d_codes_t.p, codes_t.data(), codes_t.size(), hipMemcpyHostToDevice));
        SA_HIP_TRY(hipMemcpy(c->d_fi_sorted.p, fi_pad.data(), fi_pad.size() * 4, hipMemcpyHostToDevice));
        SA_HIP_TRY(hipMemcpy(c->d_len_sorted.p, len_sorted.data(), len_sorted.size() * 2, hipMemcpyHostToDevice));
        SA_HIP_TRY(hipMemcpy(c->d_bucket_base.p, c->bucket_base.data(), c->bucket_base.size() * 4, hipMemcpyHostToDevice));
        SA_HIP_TRY(hipMemcpy(c->d_cursor.p, c->cursor.data(), c->cursor.size() * 4, hipMemcpyHostToDevice));
    }
    c->tri_ready = true;
    return SA_OK;
}

bool is_long(const sa_ctx* c) { return c->lslot > SA_LP; }

// TRIANGULAR work of a peptide context runs on the paired kernel (two first sequences per wave)
bool uses_pairs(const sa_ctx* c) { return c->pairing && packed_cells_fit(c, is_long(c)); }

// A long-sequence context whose shortest buckets hold peptides only hands their tiles to the paired peptide kernel
// (in its long-record form), if the table fits that kernel's cells; sa_set_peptide_tiles(ctx, 0) keeps every tile on
// the long-sequence kernel.
bool peptide_tiles_possible(const sa_ctx* c) {
    return is_long(c) && c->peptide_tiles && c->pairing && c->short_buckets > 0 && packed_cells_fit(c, false);
}

// paired (in/out): the caller allows the paired kernel / the items were built for it.  A row range in which
// fewer than half of the first sequences find a partner of their length bucket runs unpaired: a wave of
// the paired kernel with one first sequence does half the work for the same time.
uint32_t grid_for(const sa_ctx* c);

// nshort (out, optional): how many items at the END of the list are tiles of peptides alone inside a long-sequence
// context (peptide_tiles_possible); they run on the peptide kernel.
bool peptide_tiles_possible(const sa_ctx* c);
int build_items(sa_ctx* c, int schedule, uint64_t r0, uint64_t r1, std::vector<SaItem>& items, bool& paired,
                uint32_t* nshort = nullptr) {
    items.clear();
    if (nshort) *nshort = 0;
    SaItem it;
    memset(&it, 0, sizeof(it));
    if (schedule == SA_SCHEDULE_ADJACENT) {
        paired = false;
        for (uint64_t k = r0; k < r1; k += 64) {
            it.p = (uint32_t)k;
            it.count = (uint32_t)std::min<uint64_t>(64, r1 - k);
            items.push_back(it);
        }
        return SA_OK;
    }
    int rc = ensure_triangular(c);
    if (rc) return rc;
    // long sequences: one 64-pair tile is already milliseconds of work, and few items per workgroup leave a long tail
    // peptides: many tiles per item spread the cost of an item's query profile (on 10^7 tiles: 4 tiles 111 ms, 8 108,
    // 16 105.1, 32 104.7, 64 105.2, 128 107.4; tools/knob_sweep.py), few keep the tail short when the range is small:
    // at least 32 items per workgroup, at most 32 tiles
    uint32_t per = 1;
    if (!is_long(c)) {
        per = c->tiles_per_item;
        if (per == 0) {
            const uint64_t tiles = sa_pair_count(c, schedule, r0, r1) / (uses_pairs(c) && paired ? 128 : 64);
            const uint64_t want = tiles / (std::max<uint64_t>(1, grid_for(c)) * 32);
            per = 1;
            while (per < 32 && per * 2 <= want) per *= 2;
        }
    }
    const uint32_t nb = c->nb;
    // A set with long sequences and peptides: the tiles whose three sequences (the one or two first sequences and every
    // lane's second sequence) have at most 64 residues are kept apart.  A first sequence's valid list runs through the
    // buckets from the shortest up, so those tiles are the whole tiles at its start that end inside the peptide buckets.
    struct ShortRun { uint32_t p, p2, tiles; };
    std::vector<ShortRun> short_runs;
    const bool split = is_long(c) && peptide_tiles_possible(c);
    uint64_t short_tiles = 0;
    // items of first sequence p (and, in the paired kernel, of a second one p2 > p that rides along)
    auto emit = [&](uint64_t p, uint32_t p2) {
        const uint32_t nvalid = (uint32_t)(c->nseq - 1 - p);
        const uint32_t* cur = &c->cursor[(size_t)p * nb];
        uint32_t b = 0, cum = 0;
        uint32_t first_v0 = 0;
        if (split && c->len[p] <= SA_LP && (p2 == SA_INVALID_FI || c->len[p2] <= SA_LP)) {
            uint32_t in_short = 0;
            for (uint32_t k = 0; k < c->short_buckets; k++) in_short += c->bucket_base[k + 1] - c->bucket_base[k] - cur[k];
            const uint32_t tiles = in_short == nvalid ? (nvalid + 63) / 64 : in_short / 64;
            if (tiles) {
                short_runs.push_back({(uint32_t)p, p2, tiles});
                short_tiles += tiles;
                first_v0 = tiles * 64;
            }
        }
        for (uint32_t v0 = first_v0; v0 < nvalid; v0 += per * 64) {
            // bucket that holds entry v0 of p's valid list
            for (;;) {
                const uint32_t cnt = c->bucket_base[b + 1] - c->bucket_base[b] - cur[b];
                if (cum + cnt > v0) break;
                cum += cnt;
                b++;
            }
            it.p = (uint32_t)p; it.v0 = v0;
            it.count = std::min(per, (nvalid - v0 + 63) / 64);
            it.b0 = b; it.c0 = cum;
            it.reserved[0] = p2;
            items.push_back(it);
        }
    };
    std::vector<std::pair<uint32_t, uint32_t>> order;
    if (paired && uses_pairs(c)) {
        // first sequences of the same length bucket go two by two, in file order; the valid list of the
        // earlier one contains the later one's
        order.reserve(r1 - r0);
        for (uint64_t p = r0; p < r1; p++) order.push_back({bucket_of_length(c, c->len[p]), (uint32_t)p});
        std::sort(order.begin(), order.end());
        size_t partners = 0;
        for (size_t k = 0; k + 1 < order.size();) {
            if (order[k + 1].first == order[k].first) { partners += 2; k += 2; } else k += 1;
        }
        paired = 2 * partners >= order.size();
    } else {
        paired = false;
    }
    // the peptide tiles, behind the other items: several tiles per item as in a peptide context (one query profile
    // per item), in the order of their first sequences
    auto emit_short = [&]() {
        if (short_runs.empty()) return;
        const uint64_t want = short_tiles / (std::max<uint64_t>(1, (uint64_t)c->ncu * 16) * 32);
        uint32_t per_short = 1;
        while (per_short < 32 && per_short * 2 <= want) per_short *= 2;
        const size_t before = items.size();
        for (const ShortRun& r : short_runs) {
            const uint32_t* cur = &c->cursor[(size_t)r.p * nb];
            uint32_t b = 0, cum = 0;
            for (uint32_t t = 0; t < r.tiles; t += per_short) {
                const uint32_t v0 = t * 64;
                for (;;) {
                    const uint32_t cnt = c->bucket_base[b + 1] - c->bucket_base[b] - cur[b];
                    if (cum + cnt > v0) break;
                    cum += cnt;
                    b++;
                }
                it.p = r.p; it.v0 = v0;
                it.count = std::min(per_short, r.tiles - t);
                it.b0 = b; it.c0 = cum;
                it.reserved[0] = r.p2;
                items.push_back(it);
            }
        }
        if (nshort) *nshort = (uint32_t)(items.size() - before);
    };
    if (!paired) {
        for (uint64_t p = r0; p < r1; p++) emit(p, SA_INVALID_FI);
        // long sequences: items differ a lot in work (first sequences of 300 and of 1000 residues); the
        // longest go first so that the tail of the launch is made of short ones
        if (is_long(c))
            std::stable_sort(items.begin(), items.end(),
                             [c](const SaItem& a, const SaItem& b) { return c->len[a.p] > c->len[b.p]; });
        emit_short();
        return SA_OK;
    }
    // the partners are found in ascending bucket order; the items go out longest bucket first, so that the
    // tail of the launch is made of short first sequences
    std::vector<std::pair<uint32_t, uint32_t>> duo;      // (p1, p2 or none), ascending buckets
    for (size_t k = 0; k < order.size();) {
        if (k + 1 < order.size() && order[k + 1].first == order[k].first) {
            duo.push_back({order[k].second, order[k + 1].second});
            k += 2;
        } else {
            duo.push_back({order[k].second, SA_INVALID_FI});
            k += 1;
        }
    }
    for (size_t k = duo.size(); k-- > 0;) emit(duo[k].first, duo[k].second);
    emit_short();
    return SA_OK;
}

// long sequences: direction rows per strip, sized by the longest sequence uploaded (rows are walked eight at a time)
uint32_t long_dir_rows(const sa_ctx* c) { return (c->max_seq + 7u) & ~7u; }

// The wave-per-pair kernel (sa_kernels_coop.hip) takes a submission of long sequences when its pairs would fill only
// a fraction of the chip's lanes on the pair-per-lane kernels.  Measured on proteins of 300-1000 residues, all-vs-all
// (tools/coop_crossover.py): 16 471 pairs 3.4 ms against 21.4 ms, 65 341 pairs 12.6 against 22.1, 130 816 pairs 24.3
// against 22.6 -- the pair-per-lane kernels take about 22 ms whatever the count until the chip is full, this one
// 0.19 us per pair.  Up to 384 pairs per CU go here.
// Forced (sa_set_cooperative(ctx, 1)) it still takes at most 4096 pairs per CU: its work list is one 32-byte item
// per pair, built and sorted on the host, and a whole TRIANGULAR schedule of 10 000 proteins would be 1.6 GB of it;
// larger submissions run on the pair-per-lane kernels whatever the mode.
bool use_coop(const sa_ctx* c, uint64_t npairs) {
    if (c->lslot <= SA_LP || c->coop_mode < 0 || npairs == 0) return false;
    if (c->coop_mode > 0) return npairs <= 4096ull * (uint64_t)c->ncu;
    return npairs <= 384ull * (uint64_t)c->ncu;
}
// Waves per pair: with fewer pairs than wave slots (16 per CU) a pair gets a workgroup of several waves, which share
// its strips (sa_coop_wg_kernel): the largest power of two with which (nearly) every pair still has a workgroup in
// flight.  Measured (tools/coop_waves_sweep.py, ms at 1 / 2 / 4 / 8 / 16 waves): 64 pairs of 2000-4096 residues 9.9 /
// 5.9 / 3.4 / 2.8 / 2.7, 512 such pairs 10.4 / 7.9 / 3.6 / 3.6 / 3.9, 1024: 10.6 / 8.0 / 5.3 / 5.5 / 7.4; proteins of
// 300-1000 residues, 496 pairs 0.76 / 0.55 / 0.40 / 0.37 / 0.54, 1035 pairs 0.82 / 0.63 / 0.47 / 0.64 / 1.09, 4095 pairs
// 1.25 / 1.15 / 1.33 / 2.19 / 4.01.
// A workgroup's LDS (row residues, whole columns, rings, every lane's table row) grows with the waves and the sequence
// length; 156 KiB of the CU's 160 is the most one may take.
constexpr uint32_t kCoopLdsMax = 156u * 1024u;
int coop_waves_for(const sa_ctx* c, uint64_t npairs) {
    int w = 1;
    if (c->coop_waves) w = c->coop_waves;
    else while (w < 16 && npairs * (uint64_t)(2 * w) <= 20ull * (uint64_t)c->ncu) w *= 2;
    while (w > 1 && sa_coop_wg_lds_bytes(c->lslot, w) > kCoopLdsMax) w /= 2;
    return w;
}
// its scratch per workgroup, KiB: direction blocks (strips x 16-step blocks x 256 B), two border columns, row notes
uint32_t coop_kib_per_wg(const sa_ctx* c) {
    const uint64_t steps_cap = c->lslot + 128;
    const uint64_t bytes = (uint64_t)(c->lslot / 64) * (steps_cap / 16) * 256 + 2 * steps_cap * 4 + 2 * (c->lslot + 64) * 2;
    return (uint32_t)((bytes + 1023) / 1024);
}

// Which packed long-sequence kernel, left to the library: the checkpointing kernel fills again a 64-row tile per strip
// and path, which for short sequences is most of the matrix; keeping two direction bits per cell is then cheaper.
// All-vs-all sets, TCUPS checkpointing / direction rows: lengths 65-128 4.5 / 5.7, 100-300 6.8 / 7.6, 200-500 9.4 / 9.9,
// 600-1400 16.1 / 10.5, 1500-3000 13.7 / 4.8 (tools/midlen_time.py).
uint32_t long_auto_checkpoint_rows(uint32_t longest) { return longest >= 512 ? 64u : 0u; }

// rows of the checkpointing kernel's border columns and checkpoints: the longest uploaded sequence, rounded up to 64
uint32_t long_ckpt_rows_cap(const sa_ctx* c) { return (c->max_seq + 63u) & ~63u; }

// global scratch of one workgroup: direction rows (and, for long sequences, the strip border columns)
size_t spill_bytes_per_wg(const sa_ctx* c) {
    if (!is_long(c)) return (size_t)2 * SA_LP * 1024;      // two sets of direction rows for the paired kernel
    const size_t strips = c->lslot / 64;
    const size_t one_matrix = strips * long_dir_rows(c) * 1024;      // strips x rows x 1 KiB: the 32-bit kernel's direction rows
    if (c->ckpt_rows == 0 || !uses_pairs(c)) return 2 * one_matrix; // one per half of the paired kernel
    // checkpointing kernel: tile tags (2 x CK KiB), border columns of every strip (256 B per row), every CK-th row of
    // every strip (16 KiB each); row ranges that run unpaired still need the 32-bit kernel's direction rows
    const size_t rows = long_ckpt_rows_cap(c);
    const size_t ck = (size_t)2 * 64 * 1024 + strips * rows * 256 + (rows / c->ckpt_rows) * strips * 16384;     // tiles are 64 rows high
    return std::max(ck, one_matrix);
}

// 4-column register groups of the paired peptide kernel's row: 13 (52 columns) when no sequence is longer
int pair_groups(const sa_ctx* c) { return c->max_seq <= 52 ? 13 : 16; }

uint32_t grid_for(const sa_ctx* c) {
    if (is_long(c)) {
        // 32-bit cells: two waves per SIMD (a third one spills registers inside the row loop: measured 1.8 vs
        // 3.0 TCUPS).  Packed cells keep one register row and run four, as far as the LDS copies of the two first
        // sequences allow.  The direction area of a workgroup is large, keep the total under 96 GiB of the 288.
        const uint64_t budget = c->spill_budget;
        uint64_t per_cu = 8;
        if (uses_pairs(c)) per_cu = std::min<uint64_t>(16, (160u * 1024u) / (SA_LDS_PROF_BYTES + 64 + 2 * c->lslot));
        if (c->wg_per_cu) per_cu = std::min<uint64_t>(per_cu, c->wg_per_cu);      // sa_set_tuning: fewer, for measurements
        uint64_t g = std::min<uint64_t>((uint64_t)c->ncu * per_cu, budget / spill_bytes_per_wg(c));
        return (uint32_t)std::max<uint64_t>(1, g);
    }
    // resident workgroups (= waves) per CU: by registers 4 per SIMD for the paired kernel, 3 otherwise; by LDS what
    // fits into 160 KiB
    const int ng = pair_groups(c);
    const uint32_t lds = uses_pairs(c) ? SA_PAIR_LDS_BYTES(ng) : SA_LDS_FIXED_BYTES;
    const uint32_t resident = std::min<uint32_t>(uses_pairs(c) ? 16 : 12, (160u * 1024u) / lds);
    const uint32_t per_cu = c->wg_per_cu ? c->wg_per_cu : resident + 1;
    return (uint32_t)c->ncu * std::max<uint32_t>(1, per_cu);
}

int ensure_spill(sa_ctx* c, uint32_t grid) {
    // sized in bytes for this launch's grid and the current kernel choice (pairing and the checkpoint setting change
    // what a workgroup needs); the buffers only ever grow
    int rc = c->d_spill.reserve((size_t)grid * spill_bytes_per_wg(c));
    if (rc) return rc;
    if (is_long(c) && (rc = c->d_bcol.reserve((size_t)grid * 4 * c->lslot * 256))) return rc;
    c->spill_grid = std::max(c->spill_grid, grid);
    return SA_OK;
}

// dword of the 256-byte queue buffer where the second launch of a submission (peptide tiles) has its head; its status
// word follows at the same distance as the first one's
constexpr uint32_t kShortQueueWord = 32;

// queue the kernels of one row range on the compute stream
int launch_work(sa_ctx* c, Work& w, bool timed) {
    if (w.npairs == 0 || w.nitems == 0) return SA_OK;
    if (is_long(c) && c->spill_grid == 0 && !w.coop) {
        // First launch after an upload: the records of this row range are allocated by now, so what is free tells how
        // many workgroups' direction areas fit.  A full grid (16 workgroups per CU) of 1000-residue proteins takes
        // 125 GiB; with less the launch runs fewer waves per SIMD.  16 GiB stay free for the second slot's records.
        size_t free_b = 0, total_b = 0;
        SA_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        free_b += c->d_spill.cap + c->d_bcol.cap;      // what an earlier layout of the scratch areas holds is reused
        // Three quarters of what is free at most (other contexts may share the card) and never the last 16 GiB.
        const uint64_t margin = 16ull << 30;
        const uint64_t room = free_b > margin ? free_b - margin : 0;
        c->spill_budget = std::max<uint64_t>(std::min<uint64_t>(room, (uint64_t)free_b / 4 * 3), 1ull << 30);
    }
    const uint32_t grid = grid_for(c);
    c->last_grid = grid;
    {
        // the same count without the limit that the scratch areas' memory puts on long-sequence launches
        const uint64_t budget = c->spill_budget;
        c->spill_budget = ~0ull;
        c->last_grid_wanted = grid_for(c);
        c->spill_budget = budget;
    }
    int rc = w.coop ? SA_OK : ensure_spill(c, grid);
    if (rc) return rc;
    // peptide tiles of a long-sequence context: their kernel's grid and LDS, and its share of the scratch area (the two
    // launches follow each other on one stream and use the same area)
    const uint32_t nlong = w.nitems - w.nshort;
    const int short_ng = c->short_max <= 52 ? 13 : 16;
    const uint32_t short_lds = SA_PAIR_LDS_BYTES(short_ng);
    const uint32_t short_grid = (uint32_t)c->ncu * (std::min<uint32_t>(16, (160u * 1024u) / short_lds) + 1);
    if (w.nshort && (rc = c->d_spill.reserve((size_t)short_grid * 2 * SA_LP * 1024))) return rc;
    SA_HIP_TRY(hipMemsetAsync(w.counter.p, 0, 256, c->compute));
    SaPepParams P;
    memset(&P, 0, sizeof(P));
    P.codes_fo = (const uint8_t*)c->d_codes.p;
    P.len_fo = (const uint16_t*)c->d_len.p;
    P.codes_t = (const uint8_t*)c->d_codes_t.p;
    P.codes_s = (const uint8_t*)c->d_codes_s.p;
    P.fi_sorted = (const uint32_t*)c->d_fi_sorted.p;
    P.len_sorted = (const uint16_t*)c->d_len_sorted.p;
    P.bucket_base = (const uint32_t*)c->d_bucket_base.p;
    P.cursor = (const uint32_t*)c->d_cursor.p;
    P.nb = c->nb;
    P.npad = c->npad;
    P.sprime = (const int32_t*)c->d_sprime.p;
    P.items = (const SaItem*)w.items.p;
    P.nitems = nlong;
    P.counter = (uint32_t*)w.counter.p;
    P.out = (uint8_t*)w.out.p;
    P.stride = c->stride;
    P.out_base_pair = w.first_pair;
    P.spill = (uint32_t*)c->d_spill.p;
    P.spill_rows = is_long(c) ? long_dir_rows(c) : 2 * SA_LP;
    const bool ckpt = is_long(c) && !w.coop && w.schedule == SA_SCHEDULE_TRIANGULAR && w.paired && c->ckpt_rows != 0;
    if (ckpt) P.spill_rows = long_ckpt_rows_cap(c);
    P.nseq = (uint32_t)c->nseq;
    P.gap = c->gap;
    if (c->phase_timers) {
        if ((rc = c->d_phase.reserve((size_t)grid * 64))) return rc;
        SA_HIP_TRY(hipMemsetAsync(c->d_phase.p, 0, (size_t)grid * 64, c->compute));
        P.phase_cycles = (unsigned long long*)c->d_phase.p;
    }
    if (timed) SA_HIP_TRY(hipEventRecord(w.t0, c->compute));
    P.lslot = c->lslot;
    P.bcol = (int32_t*)c->d_bcol.p;
    const bool wide = packed_cells_mode(c, is_long(c)) == kCellsWide;
    P.rebase_mask = std::max<uint32_t>(long_rebase_rows(c, wide ? 4095 : 1983), 4) - 1;
    const bool tri = w.schedule == SA_SCHEDULE_TRIANGULAR;
    if (nlong == 0) {
        c->last_kernel = "";
    } else if (w.coop) {
        const int waves = coop_waves_for(c, w.nitems);
        // workgroups in flight per CU: by wave slots, and for several waves per pair by LDS
        uint64_t wg_per_cu = 16 / waves;
        if (waves > 1) wg_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(wg_per_cu, (160u * 1024u) / (sa_coop_wg_lds_bytes(c->lslot, waves) + 4096u)));
        const uint32_t cgrid = (uint32_t)std::min<uint64_t>(w.nitems, (uint64_t)c->ncu * wg_per_cu);
        c->last_grid = c->last_grid_wanted = cgrid;
        const uint32_t kib = coop_kib_per_wg(c);
        if ((rc = c->d_coop.reserve((size_t)cgrid * kib * 1024))) return rc;
        P.spill = (uint32_t*)c->d_coop.p;
        P.spill_rows = kib;
        if (waves == 1) {
            sa_launch_coop(&P, cgrid, c->compute);
            c->last_kernel = "sa_coop_kernel";
        } else {
            sa_launch_coop_wg(waves, &P, cgrid, c->compute);
            c->last_kernel = waves == 2 ? "sa_coop_wg_kernel<2>" : waves == 4 ? "sa_coop_wg_kernel<4>"
                           : waves == 8 ? "sa_coop_wg_kernel<8>" : "sa_coop_wg_kernel<16>";
        }
    } else if (ckpt) {
        sa_launch_long_ckpt((int)c->ckpt_rows, wide ? 1 : 0, &P, grid, SA_LDS_PROF_BYTES + 64 + 2 * c->lslot, c->compute);
        c->last_kernel = wide ? "sa_long_ckpt_wide_kernel<64, 64>"        // one form: a checkpoint every 64 rows
                       : c->ckpt_rows == 64 ? "sa_long_ckpt_kernel<64, 64>" : "sa_long_ckpt_kernel<32, 64>";
    } else if (is_long(c) && tri && w.paired) {
        // + 64: rows read table entries two groups ahead, up to 16 bytes past the last profile row; + 2 lslot: the two first sequences
        sa_launch_long_pair(wide ? 1 : 0, &P, grid, SA_LDS_PROF_BYTES + 64 + 2 * c->lslot, c->compute);
        c->last_kernel = wide ? "sa_long_pair_wide_kernel<4>" : "sa_long_pair_kernel<4>";
    } else if (is_long(c)) {
        // + 64 as above; + lslot: copy of the shared first sequence
        sa_launch_long(tri ? 0 : 1, &P, grid, SA_LDS_PROF_BYTES + 64 + c->lslot, c->compute);
        c->last_kernel = tri ? "sa_long_kernel<0, 2>" : "sa_long_kernel<1, 2>";
    } else if (tri && w.paired) {
        sa_launch_peptide_pair(pair_groups(c), wide ? 1 : 0, &P, grid, SA_PAIR_LDS_BYTES(pair_groups(c)), c->compute);
        c->last_kernel = wide ? (pair_groups(c) <= 13 ? "sa_peptide_pair_wide_kernel<13, 4>" : "sa_peptide_pair_wide_kernel<16, 4>")
                              : (pair_groups(c) <= 13 ? "sa_peptide_pair_kernel<13, 4>" : "sa_peptide_pair_kernel<16, 4>");
    } else {
        const int waves = (c->wg_per_cu == 0 || c->wg_per_cu > 8) ? 3 : 2;
        sa_launch_peptide(tri ? 0 : 1, waves, &P, grid, SA_LDS_FIXED_BYTES, c->compute);
        c->last_kernel = !tri ? "sa_peptide_kernel<1, 2>" : waves == 3 ? "sa_peptide_kernel<0, 3>" : "sa_peptide_kernel<0, 2>";
    }
    if (w.nshort) {
        SaPepParams Q = P;
        Q.items = (const SaItem*)w.items.p + nlong;
        Q.nitems = w.nshort;
        Q.counter = (uint32_t*)w.counter.p + kShortQueueWord;      // a queue head and status word of its own
        Q.spill = (uint32_t*)c->d_spill.p;
        Q.spill_rows = 2 * SA_LP;
        Q.phase_cycles = nullptr;
        const bool short_wide = packed_cells_mode(c, false) == kCellsWide;
        sa_launch_peptide_pair_longrec(short_ng, short_wide ? 1 : 0, &Q, short_grid, short_lds, c->compute);
        const char* short_name = short_wide ? (short_ng == 13 ? "sa_peptide_pair_wide_longrec_kernel<13, 4>" : "sa_peptide_pair_wide_longrec_kernel<16, 4>")
                                            : (short_ng == 13 ? "sa_peptide_pair_longrec_kernel<13, 4>" : "sa_peptide_pair_longrec_kernel<16, 4>");
        c->last_kernel_both = nlong ? std::string(c->last_kernel) + " + " + short_name : std::string(short_name);
        c->last_kernel = c->last_kernel_both.c_str();
    }
    SA_HIP_TRY(hipGetLastError());
    if (timed) SA_HIP_TRY(hipEventRecord(w.t1, c->compute));
    if ((rc = w.status_pin.reserve(16))) return rc;
    ((uint32_t*)w.status_pin.p)[1] = 0;
    SA_HIP_TRY(hipMemcpyAsync(w.status_pin.p, (const uint32_t*)w.counter.p + SA_STATUS_WORD, 4, hipMemcpyDeviceToHost, c->compute));
    if (w.nshort)
        SA_HIP_TRY(hipMemcpyAsync((uint32_t*)w.status_pin.p + 1, (const uint32_t*)w.counter.p + kShortQueueWord + SA_STATUS_WORD, 4,
                                  hipMemcpyDeviceToHost, c->compute));
    return SA_OK;
}

// after the launch's stream work is complete: did a kernel refuse to run?
int launch_status(const Work& w) {
    if (w.npairs == 0 || w.nitems == 0 || !w.status_pin.p) return SA_OK;
    const uint32_t* st = (const uint32_t*)w.status_pin.p;
    return (st[0] | st[1]) ? SA_ERR_KERNEL : SA_OK;
}

int prepare_work_impl(sa_ctx* c, Work& w, int schedule, uint64_t r0, uint64_t r1, bool need_host);
// no exception leaves the C ABI: a work list that does not fit the host's memory is SA_ERR_NOMEM
int prepare_work(sa_ctx* c, Work& w, int schedule, uint64_t r0, uint64_t r1, bool need_host) {
    try {
        return prepare_work_impl(c, w, schedule, r0, r1, need_host);
    } catch (const std::bad_alloc&) {
        return SA_ERR_NOMEM;
    }
}

int prepare_work_impl(sa_ctx* c, Work& w, int schedule, uint64_t r0, uint64_t r1, bool need_host) {
    std::vector<SaItem> items;
    bool paired = true;
    int rc = SA_OK;
    w.first_pair = sa_pair_index(c, schedule, r0);
    w.npairs = sa_pair_count(c, schedule, r0, r1);
    w.coop = use_coop(c, w.npairs);
    w.nshort = 0;
    if (w.coop) {
        // one pair per item, the most expensive first (cost = strips x steps per strip)
        paired = false;
        items.reserve(w.npairs);
        SaItem it;
        memset(&it, 0, sizeof(it));
        it.count = 1;
        uint64_t idx = 0;
        for (uint64_t i = r0; i < r1; i++) {
            const uint64_t j_end = schedule == SA_SCHEDULE_ADJACENT ? i + 2 : c->nseq;
            for (uint64_t j = i + 1; j < j_end; j++, idx++) {
                it.p = (uint32_t)i; it.v0 = (uint32_t)j;
                it.b0 = (uint32_t)idx; it.c0 = (uint32_t)(idx >> 32);
                items.push_back(it);
            }
        }
        std::stable_sort(items.begin(), items.end(), [c](const SaItem& a, const SaItem& b) {
            return (uint64_t)((c->len[a.p] + 63) / 64) * (c->len[a.v0] + 63) > (uint64_t)((c->len[b.p] + 63) / 64) * (c->len[b.v0] + 63);
        });
    } else {
        rc = build_items(c, schedule, r0, r1, items, paired, &w.nshort);
    }
    if (rc) return rc;
    w.schedule = schedule;
    w.paired = paired;
    w.nitems = (uint32_t)items.size();
    if ((rc = w.counter.reserve(256))) return rc;
    if ((rc = w.items.reserve(std::max<size_t>(16, items.size() * sizeof(SaItem))))) return rc;
    if ((rc = w.out.reserve(std::max<size_t>(16, w.npairs * (size_t)c->stride)))) return rc;
    if (need_host && (rc = w.host.reserve(std::max<size_t>(16, w.npairs * (size_t)c->stride)))) return rc;
    if (!items.empty()) {
        // through a pinned buffer of the slot: the copy is queued behind the previous submission's kernels
        // without the host waiting for them (the slot is reused only after its own kernels are done)
        if ((rc = w.items_pin.reserve(items.size() * sizeof(SaItem)))) return rc;
        memcpy(w.items_pin.p, items.data(), items.size() * sizeof(SaItem));
        SA_HIP_TRY(hipMemcpyAsync(w.items.p, w.items_pin.p, items.size() * sizeof(SaItem), hipMemcpyHostToDevice, c->compute));
    }
    return SA_OK;
}

int init_events(Work& w) {
    SA_HIP_TRY(hipEventCreate(&w.kernel_done));
    SA_HIP_TRY(hipEventCreate(&w.copied));
    SA_HIP_TRY(hipEventCreate(&w.t0));
    SA_HIP_TRY(hipEventCreate(&w.t1));
    return SA_OK;
}

void free_work(Work& w) {
    w.items.release(); w.out.release(); w.counter.release(); w.host.release();
    w.lens.release(); w.offs.release(); w.text.release(); w.total.release(); w.scan_tmp.release();
    w.host_total.release(); w.items_pin.release(); w.status_pin.release(); w.packed.release();
    if (w.kernel_done) (void)hipEventDestroy(w.kernel_done);
    if (w.copied) (void)hipEventDestroy(w.copied);
    if (w.t0) (void)hipEventDestroy(w.t0);
    if (w.t1) (void)hipEventDestroy(w.t1);
}

}  // namespace

extern "C" {

void sa_default_matrix(int32_t S[400]) {
    for (int k = 0; k < 400; k++) S[k] = kBlosum50[k];
}

const char* sa_strerror(int code) {
    switch (code) {
        case SA_OK: return "ok";
        case SA_ERR_ARG: return "bad argument";
        case SA_ERR_ALPHABET: return "residue outside the 20-letter alphabet";
        case SA_ERR_LENGTH: return "empty sequence or sequence longer than the context allows";
        case SA_ERR_NOMEM: return "out of host memory";
        case SA_ERR_STATE: return "call out of order";
        case SA_ERR_BUSY: return "two submissions already in flight";
        case SA_ERR_RECORD: return "record does not match the sequences";
        case SA_ERR_KERNEL: return "a kernel refused its launch configuration (dynamic LDS not at offset 0)";
        case SA_ERR_NOALLOC_SIZE: return "more pairs than the fixed scratch buffer of sa_align_adjacent_noalloc serves";
        default: break;
    }
    if (code <= SA_ERR_HIP) return hipGetErrorString((hipError_t)(SA_ERR_HIP - code));
    return "unknown error";
}

int sa_encode(const char* seq, uint32_t len, uint8_t* codes) {
    static const char alphabet[] = SA_ALPHABET;
    int8_t lut[128];
    memset(lut, -1, sizeof(lut));
    for (int k = 0; k < 20; k++) lut[(int)alphabet[k]] = (int8_t)k;
    for (uint32_t i = 0; i < len; i++) {
        const unsigned char ch = (unsigned char)seq[i];
        if (ch > 127 || lut[ch] < 0) return SA_ERR_ALPHABET;
        codes[i] = (uint8_t)lut[ch];
    }
    return SA_OK;
}

int sa_similarity_bp(int32_t matches, int32_t columns) {
    if (columns <= 0) return 0;
    // quotient first, then the scaling, both in IEEE double, then truncation
    volatile double q = (double)matches / (double)columns;
    volatile double scaled = q * 10000.0;
    return (int)scaled;
}

int sa_init(sa_ctx** out, const int32_t* S, int32_t gap, uint32_t max_seq_len, int device) {
    if (!out || max_seq_len == 0) return SA_ERR_ARG;
    if (max_seq_len > SA_MAX_LONG_LEN) return SA_ERR_LENGTH;
    {
        // the 32-bit cells hold 4*D + tag with D <= max_seq_len * max(S - 2*gap): refuse tables that could wrap
        int32_t dflt[400];
        if (!S) sa_default_matrix(dflt);
        int64_t low, top;
        table_range(S ? S : dflt, gap, low, top);
        if ((int64_t)max_seq_len * std::max<int64_t>(std::llabs(low), std::llabs(top)) >= (1ll << 29)) return SA_ERR_ARG;
    }
    if (device == SA_DEVICE_NONE) {
        sa_ctx* h = new (std::nothrow) sa_ctx();
        if (!h) return SA_ERR_NOMEM;
        h->device = SA_DEVICE_NONE;
        h->ncu = 256;
        if (S) memcpy(h->S, S, sizeof(h->S)); else sa_default_matrix(h->S);
        h->gap = gap;
        h->max_len = max_seq_len;
        table_range(h->S, gap, h->sprime_min, h->sprime_max);
        *out = h;
        return SA_OK;
    }
    int ndev = 0;
    SA_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return SA_ERR_ARG;
    SA_HIP_TRY(hipSetDevice(device));
    sa_ctx* c = new (std::nothrow) sa_ctx();
    if (!c) return SA_ERR_NOMEM;
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete c; return hip_code(e); }
    c->ncu = prop.multiProcessorCount;
    if (S) memcpy(c->S, S, sizeof(c->S)); else sa_default_matrix(c->S);
    c->gap = gap;
    c->max_len = max_seq_len;
    table_range(c->S, gap, c->sprime_min, c->sprime_max);
    int rc = SA_OK;
    do {
        if ((e = hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking)) != hipSuccess) { rc = hip_code(e); break; }
        if ((e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking)) != hipSuccess) { rc = hip_code(e); break; }
        if ((rc = init_events(c->slots[0]))) break;
        if ((rc = init_events(c->slots[1]))) break;
        if ((rc = init_events(c->resident))) break;
        // S' = S - 2*gap, padded: row 20 (padding residue) and columns >= 20 score 0
        std::vector<int32_t> sp(21 * SA_SPRIME_STRIDE, 0);
        for (int a = 0; a < 20; a++)
            for (int b = 0; b < 20; b++) sp[a * SA_SPRIME_STRIDE + b] = c->S[a * 20 + b] - 2 * gap;
        if ((rc = c->d_sprime.reserve(sp.size() * 4))) break;
        if ((e = hipMemcpy(c->d_sprime.p, sp.data(), sp.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) { rc = hip_code(e); break; }
    } while (0);
    if (rc) { sa_destroy(c); return rc; }
    *out = c;
    return SA_OK;
}

void sa_destroy(sa_ctx* c) {
    if (!c) return;
    if (c->device == SA_DEVICE_NONE) { delete c; return; }
    (void)hipSetDevice(c->device);
    if (c->compute) (void)hipStreamSynchronize(c->compute);
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    free_work(c->slots[0]); free_work(c->slots[1]); free_work(c->resident);
    c->d_codes.release(); c->d_len.release(); c->d_codes_t.release(); c->d_fi_sorted.release();
    c->d_len_sorted.release(); c->d_bucket_base.release(); c->d_cursor.release(); c->d_sprime.release();
    c->d_spill.release(); c->d_bcol.release(); c->d_codes_s.release(); c->d_coop.release();
    c->d_other.release(); c->d_other_off.release(); c->d_cp.release(); c->d_cq.release();
    c->text_ring[0].release(); c->text_ring[1].release();
    c->d_phase.release();
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    delete c;
}

int sa_set_tuning(sa_ctx* c, uint32_t tiles_per_item, uint32_t wg_per_cu) {
    if (!c) return SA_ERR_ARG;
    if (tiles_per_item) c->tiles_per_item = tiles_per_item;
    if (wg_per_cu) c->wg_per_cu = wg_per_cu;
    c->resident_ready = false;
    return SA_OK;
}

int sa_set_bucket_width(sa_ctx* c, uint32_t width) {
    if (!c) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    c->bucket_width_forced = width;
    c->tri_ready = false;          // the bucket order is rebuilt on the next TRIANGULAR use
    c->resident_ready = false;
    return SA_OK;
}

int sa_upload(sa_ctx* c, const uint8_t* codes, const uint64_t* offs, const uint32_t* lens, uint64_t nseq) {
    if (!c || !codes || !offs || !lens) return SA_ERR_ARG;
    if (nseq == 0 || nseq >= 0xFFFFFFF0ull) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    if (c->device != SA_DEVICE_NONE) SA_HIP_TRY(hipSetDevice(c->device));
    // validation and the length range, on the host threads
    struct Part { uint32_t lmin = 0xFFFFFFFFu, lmax = 0; uint64_t total = 0; int err = SA_OK; };
    std::vector<Part> part(sa_host_threads());
    const uint32_t max_len = c->max_len;
    sa_parallel_for(nseq, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned t) {
        Part q;
        for (uint64_t s = lo; s < hi && q.err == SA_OK; s++) {
            const uint32_t l = lens[s];
            if (l == 0 || l > max_len) { q.err = SA_ERR_LENGTH; break; }
            q.lmin = std::min(q.lmin, l); q.lmax = std::max(q.lmax, l);
            q.total = std::max(q.total, offs[s] + l);
            const uint8_t* src = codes + offs[s];
            uint32_t worst = 0;
            for (uint32_t i = 0; i < l; i++) worst = std::max<uint32_t>(worst, src[i]);
            if (worst >= 20) q.err = SA_ERR_ALPHABET;
        }
        part[t] = q;
    });
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0;
    uint64_t total = 0;
    for (const Part& q : part) {
        if (q.err) return q.err;
        lmin = std::min(lmin, q.lmin); lmax = std::max(lmax, q.lmax); total = std::max(total, q.total);
    }
    c->nseq = nseq;
    c->min_seq = lmin;
    c->max_seq = lmax;
    const bool lng = lmax > SA_LP;
    const uint32_t lslot = lng ? (lmax + 63u) / 64u * 64u : SA_LP;
    c->spill_grid = 0;                            // scratch areas are sized by the slot and the longest sequence
    if (c->ckpt_auto) c->ckpt_rows = long_auto_checkpoint_rows(lmax);
    c->lslot = lslot;
    c->row_bytes = lng ? 2 : 1;
    c->stride = lng ? (SA_LONG_HEADER_BYTES + 2 * lmax + 15u) & ~15u : (SA_REC_HEADER_BYTES + lmax + 15u) & ~15u;
    c->resident_ready = false;
    c->tri_ready = false;

    // host mirrors: lengths, suffix sums of the lengths, the caller's arrays
    c->len.resize(nseq);
    sa_parallel_for(nseq, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t s = lo; s < hi; s++) c->len[s] = (uint16_t)lens[s];
    });
    {
        std::lock_guard<std::mutex> lock(c->tables_mu);     // rebuilt for the new sequences when first asked for
        c->len_suffix.clear(); c->cp.clear(); c->cq.clear();
        c->compact_on_device = false;
    }
    if (c->device == SA_DEVICE_NONE) {
        // no device to read the slots back from: a host-only context keeps them itself
        c->slots_host.assign((size_t)(nseq + 2) * lslot, SA_PAD_CODE);
        for (uint64_t s = 0; s < nseq; s++) memcpy(&c->slots_host[(size_t)s * lslot], codes + offs[s], lens[s]);
    } else {
        c->slots_host.clear();
    }
    if (c->device == SA_DEVICE_NONE) return SA_OK;

    // device: the compact arrays go up as they are; a kernel spreads them into the file-order
    // slots padded with SA_PAD_CODE (plus two spare slots) and the 16-bit length table
    int rc;
    const uint64_t nslots = nseq + 2;
    const size_t raw_bytes = (total + 64 + 15) & ~(size_t)15;   // slack: the kernel may read a few bytes past a sequence
    DevBuf stage;
    if ((rc = stage.reserve(raw_bytes + nseq * 8 + nseq * 4))) return rc;
    uint8_t* d_raw = (uint8_t*)stage.p;
    uint64_t* d_offs = (uint64_t*)(d_raw + raw_bytes);
    uint32_t* d_lens = (uint32_t*)(d_raw + raw_bytes + nseq * 8);
    if ((rc = c->d_codes.reserve((size_t)nslots * lslot)) || (rc = c->d_len.reserve((nseq + 64) * 2))) { stage.release(); return rc; }
    hipError_t e = hipMemcpyAsync(d_raw, codes, total, hipMemcpyHostToDevice, c->compute);
    if (e == hipSuccess) e = hipMemcpyAsync(d_offs, offs, nseq * 8, hipMemcpyHostToDevice, c->compute);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lens, lens, nseq * 4, hipMemcpyHostToDevice, c->compute);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_len.p, 0, (nseq + 64) * 2, c->compute);
    if (e == hipSuccess) {
        sa_launch_pack_slots(d_raw, d_offs, d_lens, nseq, nslots, lslot, (uint8_t*)c->d_codes.p, (uint16_t*)c->d_len.p, c->compute);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->compute);
    stage.release();
    if (e != hipSuccess) return hip_code(e);
    return SA_OK;
}

uint64_t sa_schedule_rows(const sa_ctx* c, int schedule) {
    if (!c || c->nseq < 2) return 0;
    (void)schedule;
    return c->nseq - 1;   // ADJACENT: pairs 0..n-2; TRIANGULAR: first rows 0..n-2
}

uint64_t sa_pair_index(const sa_ctx* c, int schedule, uint64_t row) {
    if (!c) return 0;
    if (schedule == SA_SCHEDULE_ADJACENT) return row;
    return tri_row_start(c->nseq, row);
}

uint64_t sa_pair_count(const sa_ctx* c, int schedule, uint64_t r0, uint64_t r1) {
    if (!c || r1 <= r0) return 0;
    return sa_pair_index(c, schedule, r1) - sa_pair_index(c, schedule, r0);
}

uint64_t sa_cell_count(const sa_ctx* c, int schedule, uint64_t r0, uint64_t r1) {
    if (!c || r1 <= r0) return 0;
    if (schedule != SA_SCHEDULE_ADJACENT) ensure_suffix(c);
    std::vector<uint64_t> part(sa_host_threads(), 0);
    sa_parallel_for(r1 - r0, 1 << 18, [&](uint64_t lo, uint64_t hi, unsigned t) {
        uint64_t cells = 0;
        for (uint64_t i = r0 + lo; i < r0 + hi; i++) {
            if (schedule == SA_SCHEDULE_ADJACENT) cells += (uint64_t)c->len[i] * c->len[i + 1];
            else cells += (uint64_t)c->len[i] * c->len_suffix[i + 1];
        }
        part[t] = cells;
    });
    uint64_t cells = 0;
    for (uint64_t a : part) cells += a;
    return cells;
}

uint32_t sa_record_stride(const sa_ctx* c) { return c ? c->stride : 0; }
uint32_t sa_record_row_bytes(const sa_ctx* c) { return c ? c->row_bytes : 0; }

int sa_shard_rows(const sa_ctx* c, int schedule, uint32_t parts, uint64_t* bounds) {
    if (!c || !bounds || parts == 0) return SA_ERR_ARG;
    const uint64_t rows = sa_schedule_rows(c, schedule);
    const uint64_t total = sa_cell_count(c, schedule, 0, rows);      // builds the suffix sums a TRIANGULAR split needs
    uint64_t acc = 0, row = 0;
    bounds[0] = 0;
    for (uint32_t k = 1; k < parts; k++) {
        const uint64_t target = total / parts * k;
        while (row < rows && acc < target) {
            acc += schedule == SA_SCHEDULE_ADJACENT ? (uint64_t)c->len[row] * c->len[row + 1]
                                                    : (uint64_t)c->len[row] * c->len_suffix[row + 1];
            row++;
        }
        bounds[k] = row;
    }
    bounds[parts] = rows;
    return SA_OK;
}

int sa_prepare_schedule(sa_ctx* c, int schedule) {
    if (!c || (schedule != SA_SCHEDULE_ADJACENT && schedule != SA_SCHEDULE_TRIANGULAR)) return SA_ERR_ARG;
    if (c->nseq < 2) return SA_ERR_STATE;
    if (schedule == SA_SCHEDULE_ADJACENT) return SA_OK;
    if (c->device != SA_DEVICE_NONE) SA_HIP_TRY(hipSetDevice(c->device));
    return ensure_triangular(c);
}

int sa_clone_sequences(sa_ctx* dst, sa_ctx* src) {
    if (!dst || !src || dst == src) return SA_ERR_ARG;
    if (src->nseq == 0 || dst->inflight || src->max_seq > dst->max_len) return SA_ERR_STATE;
    if ((dst->device == SA_DEVICE_NONE) != (src->device == SA_DEVICE_NONE)) return SA_ERR_ARG;
    dst->nseq = src->nseq; dst->min_seq = src->min_seq; dst->max_seq = src->max_seq;
    if (dst->ckpt_auto) dst->ckpt_rows = long_auto_checkpoint_rows(dst->max_seq);
    dst->spill_grid = 0;
    dst->lslot = src->lslot; dst->row_bytes = src->row_bytes; dst->stride = src->stride;
    dst->len = src->len; dst->slots_host = src->slots_host;
    {
        std::lock_guard<std::mutex> lock_dst(dst->tables_mu);
        std::lock_guard<std::mutex> lock_src(src->tables_mu);
        dst->len_suffix = src->len_suffix; dst->cp = src->cp; dst->cq = src->cq;      // as far as the donor has built them
        dst->compact_on_device = false;
    }
    dst->resident_ready = false;
    dst->tri_ready = false;
    const bool dev = dst->device != SA_DEVICE_NONE;
    // device to device: between two GPUs the runtime takes the peer link (or stages through the host)
    auto copy = [&](DevBuf& to, const DevBuf& from, size_t bytes) -> int {
        int rc = to.reserve(std::max<size_t>(16, bytes));
        if (rc) return rc;
        if (bytes) SA_HIP_TRY(hipMemcpy(to.p, from.p, bytes, hipMemcpyDeviceToDevice));
        return SA_OK;
    };
    int rc;
    if (dev) {
        SA_HIP_TRY(hipSetDevice(dst->device));
        if ((rc = copy(dst->d_codes, src->d_codes, (size_t)(src->nseq + 2) * src->lslot))) return rc;
        if ((rc = copy(dst->d_len, src->d_len, (src->nseq + 64) * 2))) return rc;
    }
    if (src->tri_ready) {
        dst->nb = src->nb; dst->npad = src->npad; dst->bucket_width = src->bucket_width;
        dst->short_buckets = src->short_buckets; dst->short_max = src->short_max; dst->short_width = src->short_width;
        dst->bucket_base = src->bucket_base; dst->fi_sorted = src->fi_sorted; dst->cursor = src->cursor;
        if (dev) {
            const bool lng = src->lslot > SA_LP;
            if ((rc = copy(dst->d_codes_t, src->d_codes_t, (size_t)src->lslot * src->npad))) return rc;
            if ((rc = copy(dst->d_codes_s, src->d_codes_s, (lng && src->short_buckets == 0) ? 16 : (size_t)src->npad * SA_LP))) return rc;
            if ((rc = copy(dst->d_fi_sorted, src->d_fi_sorted, (size_t)src->npad * 4))) return rc;
            if ((rc = copy(dst->d_len_sorted, src->d_len_sorted, (size_t)src->npad * 2))) return rc;
            if ((rc = copy(dst->d_bucket_base, src->d_bucket_base, src->bucket_base.size() * 4))) return rc;
            if ((rc = copy(dst->d_cursor, src->d_cursor, src->cursor.size() * 4))) return rc;
        }
        dst->tri_ready = true;
    }
    if (dev && src->other_set) {
        if ((rc = copy(dst->d_other, src->d_other, src->d_other.cap))) return rc;
        if ((rc = copy(dst->d_other_off, src->d_other_off, (src->nseq + 1) * 4))) return rc;
        dst->max_other = src->max_other;
        dst->other_set = true;
    }
    if (src->fmt_set) { dst->fmt = src->fmt; dst->fmt_set = true; }
    return SA_OK;
}

int sa_submit(sa_ctx* c, int schedule, uint64_t r0, uint64_t r1) {
    if (!c || (schedule != SA_SCHEDULE_ADJACENT && schedule != SA_SCHEDULE_TRIANGULAR)) return SA_ERR_ARG;
    if (c->nseq < 2 || c->device == SA_DEVICE_NONE) return SA_ERR_STATE;
    if (r1 < r0 || r1 > sa_schedule_rows(c, schedule)) return SA_ERR_ARG;
    int head;
    {
        std::lock_guard<std::mutex> lock(c->slot_mu);
        if (c->inflight == 2) return SA_ERR_BUSY;
        head = c->head;
    }
    SA_HIP_TRY(hipSetDevice(c->device));
    Work& w = c->slots[head];
    const bool compact = c->compact;
    int rc = prepare_work(c, w, schedule, r0, r1, !compact);
    if (rc) return rc;
    w.compact = compact;
    w.r0 = r0; w.r1 = r1;
    w.packed_bytes = 0;
    if (compact && w.npairs) {
        // what leaves the device: the records cut to their own lengths (sa_compact.hip)
        if ((rc = ensure_compact_on_device(c))) return rc;
        w.packed_bytes = schedule == SA_SCHEDULE_ADJACENT ? c->cp[r1 + 1] - c->cp[r0 + 1] : c->cq[r1] - c->cq[r0];
        // buffers sized as for full records: the packed size changes from submission to submission with the lengths of
        // the sequences in it, and a pinned buffer that has to grow is freed and allocated again (tens of milliseconds)
        if ((rc = w.packed.reserve(std::max<size_t>(16, w.npairs * (size_t)c->stride)))) return rc;
        if ((rc = w.host.reserve(std::max<size_t>(16, w.npairs * (size_t)c->stride)))) return rc;
    }
    if ((rc = launch_work(c, w, false))) return rc;
    if (compact && w.npairs) {
        sa_launch_compact(schedule, (const uint8_t*)w.out.p, (uint8_t*)w.packed.p, (const uint64_t*)c->d_cp.p, (const uint64_t*)c->d_cq.p,
                          c->nseq, r0, r1, c->stride, c->compute);
        SA_HIP_TRY(hipGetLastError());
    }
    SA_HIP_TRY(hipEventRecord(w.kernel_done, c->compute));
    SA_HIP_TRY(hipStreamWaitEvent(c->copy, w.kernel_done, 0));
    if (w.npairs) {
        if (compact) SA_HIP_TRY(hipMemcpyAsync(w.host.p, w.packed.p, w.packed_bytes, hipMemcpyDeviceToHost, c->copy));
        else SA_HIP_TRY(hipMemcpyAsync(w.host.p, w.out.p, w.npairs * (size_t)c->stride, hipMemcpyDeviceToHost, c->copy));
    }
    SA_HIP_TRY(hipEventRecord(w.copied, c->copy));
    w.busy = true;
    w.text_mode = false;
    std::lock_guard<std::mutex> lock(c->slot_mu);
    c->head ^= 1;
    c->inflight++;
    return SA_OK;
}

int sa_next(sa_ctx* c, const uint8_t** records, uint64_t* first_pair, uint64_t* npairs) {
    if (!c || !records || !first_pair || !npairs) return SA_ERR_ARG;
    int tail;
    {
        std::lock_guard<std::mutex> lock(c->slot_mu);
        if (c->inflight == 0) return SA_ERR_STATE;
        tail = c->tail;
    }
    Work& w = c->slots[tail];
    if (w.text_mode) return SA_ERR_STATE;
    SA_HIP_TRY(hipSetDevice(c->device));
    SA_HIP_TRY(hipEventSynchronize(w.copied));
    if (int st = launch_status(w)) return st;
    *records = (const uint8_t*)w.host.p;
    *first_pair = w.first_pair;
    *npairs = w.npairs;
    return SA_OK;
}

int sa_release(sa_ctx* c) {
    if (!c) return SA_ERR_ARG;
    std::lock_guard<std::mutex> lock(c->slot_mu);
    if (c->inflight == 0) return SA_ERR_STATE;
    c->slots[c->tail].busy = false;
    c->tail ^= 1;
    c->inflight--;
    return SA_OK;
}

int sa_set_text_layout(sa_ctx* c, const sa_text_layout* L) {
    if (!c || !L || L->ncols == 0 || L->ncols > SA_FMT_MAX_COLS) return SA_ERR_ARG;
    memset(&c->fmt, 0, sizeof(c->fmt));
    c->fmt.ncols = L->ncols;
    for (uint32_t k = 0; k < L->ncols; k++) {
        if (L->kind[k] < SA_TEXT_NONE || L->kind[k] > SA_TEXT_OTHER2) return SA_ERR_ARG;
        c->fmt.kind[k] = L->kind[k];
    }
    // the format is split at its two "%s" (include/csv.h:40-54)
    const char* f = L->align_fmt;
    const char* end = f + strnlen(L->align_fmt, sizeof(L->align_fmt));
    uint32_t part = 0;
    const char* from = f;
    while (f < end) {
        if (f[0] == '%' && f + 1 < end && f[1] == 's' && part < 2) {
            c->fmt.part_len[part] = (uint32_t)(f - from);
            memcpy(c->fmt.part[part], from, c->fmt.part_len[part]);
            f += 2; from = f; part++;
        } else f++;
    }
    if (part != 2) return SA_ERR_ARG;
    c->fmt.part_len[2] = (uint32_t)(end - from);
    memcpy(c->fmt.part[2], from, c->fmt.part_len[2]);
    memcpy(c->fmt.alphabet, SA_ALPHABET "-", 21);
    c->fmt_set = true;
    return SA_OK;
}

int sa_upload_passthrough(sa_ctx* c, const char* bytes, const uint64_t* offs, uint64_t nseq) {
    if (!c || !offs || (!bytes && offs[nseq] != 0)) return SA_ERR_ARG;
    if (c->device == SA_DEVICE_NONE || nseq != c->nseq) return SA_ERR_STATE;
    if (offs[nseq] >= 0xFFFFFFF0ull) return SA_ERR_NOMEM;
    SA_HIP_TRY(hipSetDevice(c->device));
    std::unique_ptr<uint32_t[]> o32(new (std::nothrow) uint32_t[nseq + 1]);
    if (!o32) return SA_ERR_NOMEM;
    std::vector<uint32_t> longest_of(sa_host_threads(), 0);
    std::vector<int> bad_of(sa_host_threads(), 0);
    sa_parallel_for(nseq + 1, 1 << 16, [&](uint64_t lo, uint64_t hi, unsigned t) {
        uint32_t longest = 0;
        int bad = 0;
        for (uint64_t s = lo; s < hi; s++) {
            o32[s] = (uint32_t)offs[s];
            if (s && offs[s] < offs[s - 1]) bad = 1;
            if (s) longest = std::max(longest, (uint32_t)(offs[s] - offs[s - 1]));
        }
        longest_of[t] = longest;
        bad_of[t] = bad;
    });
    uint32_t longest = 0;
    for (unsigned t = 0; t < longest_of.size(); t++) {
        if (bad_of[t]) return SA_ERR_ARG;
        longest = std::max(longest, longest_of[t]);
    }
    int rc;
    if ((rc = c->d_other.reserve(std::max<size_t>(16, offs[nseq])))) return rc;
    if ((rc = c->d_other_off.reserve((nseq + 1) * 4))) return rc;
    if (offs[nseq]) SA_HIP_TRY(hipMemcpy(c->d_other.p, bytes, offs[nseq], hipMemcpyHostToDevice));
    SA_HIP_TRY(hipMemcpy(c->d_other_off.p, o32.get(), (nseq + 1) * 4, hipMemcpyHostToDevice));
    c->max_other = longest;
    c->other_set = true;
    return SA_OK;
}

int sa_submit_text(sa_ctx* c, int schedule, uint64_t r0, uint64_t r1) {
    if (!c || (schedule != SA_SCHEDULE_ADJACENT && schedule != SA_SCHEDULE_TRIANGULAR)) return SA_ERR_ARG;
    if (c->nseq < 2 || c->device == SA_DEVICE_NONE || !c->fmt_set || !c->other_set) return SA_ERR_STATE;
    if (r1 < r0 || r1 > sa_schedule_rows(c, schedule)) return SA_ERR_ARG;
    int head;
    {
        std::lock_guard<std::mutex> lock(c->slot_mu);
        if (c->inflight == 2) return SA_ERR_BUSY;
        head = c->head;
    }
    SA_HIP_TRY(hipSetDevice(c->device));
    Work& w = c->slots[head];
    int rc = prepare_work(c, w, schedule, r0, r1, false);
    if (rc) return rc;
    if ((rc = launch_work(c, w, false))) return rc;
    if (w.npairs) {
        // longest possible row: both sequences, both aligned strings (at most len1 + len2 columns each),
        // the pass-through columns, the format parts, five numbers, separators
        const size_t lmax = c->max_seq;
        const size_t bound = 6 * lmax + (size_t)c->fmt.ncols * (c->max_other + 1) + c->fmt.part_len[0] + c->fmt.part_len[1] +
                             c->fmt.part_len[2] + 5 * 12 + 8;
        const size_t scan_bytes = sa_format_scan_bytes(w.npairs);
        if ((rc = w.lens.reserve(w.npairs * 4)) || (rc = w.offs.reserve(w.npairs * 8)) || (rc = w.total.reserve(8)) ||
            (rc = w.scan_tmp.reserve(std::max<size_t>(16, scan_bytes))) || (rc = w.text.reserve(w.npairs * bound)) ||
            (rc = w.host_total.reserve(8)))
            return rc;
        SaFormatParams F = c->fmt;
        F.codes_fo = (const uint8_t*)c->d_codes.p;
        F.len_fo = (const uint16_t*)c->d_len.p;
        F.lslot = c->lslot;
        F.rec = (const uint8_t*)w.out.p;
        F.stride = c->stride;
        F.row_bytes = c->row_bytes;
        F.first_pair = w.first_pair;
        F.npairs = w.npairs;
        F.nseq = c->nseq;
        F.schedule = schedule;
        F.other = (const char*)c->d_other.p;
        F.other_off = (const uint32_t*)c->d_other_off.p;
        F.lens = (uint32_t*)w.lens.p;
        F.offs = (uint64_t*)w.offs.p;
        F.text = (char*)w.text.p;
        F.total = (uint64_t*)w.total.p;
        SA_HIP_TRY(sa_launch_format(&F, w.scan_tmp.p, scan_bytes, c->compute));
        SA_HIP_TRY(hipMemcpyAsync(w.host_total.p, w.total.p, 8, hipMemcpyDeviceToHost, c->compute));
    }
    SA_HIP_TRY(hipEventRecord(w.kernel_done, c->compute));
    w.busy = true;
    w.text_mode = true;
    std::lock_guard<std::mutex> lock(c->slot_mu);
    c->head ^= 1;
    c->inflight++;
    return SA_OK;
}

int sa_next_text(sa_ctx* c, const char** text, uint64_t* nbytes, uint64_t* first_pair, uint64_t* npairs) {
    if (!c || !text || !nbytes || !first_pair || !npairs) return SA_ERR_ARG;
    int tail;
    {
        std::lock_guard<std::mutex> lock(c->slot_mu);
        if (c->inflight == 0) return SA_ERR_STATE;
        tail = c->tail;
    }
    Work& w = c->slots[tail];
    if (!w.text_mode) return SA_ERR_STATE;
    SA_HIP_TRY(hipSetDevice(c->device));
    SA_HIP_TRY(hipEventSynchronize(w.kernel_done));
    if (int st = launch_status(w)) return st;
    uint64_t total = 0;
    if (w.npairs) {
        total = *(const uint64_t*)w.host_total.p;
    }
    PinBuf& dst = c->text_ring[c->text_ring_pos];
    c->text_ring_pos ^= 1;
    int rc = dst.reserve(std::max<size_t>(16, total));
    if (rc) return rc;
    if (total) {
        SA_HIP_TRY(hipMemcpyAsync(dst.p, w.text.p, total, hipMemcpyDeviceToHost, c->copy));
        SA_HIP_TRY(hipStreamSynchronize(c->copy));
    }
    *text = (const char*)dst.p;
    *nbytes = total;
    *first_pair = w.first_pair;
    *npairs = w.npairs;
    return SA_OK;
}

int sa_prepare_resident(sa_ctx* c, int schedule, uint64_t r0, uint64_t r1) {
    if (!c || (schedule != SA_SCHEDULE_ADJACENT && schedule != SA_SCHEDULE_TRIANGULAR)) return SA_ERR_ARG;
    if (c->nseq < 2 || c->device == SA_DEVICE_NONE) return SA_ERR_STATE;
    if (r1 < r0 || r1 > sa_schedule_rows(c, schedule)) return SA_ERR_ARG;
    SA_HIP_TRY(hipSetDevice(c->device));
    c->resident_ready = false;
    int rc = prepare_work(c, c->resident, schedule, r0, r1, false);
    if (rc) return rc;
    c->resident_ready = true;
    return SA_OK;
}

int sa_run_resident(sa_ctx* c, float* kernel_ms) {
    if (!c) return SA_ERR_ARG;
    if (!c->resident_ready) return SA_ERR_STATE;
    SA_HIP_TRY(hipSetDevice(c->device));
    int rc = launch_work(c, c->resident, true);
    if (rc) return rc;
    SA_HIP_TRY(hipStreamSynchronize(c->compute));
    if (int st = launch_status(c->resident)) return st;
    if (kernel_ms) {
        *kernel_ms = 0.f;
        if (c->resident.npairs && c->resident.nitems)
            SA_HIP_TRY(hipEventElapsedTime(kernel_ms, c->resident.t0, c->resident.t1));
    }
    return SA_OK;
}

int sa_fetch_resident(sa_ctx* c, uint64_t first, uint64_t n, uint8_t* dst) {
    if (!c || !dst) return SA_ERR_ARG;
    if (!c->resident_ready) return SA_ERR_STATE;
    if (first + n > c->resident.npairs) return SA_ERR_ARG;
    SA_HIP_TRY(hipSetDevice(c->device));
    if (n)
        SA_HIP_TRY(hipMemcpy(dst, (const uint8_t*)c->resident.out.p + first * c->stride, n * (size_t)c->stride,
                             hipMemcpyDeviceToHost));
    return SA_OK;
}

int sa_digest_resident(sa_ctx* c, uint64_t first, uint64_t n, uint64_t digest[2]) {
    if (!c || !digest) return SA_ERR_ARG;
    if (!c->resident_ready) return SA_ERR_STATE;
    if (first + n > c->resident.npairs) return SA_ERR_ARG;
    SA_HIP_TRY(hipSetDevice(c->device));
    digest[0] = digest[1] = 0;
    if (n == 0) return SA_OK;
    int rc = c->resident.counter.reserve(256);      // bytes 64..79 of the queue-head buffer take the two sums
    if (rc) return rc;
    unsigned long long* acc = (unsigned long long*)((uint8_t*)c->resident.counter.p + 64);
    SA_HIP_TRY(hipMemsetAsync(acc, 0, 16, c->compute));
    sa_launch_digest((const uint8_t*)c->resident.out.p + first * c->stride, n * (uint64_t)c->stride, acc, c->compute);
    SA_HIP_TRY(hipGetLastError());
    SA_HIP_TRY(hipMemcpyAsync(digest, acc, 16, hipMemcpyDeviceToHost, c->compute));
    SA_HIP_TRY(hipStreamSynchronize(c->compute));
    return SA_OK;
}

int sa_set_cooperative(sa_ctx* c, int mode) {
    if (!c || mode < -1 || mode > 1) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    c->coop_mode = mode;
    c->resident_ready = false;
    return SA_OK;
}

int sa_set_cooperative_waves(sa_ctx* c, int waves) {
    if (!c || (waves != 0 && waves != 1 && waves != 2 && waves != 4 && waves != 8 && waves != 16)) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    c->coop_waves = waves;
    c->resident_ready = false;
    return SA_OK;
}

int sa_reserve_submission(sa_ctx* c, uint64_t npairs, uint32_t longest_sequence) {
    if (!c || npairs == 0 || longest_sequence == 0 || longest_sequence > c->max_len) return SA_ERR_ARG;
    if (c->device == SA_DEVICE_NONE) return SA_ERR_STATE;
    if (c->inflight) return SA_ERR_STATE;
    SA_HIP_TRY(hipSetDevice(c->device));
    const bool lng = longest_sequence > SA_LP;
    const size_t stride = lng ? (SA_LONG_HEADER_BYTES + 2 * (size_t)longest_sequence + 15u) & ~(size_t)15
                              : (SA_REC_HEADER_BYTES + (size_t)longest_sequence + 15u) & ~(size_t)15;
    const size_t bytes = npairs * stride;
    for (Work& w : c->slots) {
        int rc;
        if ((rc = w.out.reserve(bytes)) || (rc = w.host.reserve(bytes))) return rc;
        if (c->compact && (rc = w.packed.reserve(bytes))) return rc;
    }
    return SA_OK;
}

int sa_reserve_text(sa_ctx* c, uint64_t npairs, uint64_t text_bytes) {
    if (!c || npairs == 0) return SA_ERR_ARG;
    if (c->device == SA_DEVICE_NONE || c->inflight) return SA_ERR_STATE;
    SA_HIP_TRY(hipSetDevice(c->device));
    int rc;
    for (Work& w : c->slots) {
        if ((rc = w.lens.reserve(npairs * 4)) || (rc = w.offs.reserve(npairs * 8)) || (rc = w.total.reserve(8)) ||
            (rc = w.scan_tmp.reserve(std::max<size_t>(16, sa_format_scan_bytes(npairs)))) || (rc = w.text.reserve(text_bytes)) ||
            (rc = w.host_total.reserve(8)))
            return rc;
    }
    for (PinBuf& b : c->text_ring)
        if ((rc = b.reserve(text_bytes))) return rc;
    return SA_OK;
}

int sa_set_compact_records(sa_ctx* c, int enable) {
    if (!c) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    c->compact = enable != 0;
    return SA_OK;
}

uint32_t sa_compact_record_size(const sa_ctx* c, uint64_t second_sequence) {
    if (!c || second_sequence >= c->nseq) return 0;
    return packed_record_bytes(c, c->len[second_sequence]);
}

uint64_t sa_compact_pair_offset(const sa_ctx* c, int schedule, uint64_t i, uint64_t j) {
    if (!c || i >= j || j >= c->nseq) return 0;
    if (schedule == SA_SCHEDULE_ADJACENT && j != i + 1) return 0;
    ensure_compact_tables(c);
    return compact_offset(c, schedule, i, j);
}

uint64_t sa_compact_bytes(const sa_ctx* c, int schedule, uint64_t r0, uint64_t r1) {
    if (!c || r1 <= r0 || r1 > sa_schedule_rows(c, schedule)) return 0;
    ensure_compact_tables(c);
    return schedule == SA_SCHEDULE_ADJACENT ? c->cp[r1 + 1] - c->cp[r0 + 1] : c->cq[r1] - c->cq[r0];
}

int sa_set_long_checkpoint_rows(sa_ctx* c, uint32_t rows) {
    if (!c || (rows != 0 && rows != 32 && rows != 64 && rows != SA_LONG_CHECKPOINT_AUTO)) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    if (c->device != SA_DEVICE_NONE) {
        SA_HIP_TRY(hipSetDevice(c->device));
        SA_HIP_TRY(hipStreamSynchronize(c->compute));
    }
    c->ckpt_auto = rows == SA_LONG_CHECKPOINT_AUTO;
    c->ckpt_rows = c->ckpt_auto ? long_auto_checkpoint_rows(c->max_seq) : rows;
    c->spill_grid = 0;             // the scratch areas are laid out per kernel
    c->resident_ready = false;
    return SA_OK;
}

int sa_set_pairing(sa_ctx* c, int enable) {
    if (!c) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    c->pairing = enable != 0;
    c->resident_ready = false;
    return SA_OK;
}

const char* sa_last_kernel(const sa_ctx* c) { return c ? c->last_kernel : ""; }

int sa_preload_kernels(sa_ctx* c, int schedule, int device_text, int compact_records) {
    if (!c || (schedule != SA_SCHEDULE_ADJACENT && schedule != SA_SCHEDULE_TRIANGULAR)) return SA_ERR_ARG;
    if (c->device == SA_DEVICE_NONE) return SA_OK;
    SA_HIP_TRY(hipSetDevice(c->device));
    // slot packing, the 32-bit peptide kernels and the digest live in one code object; the rest as the caller says.
    // The long-sequence kernels are not guessed at: whether a set needs them is known only once it has been read.
    if (sa_load_code_peptide(c->compute)) { (void)hipGetLastError(); return SA_ERR_HIP; }
    {
        // the first copies in either direction set up the runtime's staging buffers and the copy engines' queues: a
        // megabyte from pageable memory takes 5.7 ms in a fresh process, and the first large copy after it 9 ms
        // instead of 16 (138 MB)
        DevBuf warm;
        const size_t bytes = 1u << 20;
        std::unique_ptr<char[]> host(new (std::nothrow) char[bytes]);
        if (host && warm.reserve(bytes) == SA_OK) {
            memset(host.get(), 0, bytes);
            (void)hipMemcpyAsync(warm.p, host.get(), bytes, hipMemcpyHostToDevice, c->compute);
            (void)hipStreamSynchronize(c->compute);
            (void)hipMemcpyAsync(host.get(), warm.p, 4096, hipMemcpyDeviceToHost, c->copy);
            (void)hipStreamSynchronize(c->copy);
        }
        warm.release();
        (void)hipGetLastError();
    }
    if (schedule == SA_SCHEDULE_TRIANGULAR && c->max_len <= SA_LP && c->pairing && sa_load_code_peptide_pair()) { (void)hipGetLastError(); return SA_ERR_HIP; }
    if (device_text && sa_load_code_format()) { (void)hipGetLastError(); return SA_ERR_HIP; }
    if (compact_records && sa_load_code_compact()) { (void)hipGetLastError(); return SA_ERR_HIP; }
    return SA_OK;
}

int sa_set_peptide_tiles(sa_ctx* c, int enable) {
    if (!c) return SA_ERR_ARG;
    if (c->inflight) return SA_ERR_STATE;
    c->peptide_tiles = enable != 0;
    c->resident_ready = false;
    return SA_OK;
}

int sa_last_grid(const sa_ctx* c, uint32_t* workgroups, uint32_t* wanted) {
    if (!c || !workgroups || !wanted) return SA_ERR_ARG;
    *workgroups = c->last_grid;
    *wanted = c->last_grid_wanted;
    return SA_OK;
}

int sa_debug_work_items(sa_ctx* c, int schedule, uint64_t r0, uint64_t r1, uint32_t* items8, uint64_t cap,
                        uint64_t* count) {
    if (!c || !count) return SA_ERR_ARG;
    if (c->nseq < 2) return SA_ERR_STATE;
    if (r1 < r0 || r1 > sa_schedule_rows(c, schedule)) return SA_ERR_ARG;
    std::vector<SaItem> items;
    bool paired = true;
    int rc = build_items(c, schedule, r0, r1, items, paired);
    if (rc) return rc;
    *count = items.size();
    if (items8) {
        const uint64_t n = std::min<uint64_t>(cap, items.size());
        if (n) memcpy(items8, items.data(), n * sizeof(SaItem));
    }
    return SA_OK;
}

int sa_debug_bucket_order(sa_ctx* c, uint32_t* file_index, uint64_t cap, uint64_t* count) {
    if (!c || !count) return SA_ERR_ARG;
    if (c->nseq < 2) return SA_ERR_STATE;
    int rc = ensure_triangular(c);
    if (rc) return rc;
    *count = c->nseq;
    if (file_index) memcpy(file_index, c->fi_sorted.data(), std::min<uint64_t>(cap, *count) * 4);
    return SA_OK;
}

int sa_debug_phase_cycles(sa_ctx* c, int enable, uint64_t totals[8]) {
    if (!c) return SA_ERR_ARG;
    if (c->device == SA_DEVICE_NONE) return SA_ERR_STATE;
    SA_HIP_TRY(hipSetDevice(c->device));
    if (totals) {
        for (int k = 0; k < 8; k++) totals[k] = 0;
        if (c->phase_timers && c->d_phase.p) {
            const uint32_t grid = grid_for(c);
            std::vector<uint64_t> h((size_t)grid * 8);
            SA_HIP_TRY(hipMemcpy(h.data(), c->d_phase.p, h.size() * 8, hipMemcpyDeviceToHost));
            for (uint32_t g = 0; g < grid; g++)
                for (int k = 0; k < 8; k++) totals[k] += h[(size_t)g * 8 + k];
        }
    }
    c->phase_timers = enable != 0;
    return SA_OK;
}

int sa_expand_record(const uint8_t* rec, uint32_t stride, uint32_t row_bytes, const char* seq1, uint32_t len1,
                     const char* seq2, uint32_t len2, char* a1, char* a2, sa_alignment_stats* st) {
    if (!rec || !seq1 || !seq2 || !a1 || !a2 || !st || (row_bytes != 1 && row_bytes != 2)) return SA_ERR_ARG;
    const uint32_t header = row_bytes == 1 ? SA_REC_HEADER_BYTES : SA_LONG_HEADER_BYTES;
    if (stride < header || (uint64_t)len2 * row_bytes > stride - header) return SA_ERR_RECORD;
    int32_t score; uint32_t matches, gaps;
    memcpy(&score, rec, 4);
    if (row_bytes == 1) {
        uint16_t m16, g16;
        memcpy(&m16, rec + 4, 2); memcpy(&g16, rec + 6, 2);
        matches = m16; gaps = g16;
    } else {
        memcpy(&matches, rec + 4, 4); memcpy(&gaps, rec + 8, 4);
    }
    if (gaps > len2 || matches > len1) return SA_ERR_RECORD;
    const uint8_t* body = rec + header;
    const uint32_t columns = len1 + gaps;     // every column holds a residue of sequence 1 or one of its gaps
    uint32_t pos = columns, j = len1, ups = 0;
    for (uint32_t i = len2; i >= 1; i--) {
        uint32_t v;
        if (row_bytes == 1) v = body[i - 1];
        else { uint16_t v16; memcpy(&v16, body + 2 * (size_t)(i - 1), 2); v = v16; }
        const uint32_t run = v >> 1, up = v & 1;
        if (run > j || pos < run + 1) return SA_ERR_RECORD;
        for (uint32_t x = 0; x < run; x++) { --pos; a1[pos] = seq1[--j]; a2[pos] = '-'; }
        --pos;
        if (up) { a1[pos] = '-'; a2[pos] = seq2[i - 1]; ups++; }
        else {
            if (j == 0) return SA_ERR_RECORD;
            a1[pos] = seq1[--j]; a2[pos] = seq2[i - 1];
        }
    }
    if (pos != j || ups != gaps) return SA_ERR_RECORD;
    while (j > 0) { --pos; a1[pos] = seq1[--j]; a2[pos] = '-'; }
    a1[columns] = a2[columns] = '\0';
    st->score = score;
    st->columns = (int32_t)columns;
    st->matches = (int32_t)matches;
    st->gaps = (int32_t)gaps;
    st->mismatches = (int32_t)columns - (int32_t)matches - (int32_t)gaps;
    st->similarity_bp = sa_similarity_bp((int32_t)matches, (int32_t)columns);
    return SA_OK;
}

}  // extern "C"
