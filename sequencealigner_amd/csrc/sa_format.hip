// sa_format.hip -- CSV rows written on the device.
//
// The reference formats one output row per pair on its main thread (include/csv.h:84-141,
// include/fstring.h:25-55).  Here a submission can hand back finished CSV text instead of
// records: one thread per pair rebuilds the two aligned strings from the pair's record,
// prints the columns of the output layout (include/csv.h:38-82) and the similarity figure
// (int)(((double)matches / columns) * 10000) -- IEEE double on the device as on the host --
// into its slice of the text buffer.  Two passes: row lengths, an exclusive prefix sum
// over the pairs of the submission, then the rows themselves.  The host only writes.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "sa_device.h"

namespace {

__device__ __forceinline__ uint32_t digits_u(uint32_t v) {
    uint32_t n = 1;
    while (v >= 10) { v /= 10; n++; }
    return n;
}

// decimal text of v at p (the reference's int_to_str, include/fstring.h:25-55); returns the end
template <bool WRITE>
__device__ __forceinline__ uint32_t put_int(char* p, uint32_t at, int32_t v) {
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    const uint32_t n = digits_u(u);
    if (WRITE) {
        if (v < 0) p[at] = '-';
        uint32_t e = at + (v < 0 ? 1 : 0) + n;
        do { p[--e] = (char)('0' + u % 10); u /= 10; } while (u);
    }
    return at + (v < 0 ? 1 : 0) + n;
}

template <bool WRITE>
__device__ __forceinline__ uint32_t put_bytes(char* p, uint32_t at, const char* src, uint32_t n) {
    if (WRITE)
        for (uint32_t k = 0; k < n; k++) p[at + k] = src[k];
    return at + n;
}

// first sequence index of canonical triangular pair idx: the largest i with i*(2n-i-1)/2 <= idx
__device__ __forceinline__ uint64_t tri_row_of(uint64_t n, uint64_t idx) {
    const double b = 2.0 * (double)n - 1.0;
    double est = (b - sqrt(b * b - 8.0 * (double)idx)) * 0.5;
    uint64_t i = est > 0 ? (uint64_t)est : 0;
    if (i > n - 2) i = n - 2;
    while (i > 0 && i * (2 * n - i - 1) / 2 > idx) i--;
    while (i + 1 <= n - 2 && (i + 1) * (2 * n - i - 2) / 2 <= idx) i++;
    return i;
}

// One row of output for pair t of the submission.  WRITE = false: only its length.
template <bool WRITE>
__device__ __forceinline__ uint32_t emit_row(const SaFormatParams& F, uint64_t t, char* out) {
    uint64_t i, j;
    if (F.schedule == 0) { i = F.first_pair + t; j = i + 1; }
    else {
        const uint64_t idx = F.first_pair + t;
        i = tri_row_of(F.nseq, idx);
        j = i + 1 + (idx - i * (2 * F.nseq - i - 1) / 2);
    }
    const uint32_t len1 = F.len_fo[i], len2 = F.len_fo[j];
    const uint8_t* c1 = F.codes_fo + (size_t)i * F.lslot;
    const uint8_t* c2 = F.codes_fo + (size_t)j * F.lslot;
    const uint8_t* rec = F.rec + t * (size_t)F.stride;
    const int32_t score = *reinterpret_cast<const int32_t*>(rec);
    uint32_t matches, gaps;
    const uint8_t* body;
    if (F.row_bytes == 1) {
        matches = *reinterpret_cast<const uint16_t*>(rec + 4);
        gaps = *reinterpret_cast<const uint16_t*>(rec + 6);
        body = rec + SA_REC_HEADER_BYTES;
    } else {
        matches = *reinterpret_cast<const uint32_t*>(rec + 4);
        gaps = *reinterpret_cast<const uint32_t*>(rec + 8);
        body = rec + SA_LONG_HEADER_BYTES;
    }
    const uint32_t columns = len1 + gaps;     // every column holds a residue of sequence 1 or one of its gaps
    const char* alphabet = F.alphabet;

    uint32_t at = 0;
    for (uint32_t col = 0; col < F.ncols; col++) {
        if (col > 0) { if (WRITE) out[at] = ','; at++; }
        switch (F.kind[col]) {
            case SA_COL_OTHER1:
            case SA_COL_OTHER2: {
                const uint64_t s = F.kind[col] == SA_COL_OTHER1 ? i : j;
                const uint32_t o = F.other_off[s], n = F.other_off[s + 1] - o;
                at = put_bytes<WRITE>(out, at, F.other + o, n);
                break;
            }
            case SA_COL_SEQ1:
                if (WRITE) for (uint32_t k = 0; k < len1; k++) out[at + k] = alphabet[c1[k]];
                at += len1;
                break;
            case SA_COL_SEQ2:
                if (WRITE) for (uint32_t k = 0; k < len2; k++) out[at + k] = alphabet[c2[k]];
                at += len2;
                break;
            case SA_COL_SCORE: at = put_int<WRITE>(out, at, score); break;
            case SA_COL_ALIGN: {
                at = put_bytes<WRITE>(out, at, F.part[0], F.part_len[0]);
                char* a1 = out + at;
                at += columns;
                at = put_bytes<WRITE>(out, at, F.part[1], F.part_len[1]);
                char* a2 = out + at;
                at += columns;
                at = put_bytes<WRITE>(out, at, F.part[2], F.part_len[2]);
                if (WRITE) {
                    // rows of sequence 2 bottom-up: `run` LEFT moves, then UP or DIAG (see sa_expand_record)
                    uint32_t pos = columns, jj = len1;
                    for (uint32_t r = len2; r >= 1; r--) {
                        const uint32_t v = F.row_bytes == 1 ? body[r - 1] : reinterpret_cast<const uint16_t*>(body)[r - 1];
                        const uint32_t run = v >> 1, up = v & 1;
                        for (uint32_t x = 0; x < run && pos > 0 && jj > 0; x++) { --pos; a1[pos] = alphabet[c1[--jj]]; a2[pos] = '-'; }
                        if (pos == 0) break;
                        --pos;
                        const char b = alphabet[c2[r - 1]];
                        if (up) { a1[pos] = '-'; a2[pos] = b; }
                        else if (jj > 0) { a1[pos] = alphabet[c1[--jj]]; a2[pos] = b; }
                    }
                    while (jj > 0 && pos > 0) { --pos; a1[pos] = alphabet[c1[--jj]]; a2[pos] = '-'; }
                }
                break;
            }
            case SA_COL_MATCHES: at = put_int<WRITE>(out, at, (int32_t)matches); break;
            case SA_COL_MISMATCHES: at = put_int<WRITE>(out, at, (int32_t)columns - (int32_t)matches - (int32_t)gaps); break;
            case SA_COL_GAPS: at = put_int<WRITE>(out, at, (int32_t)gaps); break;
            case SA_COL_SIMILARITY: {
                // quotient first, then the scaling, both in IEEE double, then truncation (include/seqalign.h:109, include/csv.h:125)
                int32_t pct = 0;
                if (columns > 0) {
                    const double q = (double)matches / (double)columns;
                    pct = (int32_t)(q * 10000.0);
                }
                at = put_int<WRITE>(out, at, pct / 100);
                if (WRITE) out[at] = '.';
                at++;
                const int32_t dec = pct % 100;
                if (dec < 10) { if (WRITE) out[at] = '0'; at++; }
                at = put_int<WRITE>(out, at, dec);
                if (WRITE) out[at] = '%';
                at++;
                break;
            }
            default: break;
        }
    }
    if (WRITE) out[at] = '\n';
    return at + 1;
}

__global__ void __launch_bounds__(256) sa_row_lengths_kernel(const SaFormatParams F) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= F.npairs) return;
    F.lens[t] = emit_row<false>(F, t, nullptr);
}

__global__ void __launch_bounds__(256) sa_rows_kernel(const SaFormatParams F) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= F.npairs) return;
    emit_row<true>(F, t, F.text + F.offs[t]);
    if (t == F.npairs - 1) *F.total = F.offs[t] + F.lens[t];
}

}  // namespace

// bytes of temporary storage the prefix sum over n row lengths needs
extern "C" size_t sa_format_scan_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t*)nullptr, (uint64_t*)nullptr, (int)n);
    return bytes;
}

// lengths -> offsets -> rows, all on `stream`; F.lens, F.offs, F.text and F.total are device buffers
extern "C" hipError_t sa_launch_format(const SaFormatParams* F, void* scan_tmp, size_t scan_bytes, hipStream_t stream) {
    if (F->npairs == 0) return hipSuccess;
    const unsigned grid = (unsigned)((F->npairs + 255) / 256);
    hipLaunchKernelGGL(sa_row_lengths_kernel, dim3(grid), dim3(256), 0, stream, *F);
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, (const uint32_t*)F->lens, F->offs, (int)F->npairs, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sa_rows_kernel, dim3(grid), dim3(256), 0, stream, *F);
    return hipGetLastError();
}

// Loads this file's code object on the current device without launching anything (the runtime otherwise loads it at
// the first launch of one of its kernels: milliseconds on the critical path of a short job).
extern "C" int sa_load_code_format(void) {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(sa_rows_kernel));
}
