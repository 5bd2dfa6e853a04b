// sa_compact.hip -- records without their padding, for the trip over PCIe.
//
// The alignment kernels write one record of `stride` bytes per pair (sa_device.h), sized for the longest sequence: a
// pair whose second sequence has 28 residues uses 36 of 64 bytes.  For a submission whose records go to the host
// (sa_submit with sa_set_compact_records) this kernel packs them first: pair after pair in canonical order, each
// record cut to header + one code per row of ITS second sequence, rounded up to 4 bytes.  Where a pair's record
// starts follows from two prefix arrays over the sequences, which the host holds as well (sa_compact_pair_offset):
//   cp[s] = sum of the packed record sizes of second sequences 0 .. s-1
//   cq[i] = sum over first sequences 0 .. i-1 of the bytes of their TRIANGULAR rows, cp[n] - cp[i+1] each
// The reference hands whole Alignment structs of 280 bytes from its workers to its writer (include/thread.h:35-42,
// include/common.h:19-29); its "without write" mode (src/main.c:111-116) is what this path is measured as.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sa_device.h"

struct SaCompactParams {
    const uint8_t* out;          // strided records of the submission, pair 0 = its first pair
    uint8_t* packed;             // packed records of the submission
    const uint64_t* cp;          // [nseq + 1]
    const uint64_t* cq;          // [nseq]
    uint64_t nseq;
    uint64_t r0, r1;             // rows of the schedule in the submission
    uint32_t stride;
    int32_t schedule;            // 0 ADJACENT, 1 TRIANGULAR
};

// TRIANGULAR: workgroup (x, y) takes pairs 2048 y .. 2048 y + 2047 of schedule row r0 + x (first sequence i, second
// sequences j = i+1 .. n-1); ADJACENT: a workgroup takes 256 pairs.  A thread copies whole records, four bytes at a
// time: neighbouring threads write neighbouring bytes.
__global__ void __launch_bounds__(256) sa_compact_kernel(const SaCompactParams A) {
    if (A.schedule == 1) {
        const uint64_t i = A.r0 + blockIdx.x;
        const uint64_t n = A.nseq;
        const uint64_t row_first = i * (2 * n - i - 1) / 2 - A.r0 * (2 * n - A.r0 - 1) / 2;     // pair index inside the submission
        const uint64_t row_at = A.cq[i] - A.cq[A.r0];
        const uint64_t base = A.cp[i + 1];
        const uint64_t j_begin = i + 1 + (uint64_t)blockIdx.y * 2048;
        const uint64_t j_end = j_begin + 2048 < n ? j_begin + 2048 : n;
        for (uint64_t j = j_begin + threadIdx.x; j < j_end; j += blockDim.x) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(A.out + (row_first + (j - i - 1)) * A.stride);
            uint32_t* dst = reinterpret_cast<uint32_t*>(A.packed + row_at + (A.cp[j] - base));
            const uint32_t words = (uint32_t)(A.cp[j + 1] - A.cp[j]) >> 2;
            for (uint32_t w = 0; w < words; w++) dst[w] = src[w];
        }
    } else {
        const uint64_t k = A.r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;             // pair (k, k+1)
        if (k >= A.r1) return;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.out + (k - A.r0) * A.stride);
        uint32_t* dst = reinterpret_cast<uint32_t*>(A.packed + (A.cp[k + 1] - A.cp[A.r0 + 1]));
        const uint32_t words = (uint32_t)(A.cp[k + 2] - A.cp[k + 1]) >> 2;
        for (uint32_t w = 0; w < words; w++) dst[w] = src[w];
    }
}

extern "C" void sa_launch_compact(int schedule, const uint8_t* out, uint8_t* packed, const uint64_t* cp, const uint64_t* cq,
                                  uint64_t nseq, uint64_t r0, uint64_t r1, uint32_t stride, hipStream_t stream) {
    if (r1 <= r0) return;
    SaCompactParams A;
    A.out = out; A.packed = packed; A.cp = cp; A.cq = cq; A.nseq = nseq; A.r0 = r0; A.r1 = r1; A.stride = stride;
    A.schedule = schedule;
    // TRIANGULAR: the longest row of the submission is its first
    const dim3 grid = schedule == 1 ? dim3((unsigned)(r1 - r0), (unsigned)((nseq - 1 - r0 + 2047) / 2048))
                                    : dim3((unsigned)((r1 - r0 + 255) / 256));
    hipLaunchKernelGGL(sa_compact_kernel, grid, dim3(256), 0, stream, A);
}

// Loads this file's code object on the current device without launching anything (the runtime otherwise loads it at
// the first launch of one of its kernels: milliseconds on the critical path of a short job).
extern "C" int sa_load_code_compact(void) {
    hipFuncAttributes a;
    return (int)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(sa_compact_kernel));
}
