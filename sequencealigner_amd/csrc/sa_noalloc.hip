// sa_noalloc.hip -- sa_align_adjacent_noalloc: the ADJACENT peptide path for callers that cannot
// allocate device memory (code that must not call hipMalloc: serving loops, harnesses that forbid it).
// It synchronises with the host (status words come back before the main launch), so it is not for use
// inside a stream capture.
//
// The caller passes the sequences and the record buffer as device pointers.  Everything else the
// kernels need beyond inputs and outputs -- the work-queue head, the work list, the shifted
// substitution table and the direction rows of every workgroup -- lives in one file-scope device
// array, `sa_scratch`, of fixed size.  Every GPU the library's code object is loaded on holds its own
// copy of the array at its own address; the address is asked for once per device
// (hipGetSymbolAddress with that device current) and only that pointer is ever handed to kernels,
// memsets and copies.  The entry point calls no hipMalloc / hipFree / stream or event creation: it
// works on the null stream and returns when the records are complete.
//
// Calls on one GPU share that GPU's array: two overlapping calls on the same GPU race.  The array
// keeps the contents of the last call; the queue head and the status words are cleared at the start
// of every call, the work list and the table are rewritten for the call's size, and direction rows
// are always written by the fill before the traceback of the same pair reads them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include "../../include/sa_hip.h"
#include "sa_device.h"

extern "C" void sa_launch_peptide(int mode, int waves_per_simd, const SaPepParams* p, unsigned grid,
                                  unsigned lds_bytes, hipStream_t stream);

namespace {

// ---- carve-up of the scratch array (bytes)
constexpr size_t kHeadBytes = 256;                                     // [0] queue head, [16..] status words
constexpr size_t kTableBytes = 21 * SA_SPRIME_STRIDE * 4;              // S' = S - 2*gap, padded
constexpr size_t kTableOff = kHeadBytes;
constexpr size_t kItemsOff = 4096;
constexpr size_t kMaxItems = (SA_NOALLOC_MAX_PAIRS + 63) / 64;
constexpr size_t kItemsBytes = kMaxItems * sizeof(SaItem);
constexpr size_t kSpillOff = kItemsOff + kItemsBytes;
constexpr unsigned kMaxGrid = 2048;                                    // workgroups of one launch: eight per CU of a 256-CU chip
constexpr unsigned kSpillRows = SA_LP;                                 // direction rows per workgroup, 1 KiB each
constexpr size_t kSpillBytes = (size_t)kMaxGrid * kSpillRows * 1024;
constexpr size_t kScratchBytes = kSpillOff + kSpillBytes;
static_assert(kTableOff + kTableBytes <= kItemsOff, "table overlaps the work list");
static_assert(kScratchBytes <= SA_NOALLOC_SCRATCH_BYTES, "sa_hip.h states a smaller scratch size");

}  // namespace

// The scratch buffer.  In host code this name is only a handle for hipGetSymbolAddress; it is not a
// device address and is never passed to a kernel or a memory routine.
static __device__ __align__(256) unsigned char sa_scratch[kScratchBytes];

namespace {

struct SetupArgs {
    int32_t S[400];
    int32_t gap;
    uint32_t nseq;
    uint32_t stride;
};

// status words, uint32 index into the head area
enum { kStatBadLen = 4, kStatBadCode = 5, kStatMaxLen = 6 };

// One pass over the inputs and the call's tables: validates lengths (1 .. 64, record body fits the
// stride) and residue codes (0 .. 19 inside a sequence), writes the shifted table and the work list.
__global__ void __launch_bounds__(256) sa_noalloc_setup_kernel(const SetupArgs A, const uint8_t* __restrict__ slots,
                                                               const uint16_t* __restrict__ lens,
                                                               unsigned char* __restrict__ scratch) {
    uint32_t* const head = reinterpret_cast<uint32_t*>(scratch);
    int32_t* const table = reinterpret_cast<int32_t*>(scratch + kTableOff);
    SaItem* const items = reinterpret_cast<SaItem*>(scratch + kItemsOff);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = gridDim.x * blockDim.x;
    for (uint32_t e = t; e < 21 * SA_SPRIME_STRIDE; e += nthreads) {
        const uint32_t a = e / SA_SPRIME_STRIDE, b = e % SA_SPRIME_STRIDE;
        table[e] = (a < 20 && b < 20) ? A.S[a * 20 + b] - 2 * A.gap : 0;
    }
    const uint32_t npairs = A.nseq - 1;
    for (uint32_t k = t; k < (npairs + 63) / 64; k += nthreads) {
        SaItem it;
        it.p = k * 64;
        it.v0 = 0;
        it.count = min(64u, npairs - k * 64);
        it.b0 = it.c0 = 0;
        it.reserved[0] = it.reserved[1] = it.reserved[2] = 0;
        items[k] = it;
    }
    for (uint32_t s = t; s < A.nseq; s += nthreads) {
        const uint32_t len = lens[s];
        if (len == 0 || len > SA_LP || SA_REC_HEADER_BYTES + len > A.stride) { atomicOr(&head[kStatBadLen], 1u); continue; }
        atomicMax(&head[kStatMaxLen], len);
        const uint8_t* slot = slots + (size_t)s * SA_LP;
        uint32_t worst = 0;
        for (uint32_t i = 0; i < len; i++) worst = max(worst, (uint32_t)slot[i]);
        if (worst >= 20) atomicOr(&head[kStatBadCode], 1u);
    }
}

std::mutex g_mu;
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices];   // per device: where that GPU's copy of sa_scratch lives
int g_ncu[kMaxDevices];

inline int hip_code(hipError_t e) { return e == hipSuccess ? SA_OK : SA_ERR_HIP - (int)e; }
#define SA_HIP_TRY(expr)                           \
    do {                                           \
        hipError_t e_ = (expr);                    \
        if (e_ != hipSuccess) return hip_code(e_); \
    } while (0)

}  // namespace

extern "C" int sa_align_adjacent_noalloc(int device, const uint8_t* d_slots, const uint16_t* d_lens, uint64_t nseq,
                                         const int32_t* S, int32_t gap, uint8_t* d_records, uint32_t stride) {
    if (!d_slots || !d_lens || !d_records) return SA_ERR_ARG;
    if (device < 0 || device >= kMaxDevices) return SA_ERR_ARG;
    if (nseq < 2) return SA_ERR_ARG;
    if (nseq - 1 > SA_NOALLOC_MAX_PAIRS) return SA_ERR_NOALLOC_SIZE;
    if (stride < 16 || stride > SA_LDS_STAGE_DWORDS * 4 || (stride & 15)) return SA_ERR_ARG;
    SetupArgs A;
    if (S) memcpy(A.S, S, sizeof(A.S)); else sa_default_matrix(A.S);
    A.gap = gap;
    A.nseq = (uint32_t)nseq;
    A.stride = stride;
    {
        // same bound as sa_init: 4 * D + tag must fit 32 bits
        int64_t top = 0;
        for (int k = 0; k < 400; k++) {
            const int64_t sp = (int64_t)A.S[k] - 2 * (int64_t)gap;
            top = sp < 0 ? (top > -sp ? top : -sp) : (top > sp ? top : sp);
        }
        if ((int64_t)SA_LP * top >= (1ll << 29)) return SA_ERR_ARG;
    }
    int ndev = 0;
    SA_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device >= ndev) return SA_ERR_ARG;
    // the caller's current device is put back on every way out
    struct DeviceGuard {
        int previous = -1;
        ~DeviceGuard() { if (previous >= 0) (void)hipSetDevice(previous); }
    } guard;
    SA_HIP_TRY(hipGetDevice(&guard.previous));
    SA_HIP_TRY(hipSetDevice(device));
    unsigned char* scratch = nullptr;
    int ncu = 0;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_scratch_ptr[device]) {
            // once per device, with that device current: this GPU's address of the array
            unsigned char* p = nullptr;
            SA_HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(sa_scratch)));
            if (!p) return SA_ERR_HIP - (int)hipErrorInvalidSymbol;
            int cus = 0;
            SA_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            g_ncu[device] = cus;
            g_scratch_ptr[device] = p;
        }
        scratch = g_scratch_ptr[device];
        ncu = g_ncu[device];
    }
    hipStream_t stream = nullptr;     // the null stream: nothing to create
    const uint32_t npairs = (uint32_t)(nseq - 1);
    const uint32_t nitems = (npairs + 63) / 64;

    // queue head and status words start at zero (they hold the last call's values)
    SA_HIP_TRY(hipMemsetAsync(scratch, 0, kHeadBytes, stream));
    const unsigned setup_grid = (unsigned)std::min<uint64_t>(1024, (nseq + 255) / 256);
    hipLaunchKernelGGL(sa_noalloc_setup_kernel, dim3(setup_grid), dim3(256), 0, stream, A, d_slots, d_lens, scratch);
    SA_HIP_TRY(hipGetLastError());
    uint32_t status[4] = {0, 0, 0, 0};    // bad length, bad code, longest sequence
    SA_HIP_TRY(hipMemcpyAsync(status, scratch + kStatBadLen * 4, sizeof(status), hipMemcpyDeviceToHost, stream));
    SA_HIP_TRY(hipStreamSynchronize(stream));
    if (status[0]) return SA_ERR_LENGTH;
    if (status[1]) return SA_ERR_ALPHABET;

    SaPepParams P;
    memset(&P, 0, sizeof(P));
    P.codes_fo = d_slots;
    P.len_fo = d_lens;
    P.sprime = reinterpret_cast<const int32_t*>(scratch + kTableOff);
    P.items = reinterpret_cast<const SaItem*>(scratch + kItemsOff);
    P.nitems = nitems;
    P.counter = reinterpret_cast<uint32_t*>(scratch);
    P.out = d_records;
    P.stride = stride;
    P.out_base_pair = 0;
    P.spill = reinterpret_cast<uint32_t*>(scratch + kSpillOff);
    P.spill_rows = kSpillRows;
    P.nseq = (uint32_t)nseq;
    P.gap = gap;
    P.lslot = SA_LP;
    // two waves per SIMD of the ADJACENT kernel reside: eight workgroups per CU, as sa_submit launches it; never more
    // than the scratch array has direction rows for
    const unsigned grid = std::min<unsigned>(kMaxGrid, std::min<unsigned>((unsigned)std::max(1, ncu) * 8u, nitems));
    sa_launch_peptide(1 /* ADJACENT */, 2, &P, grid, SA_LDS_FIXED_BYTES, stream);
    SA_HIP_TRY(hipGetLastError());
    SA_HIP_TRY(hipStreamSynchronize(stream));
    return SA_OK;
}
