// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on the access widths the alignment kernels use:
// every wave reads (or writes) consecutive lanes' W dwords, W = 1..4, once over a 2 GiB buffer (well beyond the
// 256 MiB Infinity Cache).  Known bytes per launch = 2 GiB; compare with the counters per kernel name.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int W> struct Vec { uint32_t v[W]; };

template <int W>
__global__ void calib_read(const uint32_t* __restrict__ src, uint64_t nvec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * blockDim.x) {
        const Vec<W> x = *reinterpret_cast<const Vec<W>*>(src + i * W);
#pragma unroll
        for (int k = 0; k < W; k++) acc ^= x.v[k];
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;     // keeps the loads alive; the buffer holds zeros
}

template <int W>
__global__ void calib_write(uint32_t* __restrict__ dst, uint64_t nvec) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * blockDim.x) {
        Vec<W> x;
#pragma unroll
        for (int k = 0; k < W; k++) x.v[k] = (uint32_t)i + k;
        *reinterpret_cast<Vec<W>*>(dst + i * W) = x;
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const uint64_t bytes = 2ull << 30;
    uint32_t *buf, *sink;
    CHECK(hipMalloc(&buf, bytes + 64));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    const uint64_t dwords = bytes / 4;
    calib_read<1><<<grid, block>>>(buf, dwords / 1, sink);
    calib_read<2><<<grid, block>>>(buf, dwords / 2, sink);
    calib_read<3><<<grid, block>>>(buf, dwords / 3, sink);
    calib_read<4><<<grid, block>>>(buf, dwords / 4, sink);
    calib_write<1><<<grid, block>>>(buf, dwords / 1);
    calib_write<2><<<grid, block>>>(buf, dwords / 2);
    calib_write<3><<<grid, block>>>(buf, dwords / 3);
    calib_write<4><<<grid, block>>>(buf, dwords / 4);
    CHECK(hipDeviceSynchronize());
    printf("each kernel moved %llu bytes (%.1f KiB)\n", (unsigned long long)bytes, bytes / 1024.0);
    return 0;
}
