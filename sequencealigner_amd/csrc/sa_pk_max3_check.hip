// sa_pk_max3_check.hip -- the property the paired peptide kernel rests on: does the packed three-way half-precision maximum order 16-bit patterns below
// 0x7c00 (finite, non-negative halves, subnormals included) exactly as unsigned integers, and return
// the winning operand bit for bit?  Prints the number of mismatches over structured and random triples.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void check(unsigned long long* bad, uint32_t* first_bad, uint32_t rounds) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t v[3];
        const uint32_t s = mix(tid * 2654435761u + r * 40503u + 1);
        for (int k = 0; k < 3; k++) {
            uint32_t lo, hi;
            const uint32_t z = mix(s + k * 977u);
            if ((r & 3) == 0) { lo = z % 0x7c00u; hi = (z >> 16) % 0x7c00u; }                       // anywhere in range
            else if ((r & 3) == 1) { lo = (tid + (z & 3)) % 0x7c00u; hi = (tid * 7 + ((z >> 2) & 3)) % 0x7c00u; }  // near ties
            else if ((r & 3) == 2) { lo = z % 0x0410u; hi = (z >> 16) % 0x0410u; }                  // subnormal range and its edge
            else { lo = 0x7bffu - z % 64; hi = (z >> 16) % 6000u; }                                  // top of the range / DP-sized values
            v[k] = lo | (hi << 16);
        }
        uint32_t got;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(got) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
        uint32_t wl = 0, wh = 0;
        for (int k = 0; k < 3; k++) { wl = max(wl, v[k] & 0xffffu); wh = max(wh, v[k] >> 16); }
        if (got != (wl | (wh << 16))) { if (n == 0) { first_bad[0] = v[0]; first_bad[1] = v[1]; first_bad[2] = v[2]; first_bad[3] = got; } n++; }
    }
    if (n) atomicAdd(bad, n);
}

#define TRY(x) do { if ((x) != hipSuccess) { printf("%s failed\n", #x); return 2; } } while (0)

int main() {
    unsigned long long* d_bad; uint32_t* d_first;
    TRY(hipMalloc(&d_bad, 8)); TRY(hipMalloc(&d_first, 16)); TRY(hipMemset(d_bad, 0, 8)); TRY(hipMemset(d_first, 0, 16));
    const uint32_t blocks = 4096, rounds = 1024;
    check<<<blocks, 256>>>(d_bad, d_first, rounds);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    unsigned long long bad; uint32_t first[4];
    TRY(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); TRY(hipMemcpy(first, d_first, 16, hipMemcpyDeviceToHost));
    printf("triples %llu mismatches %llu", (unsigned long long)blocks * 256 * rounds, bad);
    if (bad) printf("  e.g. %08x %08x %08x -> %08x", first[0], first[1], first[2], first[3]);
    printf("\n");
    return bad ? 1 : 0;
}
