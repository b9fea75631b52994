// Calibration of the FETCH_SIZE counter (rocprofv3 --pmc FETCH_SIZE) for the access shapes of the render kernel: every mode
// reads a known number of bytes from a table far larger than the caches (L2 32 MiB, Infinity Cache 256 MiB), each record
// (practically) once, so the bytes that must cross the L2's memory side are known and the counter can be compared with them.
//   mode 0: streaming read, 16 bytes per lane, coalesced           (the shape the x2 correction of gfx950 is established for)
//   mode 1: random 128-byte records, one per lane (8 x 16 B)       (a filter-tree record in a full-width node step)
//   mode 2: random 64-byte half records, one per lane (4 x 16 B)   (a reference-tree record; two lanes per filter-tree record)
//   mode 3: random 32-byte pieces of 128-byte records              (four lanes per filter-tree record: 24 + 4 bytes each)
//   mode 4: random 80-byte records at an 80-byte stride            (a triangle: most straddle two 64-byte halves, a third two lines)
// usage: calib_fetch <mode> [table MiB, default 8192]; prints the bytes the lanes asked for.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) calib_stream(const uint4 *t, uint64_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = t[k];
        acc ^= v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// BYTES per lane and access, at a stride of STRIDE bytes, from records picked at random (a permutation-like hash: few repeats)
template <int BYTES, int STRIDE>
__global__ void __launch_bounds__(64) calib_gather(const unsigned char *t, uint64_t n_rec, int iters, uint32_t *out) {
    uint32_t acc = 0;
    const uint64_t lane_id = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        const uint64_t r = mix(lane_id * 1000003ull + (uint64_t)it) % n_rec;
        const unsigned char *p = t + r * STRIDE;
        if (BYTES % 16 == 0 && STRIDE % 16 == 0) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
            for (int k = 0; k < BYTES / 16; k++) { const uint4 v = q[k]; acc ^= v.x + v.y + v.z + v.w; }
        } else {
            const uint2 *q = reinterpret_cast<const uint2 *>(p);
#pragma unroll
            for (int k = 0; k < BYTES / 8; k++) { const uint2 v = q[k]; acc ^= v.x + v.y; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint64_t mib = argc > 2 ? (uint64_t)atoll(argv[2]) : 8192ull;
    const uint64_t bytes = mib << 20;
    unsigned char *t; uint32_t *out;
    CHECK(hipMalloc(&t, bytes));
    CHECK(hipMemset(t, 1, bytes));
    CHECK(hipMalloc(&out, 4));
    CHECK(hipDeviceSynchronize());
    const int waves = 256 * 8 * 4, iters = 256;
    const double accesses = (double)waves * 64 * iters;
    double asked = 0;
    switch (mode) {
        case 0: calib_stream<<<256 * 16, 256>>>((const uint4 *)t, bytes / 16, out); asked = (double)bytes; break;
        case 1: calib_gather<128, 128><<<waves, 64>>>(t, bytes / 128, iters, out); asked = accesses * 128; break;
        case 2: calib_gather<64, 64><<<waves, 64>>>(t, bytes / 64, iters, out); asked = accesses * 64; break;
        case 3: calib_gather<32, 32><<<waves, 64>>>(t, bytes / 32, iters, out); asked = accesses * 32; break;
        default: calib_gather<80, 80><<<waves, 64>>>(t, bytes / 80, iters, out); asked = accesses * 80; break;
    }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    printf("{\"mode\": %d, \"table_MiB\": %llu, \"accesses\": %.0f, \"bytes_asked\": %.0f}\n", mode, (unsigned long long)mib, mode ? accesses : 0.0, asked);
    return 0;
}
