// Micro-benchmark: cost of fetching random 128-byte records with different lane-to-record mappings.
//   mode 0: one record per lane, 8 x dwordx4 (each instruction touches 64 different lines)
//   mode 1: two lanes per record, 4 x dwordx4 each (32 lines per instruction)
//   mode 2: four lanes per record, 2 x dwordx4 each (16 half-lines per instruction)
//   mode 3: eight lanes per record, 1 x dwordx4 each (8 lines per instruction, fully coalesced)
// Every mode fetches the same number of records per wave; prints records per microsecond (whole GPU).
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o tools/microbench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ void __launch_bounds__(64) gather(const uint4 *rec, uint32_t n_rec, int iters, uint32_t *out) {
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x;
    constexpr int LPR = MODE == 0 ? 1 : MODE == 1 ? 2 : MODE == 2 ? 4 : 8; // lanes per record
    constexpr int NLD = 8 / LPR;                                          // dwordx4 loads per lane
    constexpr int PASSES = LPR;                                           // passes to cover 64 records
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            const uint32_t slot = (uint32_t)(lane / LPR) + (uint32_t)p * (64 / LPR); // 0..63: which of the 64 records
            const uint32_t r = hash32(wave * 1000003u + (uint32_t)it * 64u + slot) % n_rec;
            const uint4 *base = rec + (size_t)r * 8 + (size_t)(lane % LPR) * NLD;
            uint4 v[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) v[k] = base[k];
#pragma unroll
            for (int k = 0; k < NLD; k++) acc ^= v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_rec = argc > 1 ? (uint32_t)atoi(argv[1]) : 75609u;
    const int iters = 400;
    uint4 *rec; uint32_t *out;
    CHECK(hipMalloc(&rec, (size_t)n_rec * 128));
    CHECK(hipMemset(rec, 1, (size_t)n_rec * 128));
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int waves = 256 * 8;
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            switch (mode) {
                case 0: gather<0><<<waves, 64>>>(rec, n_rec, iters, out); break;
                case 1: gather<1><<<waves, 64>>>(rec, n_rec, iters, out); break;
                case 2: gather<2><<<waves, 64>>>(rec, n_rec, iters, out); break;
                default: gather<3><<<waves, 64>>>(rec, n_rec, iters, out); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                const double recs = (double)waves * iters * 64;
                printf("mode %d (%d lanes/record): %.3f ms, %.1f records/us, %.1f GB/s, %.2f cycles/record/CU at 2.4 GHz\n", mode,
                       mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 4 : 8, ms, recs / (ms * 1e3), recs * 128 / (ms * 1e6), ms * 1e-3 * 2.4e9 * 256 / recs);
            }
        }
    }
    return 0;
}
