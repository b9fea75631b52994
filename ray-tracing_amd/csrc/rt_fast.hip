/*
 * rt_fast.hip - the single-precision render kernel (rt_params.flags & RT_F_FAST) and its launch, in a translation unit of
 * its own: it is compiled with FMA contraction allowed (csrc/Makefile), which the double-precision kernels of rt_capi.hip
 * must not be.
 */
#include "rt_render_kernels.hip.h"
#include "rt_fast.hip.h"

using namespace rtf;

#ifndef RT_FAST_WAVES_PER_SIMD
#define RT_FAST_WAVES_PER_SIMD 3 /* resident waves per SIMD the kernel is compiled for (register budget 512 / this) */
#endif

__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_FAST_WAVES_PER_SIMD, RT_FAST_WAVES_PER_SIMD)))
rt_render_fast_kernel(DScene sc, RenderArgs ra, const FastTri *ftris, int k0, int nk, int chunk, double *accum, double *unit_sums, int *next_slot, int n_slots) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FastLDS *w = reinterpret_cast<FastLDS *>(lds_raw);
    /* the sphere list (at most RT_LDS_SPHERES_MAX: checked by the host) behind the node stack */
    unsigned char *p = const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(fast_lds_spheres(w, sc.node_cap)));
    {
        const int n16 = sc.n_spheres * (int)(sizeof(rt_sphere) / 16);
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.spheres);
        uint4 *d16 = reinterpret_cast<uint4 *>(p);
        for (int q = threadIdx.x; q < n16; q += RT_BLOCK) d16[q] = src[q];
    }
    if (threadIdx.x == 0) { w->sc = sc; w->ra = ra; }
    wave_sync();
    const rt_sphere __attribute__((address_space(3))) *sph_lds = (const rt_sphere __attribute__((address_space(3))) *)reinterpret_cast<const rt_sphere *>(p);
    fast_render_persistent(sc, w->sc, w, sph_lds, ftris, w->ra, k0, nk, chunk, accum, unit_sums, next_slot, n_slots);
}

/* the filter tree's triangles (rt_tri, doubles) rounded to float; orig_id there = position in the reference's order */
__global__ void rt_fast_tris_kernel(const rt_tri *own_tris, uint32_t n, FastTri *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const rt_tri t = own_tris[k];
    FastTri f;
    for (int a = 0; a < 3; a++) { f.A[a] = (float)t.A[a]; f.e1[a] = (float)t.e1[a]; f.e2[a] = (float)t.e2[a]; }
    f.pos = t.orig_id;
    out[k] = f;
}

size_t rt_fast_tri_bytes(void) { return sizeof(FastTri); }

size_t rt_fast_lds_bytes(int node_cap, int n_spheres) { return sizeof(FastLDS) + 16 + (size_t)node_cap * 8 + (size_t)n_spheres * sizeof(rt_sphere); }

int rt_fast_waves_per_cu(size_t lds_bytes) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_render_fast_kernel, RT_BLOCK, lds_bytes) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
    const int by_regs = 4 * RT_FAST_WAVES_PER_SIMD;
    return per_cu < by_regs ? per_cu : by_regs;
}

hipError_t rt_fast_convert_tris(const rt_tri *own_tris, uint32_t n, void *out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(rt_fast_tris_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, own_tris, n, reinterpret_cast<FastTri *>(out));
    return hipGetLastError();
}

hipError_t rt_fast_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const DScene &sc, const void *ftris, const RenderArgs &ra, int k0, int nk, int chunk,
                          double *accum, double *unit_sums, int *next_slot, int n_slots) {
    hipLaunchKernelGGL(rt_render_fast_kernel, dim3(grid), dim3(RT_BLOCK), lds_bytes, stream, sc, ra, reinterpret_cast<const FastTri *>(ftris), k0, nk, chunk, accum,
                       unit_sums, next_slot, n_slots);
    return hipGetLastError();
}
