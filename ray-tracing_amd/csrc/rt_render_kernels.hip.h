/*
 * rt_render_kernels.hip.h - the render kernels of the gfx950 path tracer (the persistent waves of rt_device.hip.h behind
 * a __global__ entry each), shared by the two translation units that instantiate them: rt_capi.hip (the production
 * variants and everything else) and rt_counted.hip (the variants with work counters, see there).
 */
#ifndef RT_RENDER_KERNELS_HIP_H
#define RT_RENDER_KERNELS_HIP_H

#include <hip/hip_runtime.h>

#include "rt_amd.h"
#include "rt_device.hip.h"

using namespace rtd;

#define RT_COUNTER_WORDS 32 /* rt_counters (8 words) + 23 diagnostic words of rt_device.hip.h (+ 1 spare) */
#define RT_BLOCK 64 /* one workgroup = one wave: the LDS pool of rt_device.hip.h is per wave */
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 2 /* resident waves per SIMD the render kernel is compiled for (register budget 512 / this) */
#endif
#define RT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_SIMD, RT_WAVES_PER_SIMD)))

/* LDS of the wave; sphere lists of up to RT_LDS_SPHERES_MAX entries are staged behind the node stack
 * (every ray tests every sphere: raytracer.cpp:934-938), longer lists are read from global memory
 * with wave-uniform addresses. Mesh root records are read with wave-uniform addresses too (scalar loads). */
__device__ __forceinline__ WaveLDS *wave_setup(const DScene &sc, unsigned char *lds_raw, const rt_sphere **sph, const RootRec **roots) {
    WaveLDS *w = reinterpret_cast<WaveLDS *>(lds_raw);
    unsigned char *p = const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(lds_spheres(w, sc.node_cap)));
    *sph = sc.spheres;
    *roots = sc.roots;
    if (sc.n_spheres <= RT_LDS_SPHERES_MAX) {
        const int n16 = sc.n_spheres * (int)(sizeof(rt_sphere) / 16);
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.spheres);
        uint4 *d16 = reinterpret_cast<uint4 *>(p);
        for (int q = threadIdx.x; q < n16; q += RT_BLOCK) d16[q] = src[q];
        *sph = reinterpret_cast<const rt_sphere *>(p);
    }
#if RT_TOP_LDS
    if (sc.own_rec4 != nullptr && sc.n_own_nodes >= 5u) {
        const float *src = reinterpret_cast<const float *>(sc.own_rec4 + 1);
        float *dst = &w->top[0][0];
        for (int q = threadIdx.x; q < 4 * 32; q += RT_BLOCK) dst[q] = src[q];
    }
#endif
    wave_sync();
    return w;
}

/* CNT: with work counters. PLAIN: no mode flag of rt_params is set (what a production render uses): the other modes'
 * code is left out. PREGEN: the launch's camera rays are read from ra.cam_rays (rt_camera_rays_kernel made them). */
template <bool CNT, bool PLAIN = false, bool PREGEN = false>
__global__ void __launch_bounds__(RT_BLOCK) RT_OCCUPANCY rt_render_kernel(DScene sc, RenderArgs ra, int k0, int nk, int chunk, double *accum, double *unit_sums,
                                                             rt_counters *counters, int *next_slot, int n_slots) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const rt_sphere *sph;
    const RootRec *roots;
    WaveLDS *w = wave_setup(sc, lds_raw, &sph, &roots);
    Cnt cnt = {};
    /* what the queries' inner loops use stays in the kernel arguments (scalar registers); everything else is read from a
     * copy in LDS where it is needed */
    if (threadIdx.x == 0) { w->sc = sc; w->ra = ra; }
    wave_sync();
    /* this kernel is launched for sphere lists that wave_setup has staged in LDS (longer ones have a tree and the other
     * kernel): saying so lets the sphere tests read them with LDS instructions */
    const rt_sphere __attribute__((address_space(3))) *sph_lds = (const rt_sphere __attribute__((address_space(3))) *)sph;
    render_persistent<CNT, false, PLAIN, PREGEN>(sc, w->sc, w, sph_lds, roots, w->ra, k0, nk, chunk, accum, unit_sums, next_slot, n_slots, cnt);
    if (CNT) flush_counters(cnt, counters);
}

/* the same for scenes whose sphere list is too long for LDS: with the walk through the sphere tree compiled in */
template <bool CNT, bool PREGEN = false>
__global__ void __launch_bounds__(RT_BLOCK) RT_OCCUPANCY rt_render_spheres_kernel(DScene sc, RenderArgs ra, int k0, int nk, int chunk, double *accum, double *unit_sums,
                                                                     rt_counters *counters, int *next_slot, int n_slots) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const rt_sphere *sph;
    const RootRec *roots;
    WaveLDS *w = wave_setup(sc, lds_raw, &sph, &roots);
    Cnt cnt = {};
    if (threadIdx.x == 0) { w->sc = sc; w->ra = ra; }
    wave_sync();
    render_persistent<CNT, true, false, PREGEN>(sc, w->sc, w, sph, roots, w->ra, k0, nk, chunk, accum, unit_sums, next_slot, n_slots, cnt);
    if (CNT) flush_counters(cnt, counters);
}


/* The counted variants are instantiated in rt_counted.hip, which is compiled without the instruction scheduler option of
 * the production kernels: with it the compiler's register allocator has crashed on them (they carry ~60 more live
 * registers, the counters of every lane) after harmless changes elsewhere. */
extern template __global__ void rt_render_kernel<true, false, false>(DScene, RenderArgs, int, int, int, double *, double *, rt_counters *, int *, int);
extern template __global__ void rt_render_spheres_kernel<true, false>(DScene, RenderArgs, int, int, int, double *, double *, rt_counters *, int *, int);

/* rt_fast.hip: the single-precision render kernel (RT_F_FAST), compiled with FMA contraction allowed */
size_t rt_fast_tri_bytes(void);
size_t rt_fast_lds_bytes(int node_cap, int n_spheres);
int rt_fast_waves_per_cu(size_t lds_bytes);
hipError_t rt_fast_convert_tris(const rt_tri *own_tris, uint32_t n, void *out, hipStream_t stream);
hipError_t rt_fast_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const rtd::DScene &sc, const void *ftris, const rtd::RenderArgs &ra, int k0, int nk, int chunk,
                          double *accum, double *unit_sums, int *next_slot, int n_slots);

#endif /* RT_RENDER_KERNELS_HIP_H */
