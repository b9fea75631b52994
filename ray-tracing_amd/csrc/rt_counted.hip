/*
 * rt_counted.hip - the render kernels' variants with work counters (RT_F_COUNTERS), in a translation unit of their own.
 * They are diagnostic kernels: what they count does not depend on how their instructions are ordered. This file is compiled
 * without -amdgpu-sched-strategy=iterative-ilp (csrc/Makefile): with that option the compiler's register allocator has
 * crashed on these variants - they carry the counters of every lane, ~60 more live registers than the production kernels
 * in the same register budget - after changes that had nothing to do with them.
 */
#include "rt_render_kernels.hip.h"

template __global__ void rt_render_kernel<true, false>(DScene, RenderArgs, int, int, int, double *, double *, rt_counters *, int *, int);
template __global__ void rt_render_spheres_kernel<true>(DScene, RenderArgs, int, int, int, double *, double *, rt_counters *, int *, int);
