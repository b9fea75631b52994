/*
 * ref_cpu.h - CPU oracle: a clean-room FP64 restatement of the reference's
 * per-pixel path-tracing hot path (raytracer.cpp, cited per function in
 * ref_cpu.cpp).
 *
 * TEST INFRASTRUCTURE ONLY. This library is the checker that the GPU path is
 * compared against and the CPU baseline that bench.py times; nothing in the
 * product (ray-tracing_amd/, include/) links, loads or calls it.
 *
 * Parity status: pinned. tests/test_oracle_vs_reference.py drives this oracle
 * and the real reference classes (oracle/_ref, compiled from
 * /root/reference/raytracer.cpp) on the same scenes, rays and injected random
 * streams and requires bit-identical results; tests/golden/ holds outputs of
 * the real reference for boxes where it is absent.
 */
#ifndef REF_CPU_H
#define REF_CPU_H
#include <stdint.h>
#include "../include/rt_amd.h" /* rt_camera, rt_params, rt_counters PODs and RT_F_* flags only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

orc_scene *orc_scene_create(double I, const double L[3]);
void orc_scene_destroy(orc_scene *s);
/* returns the objectId (position in Scene::objects) */
int orc_add_sphere(orc_scene *s, const double O[3], double R, const double albedo[3], int mirror,
                   int transp);
/* tri_idx: n_tris * 10 ints in TriangleIndices member order
 * (vtxi,vtxj,vtxk, uvi,uvj,uvk, ni,nj,nk, group). textures: n_tex RGB8
 * arrays (borrowed, must outlive the scene). Builds the BVH. */
int orc_add_mesh(orc_scene *s, int64_t n_verts, const double *verts, int64_t n_normals,
                 const double *normals, int64_t n_uvs, const double *uvs /* n_uvs*2 */,
                 int64_t n_tris, const int32_t *tri_idx, const double albedo[3], int mirror,
                 int transp, int n_tex, const uint8_t *const *tex, const int32_t *texW,
                 const int32_t *texH);

/* BVH facts of mesh object `object_id`: perm[k] = original index of the
 * triangle at permuted position k (n_tris ints, may be NULL). */
int orc_mesh_bvh_info(const orc_scene *s, int object_id, int32_t *perm, int64_t *n_nodes,
                      int64_t *n_leaves, int32_t *max_depth, int32_t *max_leaf);
/* Preorder dump of the tree: for node k: box (6 doubles), debut, fin, is_leaf. */
int64_t orc_mesh_bvh_dump(const orc_scene *s, int object_id, double *boxes, int32_t *debut_fin_leaf,
                          int64_t cap_nodes);

/* Scene::intersect on n rays (6 doubles each). Any output may be NULL. */
void orc_intersect_rays(const orc_scene *s, const double *rays, int64_t n, int32_t *obj_id,
                        int32_t *tri_id, double *t, double *PN, double *albedo,
                        rt_counters *counters);

/* radiance of single samples (i,j,k): ijk = n*3 ints, out = n*3 doubles */
void orc_sample_radiance(const orc_scene *s, const rt_camera *cam, const rt_params *p,
                         const int32_t *ijk, int64_t n, double *out);

/* the same with the path each sample took: trace = n * RT_TRACE_LEN * 4 ints, laid out and coded as rt_sample_trace of
 * rt_amd.h describes (may be NULL) */
void orc_sample_trace(const orc_scene *s, const rt_camera *cam, const rt_params *p, const int32_t *ijk,
                      int64_t n, double *out, int32_t *trace);

/* cos / sin / log of the sampling code: 0 = host libm (as the reference; the default), 1 = ray-tracing_amd/csrc/rt_math.h
 * (what the GPU computes with). Process-wide; set between calls. */
void orc_set_math(int portable);
int orc_get_math(void);
/* rt_math.h's functions themselves, for accuracy tests */
void orc_math_sincos(const double *x, int64_t n, double *sn, double *cs);
void orc_math_log(const double *x, int64_t n, double *out);

/* pixel loop: adds p->spp samples per pixel of p->rect into accum
 * (W*H*3 doubles, reference row order). n_threads <= 0: all OpenMP threads. */
void orc_render(const orc_scene *s, const rt_camera *cam, const rt_params *p, double *accum,
                int n_threads, rt_counters *counters);

/* raytracer.cpp:1488-1501 */
void orc_resolve(const double *accum, int32_t W, int32_t H, double spp_total, uint8_t *rgb8);

/* the counter-based random stream, exposed for known-answer tests */
uint64_t orc_rng_key(uint64_t seed, uint32_t pixel, uint32_t sample);
uint64_t orc_rng_draw(uint64_t key, uint32_t n);
double orc_rng_uniform(uint64_t key, uint32_t n);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
