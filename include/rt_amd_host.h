/*
 * rt_amd_host.h - C wrappers over the host-side scene classes (ray-tracing_amd/host/rt_host.hpp)
 * for callers that cannot include C++ (Python ctypes, C). They replace the scene-building part of
 * the reference's main() (raytracer.cpp:1163-1435): TriangleMesh::readOBJ (:293), loadTexture
 * (:759), the transforms (:768-860), buildBVH (:244) and Scene::objects.push_back (:1401-1435).
 * All functions return 0 on success and a negative value on failure; rth_last_error() gives the
 * message (thread-local).
 */
#ifndef RT_AMD_HOST_H
#define RT_AMD_HOST_H
#include <stdint.h>

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_mesh rth_mesh;
typedef struct rth_scene rth_scene;

const char *rth_last_error(void);

/* TriangleMesh(albedo, mirror, transp), raytracer.cpp:212 */
rth_mesh *rth_mesh_create(const double albedo[3], int mirror, int transp);
void rth_mesh_destroy(rth_mesh *m);
/* TriangleMesh::readOBJ, raytracer.cpp:293 (a missing file is an error, not a crash) */
int rth_mesh_read_obj(rth_mesh *m, const char *path);
/* TriangleMesh::loadTexture, raytracer.cpp:759: JPEG files (baseline and progressive; the texels stbi_load(..., 3) of the
 * reference's stb_image v2.26 gives, raytracer.cpp:762) and binary PPM files */
int rth_mesh_load_texture(rth_mesh *m, const char *path);
/* texture k of the mesh as loadTexture / addTexture stored it: size, and (rgb != NULL) its W*H*3 bytes */
int rth_mesh_texture(const rth_mesh *m, int32_t k, int32_t *W, int32_t *H, uint8_t *rgb);
/* the JPEG decoder of loadTexture on a file image in memory; rgb (may be NULL to ask for the size) receives W*H*3 bytes */
int rth_decode_jpeg(const uint8_t *file, int64_t n, int32_t *W, int32_t *H, uint8_t *rgb, int64_t rgb_cap);
/* a decoded RGB8 image, rows top to bottom, as stbi_load(...,3) returns it (raytracer.cpp:762) */
int rth_mesh_add_texture(rth_mesh *m, const uint8_t *rgb, int32_t W, int32_t H);
/* fill vertices / normals / uvs (2 per uv) / TriangleIndices (10 ints each, member order
 * vtxi,vtxj,vtxk, uvi,uvj,uvk, ni,nj,nk, group) directly */
int rth_mesh_set_raw(rth_mesh *m, int64_t nv, const double *v, int64_t nn, const double *n, int64_t nuv, const double *uv, int64_t nt,
                     const int32_t *idx10);
/* op: 0 invertNormals() :768, 1 translate(a,b) :776, 2 rotateNormals(a,b) :795, 3 rotate(a,b) :802,
 *     4 rotateAroundAxe(a, angle) :810, 5 zoom(b) :840 */
int rth_mesh_op(rth_mesh *m, int op, int a, int b, double angle);
/* buildBVH(BVH, 0, indices.size()), raytracer.cpp:244, :1385 */
int rth_mesh_build_bvh(rth_mesh *m);
/* the same on the GPU of ctx (rt_build_bvh of rt_amd.h): identical boxes, tree and triangle order */
int rth_mesh_build_bvh_device(rth_mesh *m, rt_ctx *ctx);
int rth_mesh_sizes(const rth_mesh *m, int64_t *nv, int64_t *nn, int64_t *nuv, int64_t *nt, int32_t *ntex, int64_t *n_nodes, int32_t *max_depth);
/* current arrays (any may be NULL); perm[k] = face-fan index of the triangle now at position k */
int rth_mesh_get(const rth_mesh *m, double *v, double *n, double *uv, int32_t *idx10, int32_t *perm);
/* preorder (fg before fd) dump of the tree: boxes = 6 doubles per node, dfl = debut, fin, is_leaf */
int64_t rth_mesh_bvh_dump(const rth_mesh *m, double *boxes, int32_t *dfl, int64_t cap);

/* Scene, raytracer.cpp:923-929 */
rth_scene *rth_scene_create(double I, const double L[3]);
void rth_scene_destroy(rth_scene *s);
/* both return the objectId = position in Scene::objects */
int rth_scene_add_sphere(rth_scene *s, const double O[3], double R, const double albedo[3], int mirror, int transp);
int rth_scene_add_mesh(rth_scene *s, rth_mesh *m); /* borrowed: the mesh must outlive the scene */
/* flatten + rt_upload_scene; flat_bytes (may be NULL) = bytes copied to HBM */
int rth_scene_upload(rth_scene *s, rt_ctx *ctx, int64_t *flat_bytes);
/* rt::render on several GPUs from one process (one host thread and one context per entry of `devices`; a device may be
 * listed twice): the p->spp samples of every pixel are split over the entries, summed on the first (rt_accum_add) and
 * tone-mapped there; rgb8 receives W*H*3 bytes. Replaces the OpenMP pixel loop raytracer.cpp:1456-1503. */
int rth_scene_render_devices(rth_scene *s, const rt_camera *cam, const rt_params *p, const int32_t *devices, int32_t n_devices, uint8_t *rgb8);
int rth_scene_flat_info(rth_scene *s, int64_t *n_spheres, int64_t *n_meshes, int64_t *n_nodes, int64_t *n_tris, int64_t *bytes);
/* The flattened scene itself - the arrays of rt_scene_desc (rt_amd.h) exactly as rth_scene_upload hands them to
 * rt_upload_scene: spheres and meshes in object order (raytracer.cpp:1401-1435), one rt_bvh_node per inner Noeud
 * (:190-196), triangles and attributes in the reference's visiting order (:656-670), the textures' texels as
 * stbi_load returned them (:762). Each pointer may be NULL; the others receive the counts of rth_scene_flat_info
 * (n_tex textures, texel_bytes bytes of texels: call once with NULL arrays to size them). */
int rth_scene_flat_arrays(rth_scene *s, rt_sphere *spheres, rt_mesh *meshes, rt_bvh_node *nodes, rt_tri *tris, rt_tri_attr *attrs, rt_texture *tex,
                          uint8_t *texels, int64_t *n_tex, int64_t *texel_bytes);

/* camera basis as main() builds it (raytracer.cpp:1441-1453) and the literals of the render loop */
int rth_camera_make(const double C[3], double fov_deg, double angle_v_deg, double angle_h_deg, int32_t W, double focus, double aa_sigma,
                    double dof_sigma, rt_camera *out);
int rth_params_default(int32_t W, int32_t H, int32_t spp, rt_params *out);

/* the output step of main(), raytracer.cpp:1505-1512: the 8-bit image (rows top to bottom, as rt_resolve returns it) as a
 * PNG file (stbi_write_png in the reference; same pixels), and the whole seconds of a duration as execution_time.txt has them */
int rth_write_png(const char *path, int32_t W, int32_t H, const uint8_t *rgb);
int rth_write_execution_time(const char *path, double seconds);

#ifdef __cplusplus
}
#endif
#endif
