/*
 * rt_amd.h - C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference program has no plugin / FFI boundary: its hot path is reached
 * by direct C++ member calls inside one translation unit
 * (raytracer.cpp:1456-1503 pixel loop -> Scene::getColor :953 ->
 * Scene::intersect :930 -> Sphere::intersect :883 / TriangleMesh::intersect
 * :642).  This header is the boundary a maintainer of that program would bind
 * instead: plain pointers and sizes, int status codes, no C++ or torch types.
 *
 * Conventions
 *  - every function returns 0 on success, a negative RT_E_* code otherwise;
 *    rt_last_error() gives the message of the last failure on that context.
 *  - all real-valued scene data is IEEE binary64: the reference computes in
 *    double everywhere (raytracer.cpp:29-60) and hit identity must match it
 *    bit for bit.
 *  - host arrays handed to rt_upload_scene are copied to HBM before the call
 *    returns; the caller keeps ownership.
 *  - one render in flight per context; a context is bound to one HIP device
 *    and one HIP stream.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID -1  /* bad argument / inconsistent scene description  */
#define RT_E_DEVICE -2   /* HIP runtime failure (message has the HIP error) */
#define RT_E_NOMEM -3    /* device allocation failed                       */
#define RT_E_STATE -4    /* call made in the wrong state (no scene, ...)   */

/* material flags (Object::isMirror / isTransparent, raytracer.cpp:156-157) */
#define RT_MAT_MIRROR 1
#define RT_MAT_TRANSPARENT 2

/* Sphere (raytracer.cpp:874-921). object_id is the position of the sphere in
 * Scene::objects (raytracer.cpp:927): the lowest id wins exact ties in t
 * (strict '<' at raytracer.cpp:938). */
typedef struct rt_sphere {
    double O[3];
    double R;
    double albedo[3];
    int32_t flags;
    int32_t object_id;
} rt_sphere; /* 64 bytes */

/* Flattened BVH node.  One record holds the boxes of BOTH children of one
 * inner Noeud (raytracer.cpp:190-196), because the reference tests fg's box
 * and then fd's box when it pops their parent (raytracer.cpp:656-665).
 * box = {min x,y,z, max x,y,z}.  Child c is an inner node when n? == 0 (then
 * ? is a node index) and a leaf when n? > 0 (then ? is the index of its first
 * triangle in the permuted triangle array, n? triangles). */
typedef struct rt_bvh_node {
    double boxL[6];
    double boxR[6];
    int32_t L, R;
    int32_t nL, nR;
    int32_t pad[4];
} rt_bvh_node; /* 128 bytes */

/* Triangle, in the order buildBVH leaves TriangleMesh::indices
 * (raytracer.cpp:269-278).  A, e1 = B-A, e2 = C-A are what the test at
 * raytracer.cpp:673-677 derives from the three vertices; they do not depend on
 * the ray, so they are formed once on the host with the same two subtractions.
 * orig_id = index of the triangle in OBJ face-fan order (before buildBVH);
 * mesh = index into rt_scene_desc.meshes.
 * ORDER: within a mesh, triangles are stored in the order in which the
 * reference's traversal visits them (it pops fd before fg,
 * raytracer.cpp:656-665: leaves in descending `debut`, ascending i inside a
 * leaf, :670), and meshes are stored in ascending object_id. "First visited
 * wins an exact tie in t" (strict '<' at :690 and :938) is then "the lower
 * array position wins", independent of the order a GPU traverses in. */
typedef struct rt_tri {
    double A[3];
    double e1[3];
    double e2[3];
    int32_t orig_id;
    int32_t mesh;
} rt_tri; /* 80 bytes */

/* Shading attributes of a triangle, same index as rt_tri: the three vertex
 * normals and uvs gathered through TriangleIndices (raytracer.cpp:694, :702)
 * and the usemtl ordinal 'group' that selects the texture (:700-701). */
typedef struct rt_tri_attr {
    double n[3][3];
    double uv[3][2];
    int32_t group;
    int32_t pad;
} rt_tri_attr; /* 128 bytes */

/* TriangleMesh as an Object (raytracer.cpp:208-218, :862-871). The root
 * Noeud's box is tested first (raytracer.cpp:644). root_count > 0 means the
 * root itself is a leaf holding triangles [root, root+root_count). */
typedef struct rt_mesh {
    double root_box[6];
    double albedo[3];
    int32_t flags;
    int32_t object_id;
    int32_t root;
    int32_t root_count;
    int32_t first_tex; /* index of textures[0] of this mesh in rt_scene_desc.tex */
    int32_t n_tex;     /* textures.size(); 0 -> colour = albedo (raytracer.cpp:697,:719) */
    int32_t first_tri; /* range of this mesh in the triangle arrays */
    int32_t n_tris;
    int32_t max_depth; /* deepest leaf, root = 0 */
    int32_t pad;
} rt_mesh; /* 112 bytes */

/* RGB8 texture as stbi_load(...,3) returns it (raytracer.cpp:762): row 0 is
 * the top row, 3 bytes per texel. offset is into rt_scene_desc.texels. */
typedef struct rt_texture {
    int64_t offset;
    int32_t W, H;
} rt_texture;

typedef struct rt_scene_desc {
    int32_t n_spheres;
    int32_t n_meshes;
    const rt_sphere *spheres;
    const rt_mesh *meshes;
    int64_t n_nodes;
    const rt_bvh_node *nodes;
    int64_t n_tris;
    const rt_tri *tris;
    const rt_tri_attr *attrs;
    int32_t n_tex;
    int32_t pad;
    const rt_texture *tex;
    const uint8_t *texels;
    int64_t texel_bytes;
    /* 256-entry table pow(i/255., 2.2) (raytracer.cpp:713-715). The caller
     * fills it with its libm so texture colours match the CPU bit for bit;
     * pass NULL to have the library compute it with the host libm. */
    const double *srgb_lut;
    double I;    /* Scene::I, raytracer.cpp:929, :1164 */
    double L[3]; /* Scene::L, raytracer.cpp:928, :1165 */
} rt_scene_desc;

/* Camera of the pixel loop (raytracer.cpp:1441-1453, :1475-1483).
 * right/up/view are the basis vectors main() builds from the two angles;
 * zdist = W / (2 tan(fov/2)); focus = 55 (:1479); aa_sigma = 0.25 (:1468);
 * dof_sigma = 0.01 "off" / 1 "on" (:1472). */
typedef struct rt_camera {
    double C[3];
    double right[3];
    double up[3];
    double view[3];
    double zdist;
    double focus;
    double aa_sigma;
    double dof_sigma;
} rt_camera;

/* render flags */
#define RT_F_POINT_LIGHT 1u /* direct light of raytracer.cpp:1040-1055 (point light, hard shadow); objects[0] is then an ordinary object */
#define RT_F_NO_INDIRECT 2u /* skip the indirect bounce of raytracer.cpp:1086-1088 */
#define RT_F_NO_FRESNEL 4u  /* dielectric always refracts unless total reflection (raytracer.cpp:1033) */
#define RT_F_NO_JITTER 8u   /* camera ray Ray(C,u) through the pixel centre (raytracer.cpp:1482), no RNG draws */
#define RT_F_COUNTERS 16u   /* accumulate rt_counters (slower) */
#define RT_F_NO_PRUNE 32u   /* trace every ray and visit every BVH node whose box the ray's line crosses, as Scene::getColor and
                               TriangleMesh::intersect do (raytracer.cpp:953-1093, :652-666); default: skip boxes that start behind
                               the closest hit found so far, end a shadow ray at its first hit before the light, and do not trace
                               rays whose answer cannot change the sample - the shadow ray of a vertex whose direct term is exactly
                               zero, the ray that leaves a vertex at depth max_rebond (intersected and then dropped, :960-963) -
                               with bit-identical results */

#define RT_F_FAST 64u       /* rt_render only, not combinable with the other flags: the same estimator, random stream and traversal in single
                               precision with FMA contraction. No counterpart in the reference (all double) and no bit-for-bit promise:
                               hit identities and radiance agree with the double-precision render statistically - the converged image
                               within the stated bias / RMSE bars - not sample by sample. Sphere lists of at most 64; the sphere tests
                               and the accumulator stay double; the surface offsets (epsilon, shadow_bias) are raised to a few units in
                               the last place of the hit position where the parameters are smaller */

typedef struct rt_params {
    int32_t W, H;
    int32_t spp;        /* samples rendered by this call, per pixel */
    int32_t spp_offset; /* global index of the first of them (spp sharding) */
    uint64_t seed;      /* RNG key; the reference seeds its engine with 10 (raytracer.cpp:26) */
    uint32_t flags;
    int32_t max_rebond;   /* 10: paths stop when rebond > max_rebond (raytracer.cpp:962) */
    double ior;           /* 1.4  (raytracer.cpp:984) */
    double epsilon;       /* 1e-5 (raytracer.cpp:955) */
    double shadow_bias;   /* 1e-4 (raytracer.cpp:1073) */
    int32_t rect[4];      /* j0, i0, j1, i1: render pixels j0<=j<j1, i0<=i<i1; all 0 = full frame */
} rt_params;

typedef struct rt_counters {
    uint64_t paths;      /* camera samples = Scene::getColor(r,0,false) calls (raytracer.cpp:1485) */
    uint64_t rays;       /* Scene::intersect calls (raytracer.cpp:930); without RT_F_NO_PRUNE: the rays actually traced (fewer) */
    uint64_t box_tests;  /* BoudingBox::intersect calls (raytracer.cpp:163) */
    uint64_t nodes;      /* BVH records popped (inner Noeuds visited) */
    uint64_t tris;       /* triangle tests (raytracer.cpp:670-687) */
    uint64_t mesh_calls; /* TriangleMesh::intersect calls that passed the root box */
    uint64_t diffuse;    /* diffuse vertices shaded (raytracer.cpp:1059-1088) */
    uint64_t specular;   /* mirror / dielectric vertices (raytracer.cpp:974-1021) */
} rt_counters;

typedef struct rt_ctx rt_ctx;

/* Create a context on HIP device `device` (its own stream). */
int rt_create(int device, rt_ctx **out);
void rt_destroy(rt_ctx *ctx);
/* Message of the last failed call on ctx (or of the calling thread's last failed rt_create when ctx is NULL). */
const char *rt_last_error(const rt_ctx *ctx);

/* Replaces scene construction + the objects vector of main()
 * (raytracer.cpp:1163-1435): copies the flattened scene to HBM. objects[0]
 * must be a sphere unless every render uses RT_F_POINT_LIGHT: the reference
 * dynamic_casts it unchecked (raytracer.cpp:970, :1062). */
int rt_upload_scene(rt_ctx *ctx, const rt_scene_desc *scene);

/* Replaces the pixel loop body raytracer.cpp:1457-1486 for p.spp samples of
 * every pixel of p.rect: adds the radiance of sample k (k = spp_offset ..
 * spp_offset+spp-1, in that order) into accum[(i*W + j)*3 + c], i = the
 * reference's row index (not flipped).  accum is a DEVICE pointer to
 * W*H*3 doubles, or NULL for the context's own accumulator. Asynchronous on
 * the context's stream; rt_sync waits.  The samples of a pixel are summed in
 * order within a work unit (8 samples on rects with at least two pixels per
 * resident lane - 262 144 pixels on an MI355X -, 4, 2 or 1 on smaller ones)
 * and the unit sums are added to accum in sample order, as calls of one unit
 * each would: results are run-to-run identical. */
int rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_params *p, double *accum);
int rt_sync(rt_ctx *ctx);

/* rt_render for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc). It has no counterpart in the reference. Same arguments and the same result bits
 * as rt_render, but it calls no hipMalloc / hipFree: the tile counter and the sums of a launch's
 * work units live in a static device array of the library (RT_STATIC_SCRATCH_BYTES for the unit sums; one copy per
 * GPU, held until the process ends) instead of buffers the context grows on demand. A launch takes as many whole
 * work units as fit the array (rect pixels * 24 bytes each, with the unit of rt_render: 8 samples from 262 144 pixels
 * up), so a large frame is rendered in more launches of fewer samples - a 1024 x 1024 frame in launches of 16 samples
 * instead of 64 - with the same units added in the same order, hence the same bits. Limit: a call for more than one
 * unit of a rectangle of which not even two units fit (more than RT_STATIC_SCRATCH_BYTES / 48 = 1 398 101 pixels)
 * fails with RT_E_INVALID and a message that says what would fit; a call of at most `unit` samples needs no unit
 * sums and always fits. accum must be a device pointer the caller owns, or NULL for the context's accumulator
 * (allocated beforehand by rt_accum_clear). Every call on one GPU shares that GPU's array: calls on one context are
 * ordered by its stream; a call while ANOTHER context's static render is still running on the same GPU would race on
 * the array and fails with RT_E_STATE instead (rt_sync the other context first). */
#define RT_STATIC_SCRATCH_BYTES (64ull << 20)
int rt_render_static(rt_ctx *ctx, const rt_camera *cam, const rt_params *p, double *accum);
/* Zero / read back the context's own accumulator (W*H*3 doubles). */
int rt_accum_clear(rt_ctx *ctx, int32_t W, int32_t H);
int rt_accum_download(rt_ctx *ctx, double *host_out, int64_t n_doubles);
/* Load the context's own accumulator from host memory (resume an interrupted render: the accumulator is a
 * plain sum of samples). rt_accum_clear(W, H) must have sized it. */
int rt_accum_upload(rt_ctx *ctx, const double *host_in, int64_t n_doubles);
/* Device address of the context's own accumulator (NULL before rt_accum_clear). */
double *rt_accum_device_ptr(rt_ctx *ctx);

/* Multi-GPU within one process: dst's own accumulator += src's own accumulator (same W, H), where the two contexts may
 * sit on different GPUs (peer copy over xGMI into a staging buffer of dst, then an add kernel on dst's stream) or share
 * one. This is the one exchange step of samples-per-pixel sharding - each context renders its own range of sample
 * indices (rt_params.spp_offset) of every pixel, the reference's "rows in parallel" (raytracer.cpp:1456) turned into
 * "samples in parallel" - for callers that drive several GPUs from threads of one process; processes-per-GPU callers
 * sum with RCCL instead. Waits for src's renders, returns when the sum is in dst. Adding the contexts in a fixed order
 * gives run-to-run identical bits. Two GPUs without peer access to each other (hipDeviceCanAccessPeer), or a peer copy
 * that the runtime refuses, are served through a pinned host buffer instead. The calling thread's current HIP device is
 * the same after the call as before it, also when the call fails; a failure is reported on dst (rt_last_error(dst))
 * with the HIP error's text. No other thread may use src (render into it, clear or destroy it) during the call. */
int rt_accum_add(rt_ctx *dst, rt_ctx *src);

/* Replaces raytracer.cpp:1488-1501: colour/spp_total, pow(.,0.45), min(255,.),
 * truncation to unsigned char, vertical flip. accum = DEVICE pointer or NULL
 * (own accumulator); rgb8_host receives W*H*3 bytes. Synchronous. */
int rt_resolve(rt_ctx *ctx, const double *accum, int32_t W, int32_t H, double spp_total,
               uint8_t *rgb8_host);

/* Parity probe = Scene::intersect (raytracer.cpp:930-951) on caller-supplied
 * rays. rays = n * 6 doubles (origin xyz, direction xyz), HOST pointer.
 * Outputs (HOST pointers, each may be NULL): obj_id = objectId or -1 on a
 * miss; tri_id = orig_id of the hit triangle or -1 for spheres/misses;
 * t = hit parameter (1e10 on a miss, as the reference initialises it);
 * PN = n*6 doubles: P then N as Scene::intersect returns them;
 * albedo = n*3 doubles. Synchronous. */
int rt_intersect_rays(rt_ctx *ctx, const double *rays, int64_t n, int32_t *obj_id,
                      int32_t *tri_id, double *t, double *PN, double *albedo);

/* Per-path radiance probe: radiance of the single sample (i, j, k) exactly as
 * rt_render would add it. out = n*3 doubles (HOST); ijk = n*3 int32 (HOST). */
int rt_sample_radiance(rt_ctx *ctx, const rt_camera *cam, const rt_params *p, const int32_t *ijk,
                       int64_t n, double *out);

/* rt_sample_radiance plus the path each sample took, for explaining a difference between two implementations bounce by
 * bounce. trace (HOST, n * RT_TRACE_LEN * 4 int32, may be NULL): entry d of a sample describes the Scene::intersect call
 * of Scene::getColor at recursion depth `rebond` = d (raytracer.cpp:960):
 *   [0] objectId of the hit or -1 (miss),  [1] orig_id of the hit triangle or -1,
 *   [2] random draws the path had used before that call,
 *   [3] what getColor did with the hit: one of RT_TRACE_*.
 * Entries the path never reached hold {-1, -1, 0, RT_TRACE_UNUSED}. */
#define RT_TRACE_LEN 12               /* rebond 0 .. 11: the call at depth 11 is made and discarded (raytracer.cpp:960-963) */
#define RT_TRACE_UNUSED 0
#define RT_TRACE_NONE 1               /* miss, or the depth limit: nothing further */
#define RT_TRACE_LIGHT 2              /* objects[0] (raytracer.cpp:966-973) */
#define RT_TRACE_MIRROR 3             /* :974-979 */
#define RT_TRACE_TOTAL_REFLECTION 4   /* :994-999 */
#define RT_TRACE_REFLECT 5            /* the Fresnel coin chose reflection (:1008-1013) */
#define RT_TRACE_REFRACT 6            /* it chose refraction (:1015-1020) */
#define RT_TRACE_DIFFUSE_LIT 7        /* diffuse vertex, shadow ray not blocked (:1073-1083) */
#define RT_TRACE_DIFFUSE_SHADOWED 8   /* diffuse vertex, shadow ray blocked */
int rt_sample_trace(rt_ctx *ctx, const rt_camera *cam, const rt_params *p, const int32_t *ijk, int64_t n, double *out,
                    int32_t *trace);

/* Counters accumulated by renders with RT_F_COUNTERS since the last reset. */
int rt_get_counters(rt_ctx *ctx, rt_counters *out);
int rt_reset_counters(rt_ctx *ctx);

/* Duration in milliseconds of the render kernels launched by the LAST
 * rt_render call, measured with HIP events on the context's stream; waits
 * for them. n_launches (may be NULL) receives how many kernels that was. */
int rt_last_render_ms(rt_ctx *ctx, double *ms, int32_t *n_launches);

/* The same over many calls: total duration in milliseconds of the render kernels of every rt_render call since the last
 * reset (each call has its own pair of HIP events on the context's stream, so nothing waits between the calls; this
 * function waits for the last of them), and how many kernels that was. reset != 0 starts a new total. */
int rt_render_time_ms(rt_ctx *ctx, double *ms_total, int32_t *n_launches, int32_t reset);

/* The HIP stream of the context as a void* (hipStream_t), for callers that
 * want to order their own work after a render. */
void *rt_stream(rt_ctx *ctx);

/* TriangleMesh::buildBVH (raytracer.cpp:220-291) on the GPU, with the reference's result: the same boxes, the same
 * tree and the same order of the triangles. verts = nv * 3 doubles, tri_vtx = nt * 3 vertex indices (vtxi, vtxj, vtxk of
 * TriangleIndices), both HOST pointers. order (nt entries) receives, for every position of the permuted triangle array,
 * the index of the triangle that the loop of :268-278 has moved there; nodes (node_cap entries; 2 * nt + 1 always
 * suffice) receives the Noeud records in the order the reference's recursion creates them (node 0 = root, a node's
 * children get the next two numbers when it is split); n_nodes their count. Synchronous. */
typedef struct rt_build_node {
    double box[6];       /* b.mini, b.maxi */
    int32_t debut, fin;  /* range in the permuted triangle array */
    int32_t fg, fd;      /* children, -1 for a leaf */
    int32_t depth, pad;
} rt_build_node; /* 72 bytes */
int rt_build_bvh(rt_ctx *ctx, const double *verts, int64_t nv, const int32_t *tri_vtx, int64_t nt, int32_t *order,
                 rt_build_node *nodes, int64_t node_cap, int64_t *n_nodes);

/* Library version / build info string. */
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
