/*
 * rt_device.hip.h - device code of the path-tracing hot path for gfx950 (wave64).
 *
 * FP64 throughout, compiled with -ffp-contract=off: every expression keeps the operation order of
 * the reference expression it replaces (cited per function), so hit parameters, hit identities and
 * noise-free radiance equal the CPU results bit for bit up to libm differences in cos/sin/log/pow.
 *
 * Execution model: one workgroup = ONE wave of 64 lanes, one lane = one pixel, persistent over its
 * samples (and, in the render kernel, over the pixels it draws from a global tile counter).
 *   - The recursive Scene::getColor (raytracer.cpp:953-1093) is an iterative state machine with
 *     one closest-hit query per loop iteration. A diffuse vertex leaves a shadow ray and an indirect
 *     ray that do not depend on each other: both go into the same query (two ray slots per lane),
 *     and a lane whose path ends starts its next sample at once.
 *   - The closest-hit query is wave-cooperative. Spheres are tested by each lane for its own rays
 *     (uniform work). Every (ray, BVH record) and (ray, triangle) test after that is an item in a
 *     pool kept in LDS; in each pool step the 64 lanes pop 64 items - whichever rays they belong
 *     to - test them, and push the children that were hit. Lane utilisation no longer depends on
 *     how many of the wave's rays reach a mesh or how deep each goes (a per-lane traversal ran at
 *     6 % VALU lane utilisation on the battle scene).
 *   - Which tree is walked: by default a filter tree of this library (rt_own_bvh.hpp: one tree over
 *     all meshes, four children per record, single-precision boxes rounded outwards), which can only
 *     say "certainly missed"; its winner is the closest triangle under the reference's tie rule, and
 *     is accepted once the reference's own leaf box for that triangle is certainly hit (rays_end).
 *     The few rays for which that cannot be said go round again on the reference's trees.
 *     With RT_F_NO_PRUNE (and in the counted variant) the pool visits exactly the BVH records and
 *     triangles the reference visits (every node whose box test passes: raytracer.cpp:652-666), in a
 *     different order. The reference's result depends on order only through exact ties in t, where
 *     the first visited wins; triangles are laid out in that visiting order (see rt_tri in rt_amd.h),
 *     so the tie-break is "lowest position" and the result is order-independent.
 *   - Box tests run in single precision with a per-ray error bound and decide only what is beyond
 *     doubt; what they leave open is settled by the reference's six divisions on the FP64 boxes.
 *     Distance pruning and any-hit shadow rays skip work that cannot change the result.
 *   - The wave's LDS is kept within an eighth of the CU's 160 KiB (the kernel needs 256 VGPRs, so
 *     eight waves per CU is the most that can be resident): ray tables hold doubles only where the
 *     triangle test needs them, and the host picks the node stack's height per scene.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"
#include "rt_math.h"
/* cos / sin / log of the sampling code: rt_math.h's functions as real calls. Inlined, their constants and temporaries
 * lengthen the caller's live ranges: 256 VGPRs with spills and 7 % less throughput, against 237 VGPRs without spills. */
namespace rtd {
__device__ __noinline__ double2 dev_sincos(double x) { double s, c; rtm::sincos_0_7(x, &s, &c); return make_double2(s, c); }
__device__ __noinline__ double dev_log(double x) { return rtm::log_pos(x); }
/* two at once: the evaluations do not depend on each other and their instructions are issued interleaved */
__device__ __noinline__ double4 dev_sincos2(double x, double y) {
    double s, c, s2, c2;
    rtm::sincos_0_7(x, &s, &c);
    rtm::sincos_0_7(y, &s2, &c2);
    return make_double4(s, c, s2, c2);
}
__device__ __noinline__ double2 dev_log2(double x, double y) { return make_double2(rtm::log_pos(x), rtm::log_pos(y)); }
}
#define RT_SINCOS(arg_, ps_, pc_) do { const double2 sc_ = rtd::dev_sincos(arg_); *(ps_) = sc_.x; *(pc_) = sc_.y; } while (0)
#define RT_LOG(arg_) rtd::dev_log(arg_)

namespace rtd {

#define RT_WAVE 64
#define RT_LDS_SPHERES_MAX 64     /* sphere lists up to this size are staged in LDS */
#ifndef RT_NODE_SOFT
#define RT_NODE_SOFT 384          /* node-item stack height up to which 64 items are popped per step: the largest value the host
                                     gives DScene::node_soft; it takes a smaller one (down to RT_NODE_SOFT_MIN) if that lets one
                                     more wave stay resident per CU */
#endif
#define RT_NODE_SOFT_MIN 288
#define RT_FEED_MARGIN (2 * RT_WAVE) /* mesh roots are fed to the pool while fewer than node_soft - RT_FEED_MARGIN node items are
                                        pending: a fed mesh adds up to 128 (measured: the more are fed the better) */
#ifndef RT_LEAF_CAP
#define RT_LEAF_CAP 320           /* leaf-item stack entries */
#endif
#ifndef RT_LEAF_TRIGGER
#define RT_LEAF_TRIGGER 0         /* pending leaf items that make the next pool step a triangle step; 0: as many as one step takes */
#endif
#define RT_RAY_SLOTS 128          /* rays of one query: two per lane (slot = lane, 64 + lane) */
#define RT_NODE_BITS 25           /* record-index / leaf-count bits in a pool item (the ray slot takes the top 7) */
#define RT_LOW_MASK ((1u << RT_NODE_BITS) - 1u)
#define RT_MAX_DEPTH 60
#define RT_STEP_LIMIT (1u << 24)  /* pool steps of one query after which it is given up (and reported) */
#ifndef RT_OWN_LEAF
#define RT_OWN_LEAF 4             /* primitives per leaf of the filter trees, at most 8 (three bits in a child code) */
#endif
#ifndef RT_OWN_PAIR_MAX
#define RT_OWN_PAIR_MAX 2000000   /* scenes up to this many triangles get filter-tree leaves of two triangles (rt_capi.hip) */
#endif
#ifndef RT_LEAF_QUAD
#define RT_LEAF_QUAD 4            /* lanes per leaf item in a triangle step (0: as many as the item has triangles, found by a prefix sum): with
                                     leaves of at most this many triangles, lane r of an item's group takes its r-th triangle and nothing
                                     has to be spread out. A step then takes 16 items and costs a third of the prefix-sum form (cfg4 +1 %). */
#endif
#ifndef RT_SPH_TRIGGER
#define RT_SPH_TRIGGER 64         /* the same for the sphere tree's leaf items: a step of theirs is cheap and best taken full (cfg5: +0.8 % against 32) */
#endif
#ifndef RT_SPH_LEAF
#define RT_SPH_LEAF 1             /* spheres per leaf of the tree over a long sphere list, at most 8; 1: a leaf item is one test, and the
                                     leaf step needs no prefix sum to spread items over lanes (cfg5: +5.6 % against 4) */
#endif
#ifndef RT_EXT_LEAF
#define RT_EXT_LEAF 4             /* reference leaves with more triangles than this get extension records down to this size */
#endif
#ifndef RT_TRAV_PRIO
#define RT_TRAV_PRIO 1            /* issue priority of a wave while it is in the pool traversal, whose steps are chains of dependent
                                     instructions (elsewhere RT_SHADE_PRIO): 1, 2, 3 all +1.2 % on the battle scene; the reverse -0.7 % */
#endif
#ifndef RT_NARROW_STEP
#define RT_NARROW_STEP 2          /* 1: node steps with at most 32 records give each record two lanes (two children per lane); 2: and those
                                     with at most 16 four lanes (one child per lane): the steps that end a query are shorter (0 -> 1: cfg4
                                     +1.3 %, cfg5 +5.3 %, cfg3 +2.3 %; 1 -> 2: +0.2 %, cfg5 +3.7 %) */
#endif
#ifndef RT_SHADE_PRIO
#define RT_SHADE_PRIO 0
#endif
#ifndef RT_TOP_LDS
#define RT_TOP_LDS 0              /* 1: the four records below the filter tree's root (its children; the root itself is read with scalar
                                     loads at feed time) are staged in the wave's LDS, and a node step all of whose records are among
                                     them reads LDS instead of global memory */
#endif
#ifndef RT_PUSH_BRANCHLESS
#define RT_PUSH_BRANCHLESS 0      /* 1: a child that is not pushed is stored to a scratch word of the lane instead of being skipped by a
                                     branch around the store */
#endif
#ifndef RT_NARROW_SPLIT
#define RT_NARROW_SPLIT 0         /* T > 32: a node step that finds 33 .. T records pops 32 of them with two lanes each and leaves the
                                     rest for the next step, instead of a full-width step that is half empty */
#endif
static_assert(RT_NODE_SOFT_MIN > RT_FEED_MARGIN + RT_WAVE && RT_NODE_SOFT_MIN <= RT_NODE_SOFT, "the feed needs room for a mesh above a useful stack height");
static_assert(RT_LEAF_CAP >= 5 * RT_WAVE, "a node step adds up to 256 leaf items (filter tree: four children) and runs only when 256 slots are free; a fed mesh adds up to 128 on top of 64");
#define RT_PHANTOM_BASE (1 << 30)  /* best_pos values from here on: "mesh (value - base) reports a hit without a triangle" */

/* what the root phase needs of a mesh: the root box (raytracer.cpp:644) and where the tree starts */
struct RootRec { /* 32 bytes; the box rounded to float like NodeRec32, the double box stays in rt_mesh */
    float box[6];
    int32_t root, root_count;
};

/* Device form of one child of a BVH record (rt_bvh_node of rt_amd.h is repacked at upload): the two
 * children of record r are halves 2r and 2r+1, 64 bytes each, so that a pair of lanes fetches one
 * record with each lane reading only the box it tests. */
struct NodeHalf {
    double box[6];
    int32_t child; /* inner child: record index; leaf: first triangle position */
    int32_t count; /* 0 for an inner child, else the leaf's triangle count */
    int32_t pad[2];
};

/* What the traversal reads: one 64-byte record per BVH node with both child boxes rounded to float. The
 * fast box test decides on these and allows for the rounding in its tolerance; only an undecided test
 * reads the double box of NodeHalf. */
struct NodeRec32 {
    float boxL[6], boxR[6];
    int32_t L, R, nL, nR;
};

/* Record of the filter tree: four children, boxes in float rounded outwards. child: RT_CHILD_EMPTY, or a leaf
 * (bit 31 set, bits 30-28 = triangle count - 1, bits 27-0 = first triangle in the filter tree's order), or the
 * index of another record. */
#define RT_CHILD_EMPTY 0xFFFFFFFFu
struct NodeRec4 {
    float box[4][6];
    uint32_t child[4];
    uint32_t pad[4];
};

#ifndef RT_TRI_FIN
#define RT_TRI_FIN 1 /* 1: the epilogue of a query reads the winner's leaf box, object id and texture group from one record per triangle
                        instead of triangle -> (leaf index, mesh) -> (leaf box, object id): one dependent load less */
#endif
struct TriFin { /* 32 bytes */
    float box[6];      /* the box of the reference leaf that holds the triangle, rounded to float (to nearest) */
    int32_t object_id; /* of the triangle's mesh */
    int32_t group;     /* rt_tri_attr::group */
};

struct DScene {
    const rt_sphere *spheres;
    const RootRec *roots;
    const rt_mesh *meshes;
    const NodeHalf *halves;   /* 2 per BVH record */
    const NodeRec32 *rec32;   /* 1 per BVH record, boxes in float */
    /* the filter tree over all triangles of the scene (rt_own_bvh.hpp); null when it could not be built */
    const NodeRec4 *own_rec4;
    const rt_tri *own_tris;   /* triangles in the filter tree's order; orig_id holds the position in `tris` */
    const int32_t *tri_leaf;  /* by triangle position: index of the reference leaf that holds it */
    const float *leaf_box32;  /* 6 floats per reference leaf: its box rounded to float (to nearest) */
    const TriFin *tri_fin;    /* by triangle position: what settles a hit on that triangle, in one record (RT_TRI_FIN) */
    uint32_t n_own_nodes;
    uint32_t n_own_tris;      /* triangles in own_tris: those the reference's traversal can reach */
    const rt_tri *tris;       /* by triangle position (80 bytes each: a leaf's triangles are contiguous) */
    const rt_tri_attr *attrs;
    const rt_texture *tex;
    const uint8_t *texels;
    const double *lut;
    int32_t n_spheres, n_meshes;
    int32_t node_cap;  /* node-item stack entries: node_soft + the deepest tree's needs + 2 */
    int32_t node_soft; /* stack height up to which 64 items are popped per step (RT_NODE_SOFT_MIN .. RT_NODE_SOFT) */
    int32_t own_leaf_shift; /* a leaf of the filter tree holds at most 1 << own_leaf_shift triangles (1 or 2): the lanes per leaf item of a triangle step */
    double bmax[3];   /* largest |coordinate| of any box of the scene, per axis (infinity if one is not finite) */
    uint32_t n_tris_total, n_nodes_total; /* array lengths: every index taken from the pool is checked against them */
    uint32_t n_ref_nodes;                 /* records below this index are the reference's; the rest are extension records */
    int32_t count_own; /* counted kernel variants only: 1 = walk what a production render walks (filter tree first, the
                          reference's trees for unconfirmed rays) and count that work; 0 = walk and count the reference's
                          traversal, whose counters equal the CPU checker's */
    int *overflow;    /* bit 0: a push would have passed a stack's capacity; bit 1: an index outside the arrays */
    /* Sphere lists too long for LDS: a tree over the spheres' boxes, in the filter tree's record format (four children per
     * record), that the wave's pool traverses like the triangles' tree; `spheres` is then in that tree's leaf order behind
     * the n_sph_big spheres that are not in it. Null for short lists. The boxes are the spheres' boxes enlarged by a
     * thousandth of the radius and rounded outwards to float: Sphere::intersect (raytracer.cpp:883-918) can report a hit
     * for a ray that passes a sphere by a rounding error, which stays inside that margin as long as the ray starts within
     * a hundred thousand radii (sph_reach). */
    const NodeRec4 *sph_rec4;
    const int32_t *sph_of_obj; /* by objectId: position of that sphere in `spheres` (-1 for a mesh) */
    uint32_t n_sph_rec4;
    int32_t n_sph_big;   /* spheres at the front of the list that are not in the tree (far larger than the rest): tested by every ray */
    double sph_reach; /* rays whose origin lies farther than this from the coordinate origin (any axis) test every sphere instead */
    double I;
    double L[3];
    double lightO[3];
    double lightR;
};

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
template <typename P>
__device__ __forceinline__ V3 ld3(P p) { return V3{p[0], p[1], p[2]}; } /* P: a pointer to three doubles, in any address space */
/* Vector operators, raytracer.cpp:62-110 */
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 smul(double s, V3 b) { return mk(s * b.x, s * b.y, s * b.z); }
__device__ __forceinline__ V3 muls(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 divs(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double sqr(double v) { return v * v; }
__device__ __forceinline__ double norm2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ V3 unit(V3 a) { /* get_normalized, raytracer.cpp:44-48: no zero guard */
    double n = sqrt(norm2(a));
    return mk(a.x / n, a.y / n, a.z / n);
}
/* std::min / std::max semantics: first argument unless the comparison holds (NaN handling) */
__device__ __forceinline__ double smin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double smax(double a, double b) { return (a < b) ? b : a; }

#define RT_PI 3.14159265358979323846

/* ---- counter-based random stream: draw n of path (seed, pixel, sample) ------------------- */
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rng_key(uint64_t seed_mixed, uint32_t pixel, uint32_t sample) {
    return mix64(seed_mixed + (((uint64_t)pixel << 32) | (uint64_t)sample));
}
__device__ __forceinline__ uint64_t rng_draw(uint64_t key, uint32_t n) {
    return mix64(key + ((uint64_t)n + 1ull) * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ double rng_uniform(uint64_t key, uint32_t n) {
    return (double)(rng_draw(key, n) >> 11) * (1.0 / 9007199254740992.0);
}

struct Cnt {
    uint64_t paths, rays, box_tests, nodes, tris, mesh_calls, diffuse, specular;
    /* diagnostic build only (RT_F_COUNTERS): wave-level cycle stamps and pool occupancy */
    uint64_t cyc_fresh, cyc_roots, cyc_node, cyc_leaf, cyc_finish, cyc_shade;
    uint64_t node_steps, node_lanes, leaf_steps, leaf_lanes, queries, query_lanes;
    uint64_t dbg[8];
    uint64_t sph_tests, sph_nodes; /* exact sphere tests done and sphere-tree nodes visited (per lane, like the public counters) */
    /* count_own runs: box tests in the reference's double-precision form (undecided single-precision tests, confirmation of a
     * filter-tree winner) */
    uint64_t box_exact;
};
#define RT_STAMP() ((uint64_t)__builtin_amdgcn_s_memtime())
#ifndef RT_STAMP_STEPS
#define RT_STAMP_STEPS 0 /* 1: the counted kernel also times the parts of a pool step (it then waits for the loads where it stamps) */
#endif

/* BoudingBox::intersect, raytracer.cpp:163-186 (six true divisions, no bound on t) */
struct Box6 {
    double lx, ly, lz, hx, hy, hz;
};
__device__ __forceinline__ Box6 ldbox(const double *b) { return Box6{b[0], b[1], b[2], b[3], b[4], b[5]}; }
__device__ __noinline__ bool box_hit_exact(const double *b, double cx, double cy, double cz, double ux, double uy, double uz) {
    double t1x = (b[0] - cx) / ux, t2x = (b[3] - cx) / ux;
    double txMin = smin(t1x, t2x), txMax = smax(t1x, t2x);
    double t1y = (b[1] - cy) / uy, t2y = (b[4] - cy) / uy;
    double tyMin = smin(t1y, t2y), tyMax = smax(t1y, t2y);
    double t1z = (b[2] - cz) / uz, t2z = (b[5] - cz) / uz;
    double tzMin = smin(t1z, t2z), tzMax = smax(t1z, t2z);
    double tMax = smin(txMax, smin(tyMax, tzMax));
    double tMin = smax(txMin, smax(tyMin, tzMin));
    if (tMax < 0) return false;
    return tMax > tMin;
}

/* Per-ray constants of the fast box test: iu = 1/u (to 2^-45, rcp_near), nc = -(C * iu), and the tolerance below
 * which the fast test does not decide. The slab parameter (b - C) / u of the reference is formed as
 * fma(b', iu, nc), where b' is the box coordinate b rounded to float (|b' - b| <= 2^-24 |b|). With T the
 * real-number value and M = max over the axes of (bmax + |C|) * |iu|:
 *   |fma(b', iu, nc) - T| <= 2^-24 |b iu| + 2^-44 (|T| + |b iu| + |C iu|),   |reference - T| <= 2^-53 * 2 |T| (1 + tiny),
 * so the two differ by less than 6.0e-8 M; min and max do not enlarge a difference. tol = 1.3e-7 M
 * covers that twice over (the test compares tMax - tMin, which carries two such errors). Rays for which
 * the bound cannot be trusted (a zero, huge, tiny or non-finite direction component or origin, or a
 * scene whose boxes do not fit a float) get tol = NaN: no comparison with it holds and every one of
 * their boxes goes to the exact test. This double-precision form confirms the filter tree's winners (rays_end);
 * the traversal itself uses the single-precision form below. */
struct RayK {
    V3 iu, nc;
    double tol;
};
/* 1/x to a relative error below 2^-45 (hardware estimate + one Newton step) instead of the ~35 instructions of a
 * correctly rounded division: far inside the margins of the box tests. Zero, denormal and non-finite x give
 * infinities or NaN, which ray_constants turns into tol = NaN. */
__device__ __forceinline__ double rcp_near(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ RayK ray_constants(const DScene &sc, V3 C, V3 u) {
    RayK k;
    k.iu = mk(rcp_near(u.x), rcp_near(u.y), rcp_near(u.z));
    k.nc = mk(-(C.x * k.iu.x), -(C.y * k.iu.y), -(C.z * k.iu.z));
    const double ax = fabs(k.iu.x), ay = fabs(k.iu.y), az = fabs(k.iu.z);
    const double mx = (sc.bmax[0] + fabs(C.x)) * ax, my = (sc.bmax[1] + fabs(C.y)) * ay, mz = (sc.bmax[2] + fabs(C.z)) * az;
    const bool ok = (sc.bmax[0] < 1e30) && (sc.bmax[1] < 1e30) && (sc.bmax[2] < 1e30) && (mx < 1e30) && (my < 1e30) && (mz < 1e30) && (ax > 1e-25) && (ay > 1e-25) && (az > 1e-25) &&
                    (ax < 1e25) && (ay < 1e25) && (az < 1e25) && (fabs(C.x) < 1e200) && (fabs(C.y) < 1e200) && (fabs(C.z) < 1e200);
    const double m = fmax(mx, fmax(my, mz));
    k.tol = ok ? 1.3e-7 * m : __builtin_nan("");
    return k;
}

/* The decision of BoudingBox::intersect (raytracer.cpp:163-186: "tMax < 0 -> miss, else tMax > tMin")
 * whenever it is beyond doubt: hit if both tMax - tMin and tMax exceed tol, miss if either is below
 * -tol, otherwise `unsure` is set and the caller runs box_hit_exact. A zero-thickness box (tMax == tMin
 * in the reference, never hit) is always unsure. tlo is a lower bound of the reference's tMin. */
__device__ __forceinline__ void box_fast(const Box6 &b, const RayK &k, bool &hit, bool &unsure, double &tlo) {
    const double t1x = __builtin_fma(b.lx, k.iu.x, k.nc.x), t2x = __builtin_fma(b.hx, k.iu.x, k.nc.x);
    const double t1y = __builtin_fma(b.ly, k.iu.y, k.nc.y), t2y = __builtin_fma(b.hy, k.iu.y, k.nc.y);
    const double t1z = __builtin_fma(b.lz, k.iu.z, k.nc.z), t2z = __builtin_fma(b.hz, k.iu.z, k.nc.z);
    const double tMax = fmin(fmin(fmax(t1x, t2x), fmax(t1y, t2y)), fmax(t1z, t2z));
    const double tMin = fmax(fmax(fmin(t1x, t2x), fmin(t1y, t2y)), fmin(t1z, t2z));
    const double d = tMax - tMin;
    hit = (d > k.tol) && (tMax > k.tol);
    const bool miss = (d < -k.tol) || (tMax < -k.tol);
    unsure = !(hit || miss);
    tlo = tMin - k.tol;
}

/* a float that is certainly not above t (NaN for t beyond the float range: never "greater than" anything) */
__device__ __forceinline__ float lower_float(double t) {
    const float f = (float)t;
    return __builtin_fmaf(-__builtin_fabsf(f), 0x1p-22f, f) - 0x1p-125f;
}
__device__ __forceinline__ float lower_float(float f) { return __builtin_fmaf(-__builtin_fabsf(f), 0x1p-22f, f) - 0x1p-125f; }
/* a float that is certainly not below t */
__device__ __forceinline__ float upper_float(double t) {
    const float f = (float)t;
    return __builtin_fmaf(__builtin_fabsf(f), 0x1p-22f, f) + 0x1p-125f;
}

/* The traversal's form of the same test, in single precision: iu, nc rounded to float, slab parameter
 * fmaf(b', iu, nc). Against the real value that is off by at most 2^-24 (3 |b iu| + 2 |C iu|) <= 3 * 2^-24 M = 1.8e-7 M
 * (rounding of b, iu, nc and of the result; ray_constants keeps |iu| and M inside the normal float range or marks the
 * ray), so tMax - tMin carries 3.6e-7 M and RT_TOLF_SCALE * tol = 4.2e-7 M decides safely. Only what the fast test
 * leaves undecided reaches the FP64 boxes and the reference's divisions. */
#define RT_TOLF_SCALE 3.2
struct RayKf {
    float iux, iuy, iuz, ncx, ncy, ncz;
    float tol; /* >= 0, or NaN */
};
__device__ __forceinline__ void box_fast_f(float lx, float ly, float lz, float hx, float hy, float hz, const RayKf &k, bool &hit, bool &unsure, float &tlo) {
    const float t1x = __builtin_fmaf(lx, k.iux, k.ncx), t2x = __builtin_fmaf(hx, k.iux, k.ncx);
    const float t1y = __builtin_fmaf(ly, k.iuy, k.ncy), t2y = __builtin_fmaf(hy, k.iuy, k.ncy);
    const float t1z = __builtin_fmaf(lz, k.iuz, k.ncz), t2z = __builtin_fmaf(hz, k.iuz, k.ncz);
    const float tMax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
    const float tMin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
    const float d = tMax - tMin;
    hit = (d > k.tol) && (tMax > k.tol);
    const bool miss = (d < -k.tol) || (tMax < -k.tol);
    unsure = !(hit || miss);
    tlo = tMin - k.tol;
}

/* triangle test, raytracer.cpp:673-687 */
/* keeps a value in a register at this point: what produces it cannot be moved below, what uses it not above */
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin3(V3 &a) { pin(a.x); pin(a.y); pin(a.z); }

struct Tri3 {
    V3 A, e1, e2;
    int32_t id; /* rt_tri::orig_id */
};
template <typename P>
__device__ __forceinline__ Tri3 tri_load(P base, uint32_t pos) {
    const auto tr = base + pos;
    Tri3 t;
    t.A = ld3(tr->A);
    t.e1 = ld3(tr->e1);
    t.e2 = ld3(tr->e2);
    t.id = tr->orig_id;
    return t;
}
__device__ __forceinline__ bool tri_eval(const Tri3 &t, V3 C, V3 u, double &beta, double &gamma, double &localt) {
    V3 N = cross(t.e1, t.e2), AO = sub(C, t.A), AOu = cross(AO, u);
    double invUN = 1. / dot(u, N);
    beta = -dot(t.e2, AOu) * invUN;
    gamma = dot(t.e1, AOu) * invUN;
    double alpha = 1 - beta - gamma;
    localt = -dot(AO, N) * invUN;
    return beta >= 0 && gamma >= 0 && beta <= 1 && gamma <= 1 && alpha >= 0 && localt > 0;
}
__device__ __forceinline__ bool tri_test(const DScene &sc, uint32_t pos, V3 C, V3 u, double &beta, double &gamma, double &localt) {
    const Tri3 t = tri_load(sc.tris, pos);
    return tri_eval(t, C, u, beta, gamma, localt);
}

/* closest hit, in the form that lets the attributes be computed once at the end */
struct HitRec {
    double t;
    int32_t obj;  /* objectId, INT32_MAX while nothing is hit */
    int32_t prim; /* sphere index, or global triangle position */
    int32_t mesh; /* mesh index or -1 for a sphere */
    int32_t group; /* the triangle's texture group, for a path segment's mesh hit: read while the hit is settled, which also
                      brings the triangle's attribute record into the cache before the surface is evaluated */
    double beta, gamma;
    int32_t sph_tests, sph_nodes; /* work done on the spheres for this ray; read only by the counted kernel variant */
};

/* ---- per-wave LDS ------------------------------------------------------------------------- */
struct RenderArgs {
    rt_camera cam;
    rt_params p;
    uint64_t seed_mixed; /* mix64(seed + golden) */
    int32_t j0, i0, j1, i1;
    const double *cam_rays; /* camera rays made ahead of the launch (RT_CAM_RAY_DOUBLES each, [sample - cam_k0][row of the rectangle][column]), or null */
    int32_t cam_k0, cam_pad;
};
#define RT_CAM_RAY_DOUBLES 5 /* origin x, y (z is the camera's) and the direction */

struct WaveLDS {
    double ray[6][RT_RAY_SLOTS];          /* C, u of each ray */
    float rayf[6][RT_RAY_SLOTS];          /* 1/u, -(C/u) rounded to float: the traversal's box test */
    float tolf[RT_RAY_SLOTS];             /* |tolf|: below this the single-precision box test does not decide (NaN: never). The sign
                                             bit is set when the caller only asks whether something is hit before tmax */
    float thrf[RT_RAY_SLOTS];             /* not below the distance beyond which nothing can change the ray's result */
    unsigned long long best_t[RT_RAY_SLOTS]; /* bits of the smallest accepted triangle t (> 0, so the bit order is the value order) */
    int32_t best_pos[RT_RAY_SLOTS];       /* lowest triangle position among those with that t */
    int32_t mark[RT_WAVE];                /* triangle step: slot -> leaf item that starts there, or -1 */
    int32_t ex[RT_WAVE];                  /* triangle step: first slot of each leaf item */
#if RT_TOP_LDS
    float top[4][32];                     /* records 1 .. 4 of the filter tree (NodeRec4 as 32 words each) */
#endif
    DScene sc;                            /* the render kernel's copy of its scene and frame arguments: what is used between the
                                             queries is read from here, so that it does not occupy scalar registers during them */
    RenderArgs ra;
#ifdef RT_LDS_PAD /* diagnostic: fewer resident waves per CU */
    double lds_pad[RT_LDS_PAD / 8];
#endif
    uint2 leaf[RT_LEAF_CAP];              /* leaf items, see leaf_item() */
    uint2 node[1];                        /* node items: x = record index | ray slot << 25, y = entry parameter (float bits, rounded
                                             down); node_cap entries, then the staged spheres */
};
__device__ __forceinline__ const rt_sphere *lds_spheres(const WaveLDS *w, int node_cap) {
    size_t off = (offsetof(WaveLDS, node) + (size_t)node_cap * 8 + 15) & ~(size_t)15;
    return reinterpret_cast<const rt_sphere *>(reinterpret_cast<const unsigned char *>(w) + off);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
/* A scene array's pointer that was read from the LDS copy of the scene: the compiler cannot see where it points and would
 * use flat loads, which wait on the LDS counter as well. All scene arrays are in global memory. */
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
__device__ __forceinline__ const T __attribute__((address_space(1))) *in_global(const T *p) {
    return (const T __attribute__((address_space(1))) *)p;
}
#else
template <typename T>
__device__ __forceinline__ const T *in_global(const T *p) { return p; }
#endif
/* the lanes for which p holds, as the compiler's own lane mask (the __ballot of the HIP headers goes through an integer
 * compare: two more instructions per call) */
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int rank_in(unsigned long long mask) { /* set bits below this lane */
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* inclusive scans over the 64 lanes with DPP row shifts / row broadcasts (no LDS) */
#define RT_DPP(old, src, ctrl, rmask, bmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), (bmask), false)
__device__ __forceinline__ int wave_scan_add(int v) {
    int x = v;
    x += RT_DPP(0, v, 0x111, 0xf, 0xf); /* row_shr:1 */
    x += RT_DPP(0, v, 0x112, 0xf, 0xf); /* row_shr:2 */
    x += RT_DPP(0, v, 0x113, 0xf, 0xf); /* row_shr:3 */
    x += RT_DPP(0, x, 0x114, 0xf, 0xe); /* row_shr:4, banks 1-3 */
    x += RT_DPP(0, x, 0x118, 0xf, 0xc); /* row_shr:8, banks 2-3 */
    x += RT_DPP(0, x, 0x142, 0xa, 0xf); /* row_bcast:15 into rows 1 and 3 */
    x += RT_DPP(0, x, 0x143, 0xc, 0xf); /* row_bcast:31 into rows 2 and 3 */
    return x;
}
__device__ __forceinline__ int wave_scan_max(int v) {
    int x = v, t;
    t = RT_DPP(-1, v, 0x111, 0xf, 0xf); x = t > x ? t : x;
    t = RT_DPP(-1, v, 0x112, 0xf, 0xf); x = t > x ? t : x;
    t = RT_DPP(-1, v, 0x113, 0xf, 0xf); x = t > x ? t : x;
    t = RT_DPP(-1, x, 0x114, 0xf, 0xe); x = t > x ? t : x;
    t = RT_DPP(-1, x, 0x118, 0xf, 0xc); x = t > x ? t : x;
    t = RT_DPP(-1, x, 0x142, 0xa, 0xf); x = t > x ? t : x;
    t = RT_DPP(-1, x, 0x143, 0xc, 0xf); x = t > x ? t : x;
    return x;
}

/* Orders this wave's LDS accesses for the compiler. A wave's LDS instructions execute in program
 * order, so no hardware barrier is needed between the lanes of one wave (workgroup = one wave). */
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* Pruning margin: a box is skipped when its entry parameter exceeds the ray's threshold by more than
 * one part in a million. The triangle parameters the reference computes (raytracer.cpp:681-686) carry a
 * relative error of about 6e-16 / |cos(ray, normal)|; an accepted triangle whose computed t falls
 * outside its own box by more than 1e-6 would need |cos| < 1e-9, and accepted hits have a cos-weighted
 * distribution, so this happens about once in 1e18 hits. RT_F_NO_PRUNE visits everything. */
#define RT_PRUNE_SLACK 1.000001

/* One ray of a query. tmax / anyhit describe what the caller will do with the answer: anyhit = it only tests
 * "hit && t < tmax" (the shadow test, raytracer.cpp:1073), so hits at or beyond tmax need not be found and the
 * first one below it settles the ray. */
struct RayQ {
    bool active;
    V3 C, u;
    double tmax;
    bool anyhit;
};

/* Sphere::intersect, raytracer.cpp:883-918, and Scene::intersect's choice (:938: the lowest objectId wins an exact tie) */
/* SP (here and below): the type of the pointer to the sphere list - in LDS for the render kernel of short lists, anywhere otherwise */
template <typename SP>
__device__ __forceinline__ void sphere_test(SP sph, int s, V3 C, V3 u, HitRec &h) {
    V3 O = ld3(sph[s].O);
    double R = sph[s].R;
    int32_t oid = sph[s].object_id;
    double a = 1;
    V3 CO = sub(C, O);
    double b = 2 * dot(u, CO);
    double c = norm2(CO) - R * R;
    double delta = b * b - 4 * a * c;
    if (delta < 0) return;
    double sq = sqrt(delta);
    double t2 = (-b + sq) / (2 * a);
    if (t2 < 0) return;
    double t1 = (-b - sq) / (2 * a);
    double t = (t1 > 0) ? t1 : t2;
    if (t < h.t || (t == h.t && oid < h.obj)) { h.t = t; h.obj = oid; h.prim = s; h.mesh = -1; }
}
template <typename SP>
__device__ __forceinline__ void sphere_test_counted(SP sph, int s, V3 C, V3 u, HitRec &h) {
    h.sph_tests++;
    sphere_test(sph, s, C, u, h);
}

__device__ __forceinline__ void hit_none(HitRec &h) {
    h.t = 1E10;
    h.obj = INT32_MAX;
    h.prim = -1;
    h.mesh = -1;
    h.group = 0;
    h.beta = h.gamma = 0;
    h.sph_tests = h.sph_nodes = 0;
}
/* the margins of the sphere tree's boxes hold for rays that start within DScene::sph_reach of the coordinate origin */
__device__ __forceinline__ bool sphere_walk_ok(const DScene &sc, V3 C) { return fabs(C.x) < sc.sph_reach && fabs(C.y) < sc.sph_reach && fabs(C.z) < sc.sph_reach; }

__device__ __forceinline__ void ray_tables(const DScene &sc, WaveLDS *w, int slot, const RayQ &q, bool prune, const HitRec &h, bool ray_in_place = false);
/* Spheres for one ray and the ray's entry in the pool's tables. */
template <bool CNT, bool SPH, typename SP>
__device__ __forceinline__ void ray_begin(const DScene &sc, WaveLDS *w, SP sph, int slot, const RayQ &q, bool prune, bool spheres_done,
                                          HitRec &h, Cnt &cnt) {
    if (!spheres_done) hit_none(h);
    if (!q.active) return;
    const V3 C = q.C, u = q.u;
    if (CNT) cnt.rays++;
    const int n_sph = uni(sc.n_spheres); /* wave-uniform also when read from LDS */
    if (spheres_done) {
        /* spheres_pool has been here */
    } else {
        for (int s = 0; s < n_sph; s++) sphere_test_counted(sph, s, C, u, h);
    }
    if (CNT) { cnt.sph_tests += (uint64_t)h.sph_tests; cnt.sph_nodes += (uint64_t)h.sph_nodes; }
    ray_tables(sc, w, slot, q, prune, h, spheres_done); /* spheres_pool has left the ray and its constants in the tables */
}

/* Sphere::intersect and Scene::intersect's choice (sphere_test above) for a lane's two rays at once and without branches:
 * the two chains of dependent instructions (square root, divisions) do not depend on each other and are issued
 * interleaved, and the sphere is read once. A ray that is not active, a negative discriminant (its root is then a NaN) and
 * t2 < 0 all end in "no update", as the early returns of the reference do. */
template <typename SP>
__device__ __forceinline__ void sphere_test2(SP sph, int s, const RayQ &qa, const RayQ &qb, HitRec &ha, HitRec &hb) {
    const V3 O = ld3(sph[s].O);
    const double R = sph[s].R;
    const int32_t oid = sph[s].object_id;
    const double a = 1;
    const RayQ *const q[2] = {&qa, &qb};
    HitRec *const h[2] = {&ha, &hb};
    double t[2];
    bool hit[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const V3 CO = sub(q[r]->C, O);
        const double b = 2 * dot(q[r]->u, CO);
        const double c = norm2(CO) - R * R;
        const double delta = b * b - 4 * a * c;
        const double sq = sqrt(delta);
        const double t2 = (-b + sq) / (2 * a);
        const double t1 = (-b - sq) / (2 * a);
        t[r] = (t1 > 0) ? t1 : t2;
        hit[r] = q[r]->active & !(delta < 0) & !(t2 < 0);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const bool upd = hit[r] & ((t[r] < h[r]->t) | ((t[r] == h[r]->t) & (oid < h[r]->obj)));
        h[r]->t = upd ? t[r] : h[r]->t;
        h[r]->obj = upd ? oid : h[r]->obj;
        h[r]->prim = upd ? s : h[r]->prim;
        h[r]->mesh = upd ? -1 : h[r]->mesh;
    }
}

/* The pool's table entries of one ray once its closest sphere is known (h). */
/* ray_in_place: the ray, its box-test constants and its tolerance are in the tables already (same ray, same mode) */
__device__ __forceinline__ void ray_tables(const DScene &sc, WaveLDS *w, int slot, const RayQ &q, bool prune, const HitRec &h, bool ray_in_place) {
    if (!q.active || sc.n_meshes == 0) return;
    const V3 C = q.C, u = q.u;
    /* Mesh hits are accepted only below t0 = min(1E9, sphere hit, tmax): TriangleMesh::intersect starts at
     * t = 1E9 (raytracer.cpp:647), a mesh hit beyond the closest sphere loses in Scene::intersect (:938)
     * and one beyond tmax cannot matter to the caller. A mesh hit AT t0 is still recorded (tie rules). */
    double t0 = 1E9;
    if (prune) {
        if (h.t < t0) t0 = h.t;
        if (q.tmax < t0) t0 = q.tmax;
        if (!(t0 > 0)) t0 = 0; /* also turns -0.0 into +0.0: the bit pattern must order like the value */
    }
    if (!ray_in_place) {
        const RayK k = ray_constants(sc, C, u);
        w->ray[0][slot] = C.x; w->ray[1][slot] = C.y; w->ray[2][slot] = C.z;
        w->ray[3][slot] = u.x; w->ray[4][slot] = u.y; w->ray[5][slot] = u.z;
        w->rayf[0][slot] = (float)k.iu.x; w->rayf[1][slot] = (float)k.iu.y; w->rayf[2][slot] = (float)k.iu.z;
        w->rayf[3][slot] = (float)k.nc.x; w->rayf[4][slot] = (float)k.nc.y; w->rayf[5][slot] = (float)k.nc.z;
        const float tf = (float)(k.tol * RT_TOLF_SCALE);
        w->tolf[slot] = (q.anyhit && prune) ? -tf : tf; /* the sign bit of a NaN is kept too */
    }
    w->best_t[slot] = (unsigned long long)__double_as_longlong(t0);
    w->best_pos[slot] = INT32_MAX;
    /* While nothing closer than 1E9 is known, nothing may be skipped by distance: a triangle hit at or beyond
     * 1E9 still makes TriangleMesh::intersect return true (raytracer.cpp:689 sets has_inter before :690 compares). */
    w->thrf[slot] = (prune && t0 < 1E9) ? upper_float(t0 * RT_PRUNE_SLACK) : __builtin_huge_valf();
}

/* ray_begin for both rays of a lane where the sphere list is short (every sphere is tested for every ray) */
template <bool CNT, typename SP>
__device__ __forceinline__ void rays_begin2(const DScene &sc, WaveLDS *w, SP sph, int lane, const RayQ &qa, const RayQ &qb, bool prune, HitRec &ha, HitRec &hb,
                                            Cnt &cnt) {
    hit_none(ha);
    hit_none(hb);
    const int n_sph = uni(sc.n_spheres); /* wave-uniform also when read from LDS */
    for (int s = 0; s < n_sph; s++) sphere_test2(sph, s, qa, qb, ha, hb);
    if (CNT) {
        const uint64_t n = (uint64_t)(qa.active ? 1 : 0) + (uint64_t)(qb.active ? 1 : 0);
        cnt.rays += n;
        cnt.sph_tests += n * (uint64_t)n_sph;
    }
    ray_tables(sc, w, lane, qa, prune, ha);
    ray_tables(sc, w, RT_WAVE + lane, qb, prune, hb);
}

/* The mesh hits of a lane's two rays, and Scene::intersect's choice between each and the sphere hit. After a
 * traversal of the filter tree (own) the winner is first checked against the reference's own tree: the reference
 * reaches a triangle iff the box test passes for its leaf and every box above it (raytracer.cpp:644-665), those boxes
 * contain the leaf's box coordinate by coordinate, and the slab test is monotone in the box (each of its operations
 * is), so "the leaf's box is certainly hit" settles all of them. Anything less - a grazed or flat leaf box, a hit
 * without a triangle - sets the ray's `doubt`: it has to be traced again on the reference's tree.
 * Both rays go through the same straight-line code so that their loads (triangle, leaf index; then leaf box and
 * object id) are in flight together; a ray without a triangle reads entry 0 and is masked. */
__device__ __forceinline__ void rays_end(const DScene &sc, WaveLDS *w, int lane, const RayQ &qa, const RayQ &qb, bool own, HitRec &ha, HitRec &hb,
                                         bool &doubt_a, bool &doubt_b, Cnt *cnt = nullptr) {
    const RayQ *const q[2] = {&qa, &qb};
    HitRec *const h[2] = {&ha, &hb};
    bool doubt[2] = {false, false};
    int32_t pos[2], mesh[2], leaf[2] = {0, 0}, oid[2], group = 0;
    double tm[2];
    bool live[2], ph[2], tr[2];
    Tri3 T[2];
    float lb[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int slot = lane + r * RT_WAVE;
        pos[r] = w->best_pos[slot];
        tm[r] = __longlong_as_double((long long)w->best_t[slot]);
        live[r] = q[r]->active && pos[r] != INT32_MAX;
        ph[r] = live[r] && pos[r] >= RT_PHANTOM_BASE; /* the mesh reported a hit at t = 1E9 without a triangle */
        tr[r] = live[r] && !ph[r];
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t p0 = tr[r] ? (uint32_t)pos[r] : 0u;
        T[r] = tri_load(in_global(sc.tris), p0);
        mesh[r] = in_global(sc.tris)[p0].mesh;
#if RT_TRI_FIN
        if (own) { /* everything else about the triangle in the same round trip */
            const auto *f = in_global(sc.tri_fin) + p0;
            for (int a = 0; a < 6; a++) lb[r][a] = f->box[a];
            oid[r] = f->object_id;
            if (r == 0) group = f->group;
            if (ph[r]) oid[r] = in_global(sc.meshes)[pos[r] - RT_PHANTOM_BASE].object_id;
        } else
#endif
        {
            if (r == 0) group = in_global(sc.attrs)[p0].group; /* only a path segment's hit is shaded */
            if (own) leaf[r] = in_global(sc.tri_leaf)[p0];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (!tr[r]) mesh[r] = ph[r] ? pos[r] - RT_PHANTOM_BASE : 0;
        if (RT_TRI_FIN && own) continue;
        oid[r] = in_global(sc.meshes)[mesh[r]].object_id;
        if (own) {
            const float *b = in_global(sc.leaf_box32) + (size_t)leaf[r] * 6;
            for (int a = 0; a < 6; a++) lb[r][a] = b[a];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (own) {
            if (ph[r]) doubt[r] = true;
            if (tr[r]) {
                const RayK k = ray_constants(sc, q[r]->C, q[r]->u); /* the double-precision form: as few rays as possible should go round again */
                bool hit, unsure;
                double tlo;
                box_fast(Box6{(double)lb[r][0], (double)lb[r][1], (double)lb[r][2], (double)lb[r][3], (double)lb[r][4], (double)lb[r][5]}, k, hit, unsure, tlo);
                if (!hit) doubt[r] = true;
                if (cnt) cnt->box_exact++; /* a double-precision box test, like the reference's own */
            }
        }
        if (live[r] && !doubt[r]) {
            const double tmm = ph[r] ? 1E9 : tm[r];
            if (tmm < h[r]->t || (tmm == h[r]->t && oid[r] < h[r]->obj)) {
                h[r]->t = tmm; h[r]->obj = oid[r]; h[r]->prim = ph[r] ? -1 : pos[r]; h[r]->mesh = mesh[r];
                if (r == 0) h[r]->group = group;
                double lt;
                if (!ph[r]) (void)tri_eval(T[r], q[r]->C, q[r]->u, h[r]->beta, h[r]->gamma, lt); /* alpha, beta, gamma of the winner (:683-685) */
                if (cnt && sc.count_own && !ph[r]) cnt->tris++; /* the production count includes this second evaluation */
            }
        }
    }
    doubt_a = doubt[0];
    doubt_b = doubt[1];
}

/* the mesh root box of one ray (raytracer.cpp:644); k and thr are the ray's constants as read at the start of the feed */
template <bool CNT>
__device__ __forceinline__ bool root_test(const DScene &sc, const WaveLDS *w, const RootRec &m, int mesh, bool active, int slot, const RayKf &k, float thr, float &fn, Cnt &cnt) {
    if (!active) return false;
    if (CNT) cnt.box_tests++;
    bool pass, unsure;
    float tn;
    box_fast_f(m.box[0], m.box[1], m.box[2], m.box[3], m.box[4], m.box[5], k, pass, unsure, tn);
    if (__builtin_expect(unsure, 0)) { /* the ray as ray_begin stored it */
        pass = box_hit_exact(sc.meshes[mesh].root_box, w->ray[0][slot], w->ray[1][slot], w->ray[2][slot], w->ray[3][slot], w->ray[4][slot], w->ray[5][slot]);
        tn = -__builtin_huge_valf();
        if (CNT) cnt.box_exact++;
    }
    if (CNT && pass) cnt.mesh_calls++;
    fn = lower_float(tn);
    return pass && !(tn > thr);
}

/* a ray's constants of the fast box test, back from the pool's tables */
__device__ __forceinline__ RayKf ray_k_from_lds(const WaveLDS *w, int slot) {
    RayKf k;
    k.iux = w->rayf[0][slot]; k.iuy = w->rayf[1][slot]; k.iuz = w->rayf[2][slot];
    k.ncx = w->rayf[3][slot]; k.ncy = w->rayf[4][slot]; k.ncz = w->rayf[5][slot];
    k.tol = __builtin_fabsf(w->tolf[slot]);
    return k;
}

/* A leaf item of the pool: x = first primitive (25 bits) | count - 1 in bits 28-30 (bit 31 is ignored: a leaf code of the
 * filter tree can be stored as it is), y = the entry parameter of the leaf's box as float bits with the ray slot in place of
 * the seven lowest mantissa bits. Dropping those bits lowers a positive parameter (it is a lower bound anyway) and leaves a
 * negative one negative; the parameter is only ever compared with the ray's threshold, which is never negative. One 8-byte
 * LDS store per pushed leaf. */
#define RT_LEAF_FIRST_MASK 0x01FFFFFFu
static_assert(RT_RAY_SLOTS <= 128 && RT_NODE_BITS == 25, "leaf items: seven bits for the ray slot, 25 for the first primitive");
__device__ __forceinline__ uint32_t leaf_word(float t, uint32_t slot) { return (__float_as_uint(t) & ~0x7Fu) | slot; }
__device__ __forceinline__ uint2 leaf_item(uint32_t first, uint32_t count, float t, uint32_t slot) { return make_uint2(first | ((count - 1u) << 28), leaf_word(t, slot)); }
__device__ __forceinline__ uint32_t leaf_first(uint2 it) { return it.x & RT_LEAF_FIRST_MASK; }
__device__ __forceinline__ int leaf_count(uint2 it) { return (int)((it.x >> 28) & 7u) + 1; }
__device__ __forceinline__ int leaf_slot(uint2 it) { return (int)(it.y & 0x7Fu); }
__device__ __forceinline__ float leaf_entry(uint2 it) { return __uint_as_float(it.y & ~0x7Fu); }

/* One record of the filter tree for ray slot rl: a child is pushed unless the ray certainly misses its box (box_fast_f's
 * miss; the filter's boxes are rounded outwards) or the box starts beyond the ray's threshold. Called by all 64 lanes;
 * have_mask: the lanes that hold a record. item_t is the entry parameter the record was pushed with. */
/* NC: the children a lane tests - all four of its record, or two when two lanes share a record (bx: their 6 NC box
 * coordinates, code: their codes). */
/* W: the wave's LDS structure - WaveLDS here; the single-precision kernel (rt_fast.hip.h) has tables of its own with the same
 * members for the box test (rayf, tolf, thrf) and the two stacks (node, leaf) */
template <int NC, typename W>
__device__ __forceinline__ void children_step(const DScene &sc, W *w, int lane, unsigned long long have_mask, int rl, float item_t, const float (&bx)[6 * NC],
                                              const uint32_t (&code)[NC], int &n_nodes, int &n_leaves) {
    const float thrf = w->thrf[rl];
    const float iux = w->rayf[0][rl], iuy = w->rayf[1][rl], iuz = w->rayf[2][rl];
    const float ncx = w->rayf[3][rl], ncy = w->rayf[4][rl], ncz = w->rayf[5][rl];
    const float tolf = __builtin_fabsf(w->tolf[rl]); /* NaN: then nothing is ever certain */
    /* The lanes' yes/no answers are kept as wave masks (one compare each, combined by scalar instructions) and turned back
     * into lane predicates only where something is stored: the ballot of a combined lane condition costs two vector
     * instructions more than the ballot of a plain compare. */
    const unsigned long long bh = have_mask & wave_ballot(!(item_t > thrf)); /* not fallen behind the threshold since it was pushed */
    unsigned long long mn[NC], ml[NC];
    float f[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float t1x = __builtin_fmaf(bx[6 * c], iux, ncx), t2x = __builtin_fmaf(bx[6 * c + 3], iux, ncx);
        const float t1y = __builtin_fmaf(bx[6 * c + 1], iuy, ncy), t2y = __builtin_fmaf(bx[6 * c + 4], iuy, ncy);
        const float t1z = __builtin_fmaf(bx[6 * c + 2], iuz, ncz), t2z = __builtin_fmaf(bx[6 * c + 5], iuz, ncz);
        const float tMax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
        const float tMin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
        /* certain miss: tMax - tMin < -tolf or tMax < -tolf (a NaN on either side never says "miss") */
        const unsigned long long not_missed = wave_ballot(!(fminf(tMax, tMax - tMin) < -tolf));
        const float tlo = tMin - tolf;
        const unsigned long long go = bh & not_missed & wave_ballot(!(tlo > thrf));
        /* child codes: record (top bit clear), leaf (top bit set), empty (all ones) */
        const unsigned long long is_rec = wave_ballot((int32_t)code[c] >= 0), not_empty = wave_ballot(code[c] != RT_CHILD_EMPTY);
        mn[c] = go & is_rec;
        ml[c] = go & ~is_rec & not_empty;
        f[c] = tlo; /* already below the real entry parameter: the tolerance exceeds the error of tMin by 1.7e-7 M, the rounding of
                       the subtraction is 6e-8 M at most */
    }
    /* Places on the stacks: all first children, then all second children, ... - the lane's place among them by mbcnt,
     * the offsets by scalar popcounts. No prefix sum over the lanes is needed. */
    int jn[NC], jl[NC];
    int tn = n_nodes, tl = n_leaves;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        jn[c] = tn + rank_in(mn[c]);
        jl[c] = tl + rank_in(ml[c]);
        tn += __popcll(mn[c]);
        tl += __popcll(ml[c]);
    }
    if (tn > sc.node_cap || tl > RT_LEAF_CAP) { /* cannot happen for a validated scene; never write past the stacks */
        if (lane == 0) atomicOr(sc.overflow, 1);
    } else {
#if RT_PUSH_BRANCHLESS
        uint2 *const dump = reinterpret_cast<uint2 *>(&w->mark[0]) + (lane & 31); /* mark / ex: 512 bytes that only the prefix-sum triangle step uses, never alive across a node step */
#pragma unroll
        for (int c = 0; c < NC; c++) {
            uint2 *pn = __builtin_amdgcn_inverse_ballot_w64(mn[c]) ? &w->node[jn[c]] : dump;
            uint2 *pl = __builtin_amdgcn_inverse_ballot_w64(ml[c]) ? &w->leaf[jl[c]] : dump;
            *pn = make_uint2(code[c] | ((uint32_t)rl << RT_NODE_BITS), __float_as_uint(f[c]));
            *pl = make_uint2(code[c], leaf_word(f[c], (uint32_t)rl));
        }
#else
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (__builtin_amdgcn_inverse_ballot_w64(mn[c])) w->node[jn[c]] = make_uint2(code[c] | ((uint32_t)rl << RT_NODE_BITS), __float_as_uint(f[c]));
            if (__builtin_amdgcn_inverse_ballot_w64(ml[c])) w->leaf[jl[c]] = make_uint2(code[c], leaf_word(f[c], (uint32_t)rl));
        }
#endif
        n_nodes = tn;
        n_leaves = tl;
    }
}
template <typename W>
__device__ __forceinline__ void wide_children(const DScene &sc, W *w, int lane, unsigned long long have_mask, int rl, float item_t, const float4 &q0, const float4 &q1,
                                              const float4 &q2, const float4 &q3, const float4 &q4, const float4 &q5, const uint4 &ch, int &n_nodes, int &n_leaves) {
    const float bx[24] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w,
                          q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
    const uint32_t code[4] = {ch.x, ch.y, ch.z, ch.w};
    children_step<4>(sc, w, lane, have_mask, rl, item_t, bx, code, n_nodes, n_leaves);
}

/* The NC = 4 / LPR children that lane `sub` of a record's LPR lanes tests: 6 NC consecutive box coordinates and NC codes,
 * fetched with the widest loads that fit (one lane: the whole 128-byte record). */
template <int NC>
__device__ __forceinline__ void load_children(const NodeRec4 *rec, int sub, float (&bx)[6 * NC], uint32_t (&code)[NC]) {
    if constexpr (NC == 4) {
        const float4 *rp = reinterpret_cast<const float4 *>(rec);
        const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3], q4 = rp[4], q5 = rp[5];
        const uint4 ch = *reinterpret_cast<const uint4 *>(rp + 6);
        const float b[24] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w,
                             q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
#pragma unroll
        for (int a = 0; a < 24; a++) bx[a] = b[a];
        code[0] = ch.x; code[1] = ch.y; code[2] = ch.z; code[3] = ch.w;
    } else if constexpr (NC == 2) {
        const float4 *hp = reinterpret_cast<const float4 *>(rec) + 3 * sub;
        const float4 h0 = hp[0], h1 = hp[1], h2 = hp[2];
        const uint2 hc = reinterpret_cast<const uint2 *>(reinterpret_cast<const float4 *>(rec) + 6)[sub];
        const float b[12] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w};
#pragma unroll
        for (int a = 0; a < 12; a++) bx[a] = b[a];
        code[0] = hc.x; code[1] = hc.y;
    } else {
        static_assert(NC == 1, "one, two or four lanes per record");
        const float2 *cp = reinterpret_cast<const float2 *>(rec) + 3 * sub;
        const float2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
        bx[0] = c0.x; bx[1] = c0.y; bx[2] = c1.x; bx[3] = c1.y; bx[4] = c2.x; bx[5] = c2.y;
        code[0] = reinterpret_cast<const uint32_t *>(reinterpret_cast<const float4 *>(rec) + 6)[sub];
    }
}

#if RT_TOP_LDS
/* the same from the staged copy of records 1 .. 4 (rec_m1 = record - 1, in 0 .. 3) */
template <int NC, typename W>
__device__ __forceinline__ void load_children_lds(const W *w, uint32_t rec_m1, int sub, float (&bx)[6 * NC], uint32_t (&code)[NC]) {
    const float *r = w->top[rec_m1];
#pragma unroll
    for (int a = 0; a < 6 * NC; a++) bx[a] = r[sub * 6 * NC + a];
#pragma unroll
    for (int a = 0; a < NC; a++) code[a] = __float_as_uint(r[24 + sub * NC + a]);
}
#endif

/* One node step on a tree of four-child records (the filter tree, or the tree over a long sphere list): the top k node
 * items are popped, LPR lanes per record (k <= 64 / LPR), each lane tests 4 / LPR children of its record and pushes those
 * that its ray may hit (children_step). Branch-free up to the stores, so that all loads of the step are in flight before
 * the first is waited for: a lane without a record reads item 0 / record 0 and is masked out. */
template <bool CNT, bool SPHP, int LPR, typename W>
__device__ __forceinline__ void own_node_step(const DScene &sc, W *w, const NodeRec4 *tree4, uint32_t n_recs, int lane, int k, int &n_nodes, int &n_leaves,
                                              Cnt &cnt, uint64_t &st0) {
    constexpr int NC = 4 / LPR;
    constexpr int SH = LPR == 4 ? 2 : (LPR == 2 ? 1 : 0);
    const int li = lane >> SH, sub = lane & (LPR - 1);
    const unsigned long long in_k = wave_ballot(li < k);
    const uint2 it = w->node[__builtin_amdgcn_inverse_ballot_w64(in_k) ? n_nodes - 1 - li : 0];
    n_nodes -= k;
    wave_sync();
    const int rl = (int)(it.x >> RT_NODE_BITS);
    uint32_t rec = it.x & RT_LOW_MASK;
    if (CNT && RT_STAMP_STEPS && LPR == 1) { asm volatile("" :: "v"(rec)); const uint64_t t = RT_STAMP(); cnt.dbg[0] += t - st0; st0 = t; } /* the item has arrived */
    /* never fetch outside the arrays: the lanes' answers as one mask, so that the step has no branch that is taken unless
     * something is wrong */
    const unsigned long long outside = in_k & wave_ballot(rec >= n_recs);
    if (__builtin_expect(outside != 0ull, 0)) {
        if (lane == 0) atomicOr(sc.overflow, 2);
    }
    const unsigned long long have_mask = in_k & ~outside;
    const bool have = __builtin_amdgcn_inverse_ballot_w64(have_mask);
    if (!have) rec = 0;
    if (CNT && have) {
        if (SPHP) cnt.sph_nodes += NC; /* box tests of this lane */
        else { cnt.box_tests += NC; cnt.nodes += (sub == 0) ? 1 : 0; }
    }
    float bx[6 * NC];
    uint32_t code[NC];
#if RT_TOP_LDS
    /* the first steps of a query pop nothing but children of the root: records 1 .. 4, staged in LDS */
    if (!SPHP && sc.n_own_nodes >= 5u && (have_mask & ~wave_ballot(rec - 1u < 4u)) == 0ull) load_children_lds<NC>(w, (rec - 1u) & 3u, sub, bx, code);
    else
#endif
    load_children<NC>(tree4 + rec, sub, bx, code);
    if (CNT && RT_STAMP_STEPS && LPR == 1) { /* the record has arrived */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t = RT_STAMP(); cnt.dbg[1] += t - st0; st0 = t;
    }
#ifdef RT_DIAG_STALE /* diagnostic: popped node items / those that had fallen behind their ray's threshold */
    if (CNT && !SPHP && LPR == 1) { cnt.dbg[1] += __popcll(__ballot(have)); cnt.dbg[0] += __popcll(__ballot(have && __uint_as_float(it.y) > w->thrf[rl])); }
#endif
    children_step<NC>(sc, w, lane, have_mask, rl, __uint_as_float(it.y), bx, code, n_nodes, n_leaves);
    wave_sync();
    if (CNT) {
        const uint64_t t = RT_STAMP();
        if (SPHP && !RT_STAMP_STEPS) { cnt.dbg[0] += t - st0; cnt.dbg[1]++; cnt.dbg[2] += (uint64_t)k; } /* the sphere tree's share of the node steps */
        cnt.cyc_node += t - st0; st0 = t; cnt.node_steps++; cnt.node_lanes += (uint64_t)k;
    }
}

/* The pool traversal of one query: everything between "the rays' tables are in LDS" (ray_begin) and "the rays' best
 * triangles are in LDS" (rays_end). It takes nothing per lane but the two activity masks and leaves nothing per lane
 * behind: all of its inputs and outputs are the wave's LDS tables. */
/* SPHP: the same traversal over the tree of a long sphere list (own must be true): its leaves are ranges of spheres, the
 * leaf step tests 64 (ray, sphere) pairs with Sphere::intersect's arithmetic, and the per-ray minimum is kept under
 * Scene::intersect's rule for spheres (smallest t, then smallest objectId) in the same two tables. */
template <bool CNT, bool SPHP = false>
__device__ __forceinline__ void pool_traverse(const DScene &sc, const DScene &cold, WaveLDS *w, const RootRec *roots, bool own, bool prune,
                                              unsigned long long mask_a, unsigned long long mask_b, Cnt &cnt, uint64_t &st0) {
    const int lane = lane_id();
    const bool act_a = (mask_a >> lane) & 1ull, act_b = (mask_b >> lane) & 1ull;
    const NodeRec32 *const recs = sc.rec32;
    const NodeRec4 *const tree4 = SPHP ? sc.sph_rec4 : sc.own_rec4;
    const rt_tri *const tri_base = own ? sc.own_tris : sc.tris;
    const uint32_t n_recs = SPHP ? sc.n_sph_rec4 : (own ? sc.n_own_nodes : sc.n_nodes_total);
    const uint32_t n_tri_lim = SPHP ? (uint32_t)(sc.n_spheres - sc.n_sph_big) : (own ? sc.n_own_tris : sc.n_tris_total);
    const int node_soft = sc.node_soft, feed_below = sc.node_soft - RT_FEED_MARGIN;
    const int n_feed = own ? 1 : sc.n_meshes;

    /* lanes per leaf item of a triangle step: as many as the longest leaf of the tree in use has triangles (two or four) */
    const int quad_shift = (own && RT_OWN_LEAF <= RT_LEAF_QUAD) ? sc.own_leaf_shift : 2;
    const int leaf_trigger = RT_LEAF_TRIGGER ? RT_LEAF_TRIGGER : (RT_WAVE >> quad_shift); /* one full step */
    int n_nodes = 0, n_leaves = 0; /* wave-uniform stack heights */
    int mi = 0;
    unsigned steps = 0;
    for (;;) {
        /* Root boxes (raytracer.cpp:644): every lane for its own rays; roots that pass enter the pool. The
         * next mesh is fed whenever fewer than node_soft - RT_FEED_MARGIN node items are pending, so the traversals of
         * the meshes share their steps, and a root is already tested against the distance of the hits
         * found so far. */
        if (own && mi == 0) {
            /* The filter tree has no root box: every active ray starts at record 0, whose children are tested here
             * right away (the record is read with wave-uniform addresses). The second rays go the same way if their
             * pushes are sure to fit; otherwise they enter as items for record 0 and take an ordinary step. */
            mi = 1;
            const float4 *rp = reinterpret_cast<const float4 *>(tree4);
            const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3], q4 = rp[4], q5 = rp[5];
            const uint4 ch = *reinterpret_cast<const uint4 *>(rp + 6);
            const float ninf = -__builtin_huge_valf();
            wide_children(sc, w, lane, mask_a, lane, ninf, q0, q1, q2, q3, q4, q5, ch, n_nodes, n_leaves);
            if (CNT && act_a) { if (SPHP) cnt.sph_nodes += 4; else { cnt.box_tests += 4; cnt.nodes++; } }
            const unsigned long long bb = __ballot(act_b);
            const int nb = __popcll(bb);
            if (n_nodes + 4 * nb <= node_soft && n_leaves + 4 * nb <= RT_LEAF_CAP) {
                wide_children(sc, w, lane, mask_b, RT_WAVE + lane, ninf, q0, q1, q2, q3, q4, q5, ch, n_nodes, n_leaves);
                if (CNT && act_b) { if (SPHP) cnt.sph_nodes += 4; else { cnt.box_tests += 4; cnt.nodes++; } }
            } else {
                if (act_b) w->node[n_nodes + rank_in(bb)] = make_uint2((uint32_t)(RT_WAVE + lane) << RT_NODE_BITS, __float_as_uint(ninf));
                n_nodes += nb;
            }
            wave_sync();
        }
        if (mi < n_feed && n_nodes < feed_below && n_leaves <= RT_WAVE) {
            const RayKf ka = ray_k_from_lds(w, lane), kb = ray_k_from_lds(w, RT_WAVE + lane);
            const float thr_a = w->thrf[lane], thr_b = w->thrf[RT_WAVE + lane]; /* as of now: thresholds only fall */
            do {
                const RootRec &m = roots[mi];
                float fa = 0.f, fb = 0.f;
                const bool pa = root_test<CNT>(cold, w, m, mi, act_a, lane, ka, thr_a, fa, cnt);
                const bool pb = root_test<CNT>(cold, w, m, mi, act_b, RT_WAVE + lane, kb, thr_b, fb, cnt);
                mi++;
                const unsigned long long ba = __ballot(pa), bb = __ballot(pb);
                if ((ba | bb) == 0ull) continue;
                const int ca = __popcll(ba);
                if (m.root_count == 0) {
                    if (pa) w->node[n_nodes + rank_in(ba)] = make_uint2((uint32_t)m.root | ((uint32_t)lane << RT_NODE_BITS), __float_as_uint(fa));
                    if (pb) w->node[n_nodes + ca + rank_in(bb)] = make_uint2((uint32_t)m.root | ((uint32_t)(RT_WAVE + lane) << RT_NODE_BITS), __float_as_uint(fb));
                    n_nodes = uni(n_nodes + ca + __popcll(bb));
                } else { /* the root is itself a leaf */
                    if (pa) {
                        const int qi = n_leaves + rank_in(ba);
                        w->leaf[qi] = leaf_item((uint32_t)m.root, (uint32_t)m.root_count, fa, (uint32_t)lane);
                    }
                    if (pb) {
                        const int qi = n_leaves + ca + rank_in(bb);
                        w->leaf[qi] = leaf_item((uint32_t)m.root, (uint32_t)m.root_count, fb, (uint32_t)(RT_WAVE + lane));
                    }
                    n_leaves = uni(n_leaves + ca + __popcll(bb));
                }
            } while (mi < n_feed && n_nodes < feed_below && n_leaves <= RT_WAVE);
            wave_sync();
            if (CNT) { uint64_t t = RT_STAMP(); cnt.cyc_roots += t - st0; st0 = t; }
        }
        if (n_nodes + n_leaves == 0) break; /* then every mesh has been fed: the feed runs whenever the pool is empty */
        if (++steps > RT_STEP_LIMIT) { /* a traversal visits every record and triangle at most once per ray: far more steps mean corrupt items */
            if (lane == 0) atomicOr(sc.overflow, 8);
            break;
        }
        /* one pool step: up to 64 items */
        {
            const bool do_leaf = (n_leaves >= (SPHP ? RT_SPH_TRIGGER : leaf_trigger)) || (n_nodes == 0) ||
                                 (n_leaves + (own ? 4 : 2) * RT_WAVE > RT_LEAF_CAP);
            if (do_leaf) {
                /* A triangle step tests up to 64 (ray, triangle) pairs. Three ways from leaf items to lanes:
                 * - RT_LEAF_QUAD lanes per item when no leaf of the tree in use holds more triangles than that: lane r
                 *   of an item's group takes its r-th triangle (the default for triangles);
                 * - one lane per item when every leaf is one primitive (the sphere tree);
                 * - otherwise 64 triangles whichever leaves they sit in: the top m <= 64 leaf items are (first, count)
                 *   ranges; a prefix sum of the counts gives every item its slots, slot s (= lane s) finds its item
                 *   through a marker + running maximum, and tests triangle first + (s - first slot of the item). Items
                 *   used up are popped, the one cut by slot 64 is shortened in place and becomes the new top.
                 * A leaf that has fallen behind its ray's threshold since it was pushed counts as empty. */
                const int m = n_leaves < RT_WAVE ? n_leaves : RT_WAVE;
                uint2 it = make_uint2(0u, 0u);
                int cntl = 0;
                const bool quad = RT_LEAF_QUAD && !SPHP && (own ? RT_OWN_LEAF : RT_EXT_LEAF) <= RT_LEAF_QUAD;
                if (!quad && lane < m) {
                    it = w->leaf[n_leaves - 1 - lane];
                    cntl = leaf_count(it);
                    if (leaf_entry(it) > w->thrf[leaf_slot(it)]) cntl = 0;
                }
                bool have;
                uint32_t tri;
                int rl, take;
#if RT_LEAF_QUAD
                if (quad) {
                    const int mq = n_leaves < (RT_WAVE >> quad_shift) ? n_leaves : (RT_WAVE >> quad_shift);
                    const int iq = lane >> quad_shift, rq = lane & ((1 << quad_shift) - 1);
                    uint2 itq = make_uint2(0u, 0u);
                    int cq = 0;
                    if (iq < mq) {
                        itq = w->leaf[n_leaves - 1 - iq];
                        cq = leaf_count(itq);
                        if (leaf_entry(itq) > w->thrf[leaf_slot(itq)]) cq = 0;
                    }
#ifdef RT_DIAG_STALE
                    if (CNT && !SPHP) { cnt.dbg[3] += __popcll(__ballot(iq < mq && rq == 0)); cnt.dbg[2] += __popcll(__ballot(iq < mq && rq == 0 && cq == 0)); }
#endif
                    have = rq < cq;
                    tri = leaf_first(itq) + (uint32_t)rq;
                    rl = leaf_slot(itq);
                    take = mq << quad_shift;
                    n_leaves -= mq;
                } else
#endif
                if ((SPHP ? RT_SPH_LEAF : RT_OWN_LEAF) == 1 && own) {
                    /* every leaf of the filter trees is one primitive: an item is a slot, nothing to spread out */
                    have = cntl > 0;
                    tri = leaf_first(it);
                    rl = leaf_slot(it);
                    take = m;
                    n_leaves -= m;
                } else {
                const int incl = wave_scan_add(cntl);
                const int excl = incl - cntl;
                const int total = __builtin_amdgcn_readlane(incl, 63);
                take = total < RT_WAVE ? total : RT_WAVE;
                w->mark[lane] = -1;
                wave_sync();
                if (cntl > 0 && excl < RT_WAVE) w->mark[excl] = lane;
                w->ex[lane] = excl;
                wave_sync();
                const int j = wave_scan_max(w->mark[lane]);
                have = lane < take; /* then j >= 0: slot 0 carries the marker of the first non-empty item */
                uint2 itj = make_uint2(0u, 0u);
                int exj = 0;
                if (have) { itj = w->leaf[n_leaves - 1 - j]; exj = w->ex[j]; }
                tri = leaf_first(itj) + (uint32_t)(lane - exj);
                rl = leaf_slot(itj);
                wave_sync();
                if (lane < m && excl < RT_WAVE && incl > RT_WAVE) { /* the item cut by the last slot */
                    const uint32_t used = (uint32_t)(RT_WAVE - excl);
                    w->leaf[n_leaves - 1 - lane] = make_uint2((leaf_first(it) + used) | (((uint32_t)cntl - used - 1u) << 28), it.y);
                }
                n_leaves = uni(n_leaves - __popcll(__ballot(lane < m && incl <= RT_WAVE)));
                }
                const int k = take;
                bool acc = false;
                double beta, gamma, lt = 0;
                bool ok = have;
                if (__builtin_expect(wave_ballot(have && tri >= n_tri_lim) != 0ull, 0)) { /* never fetch outside the arrays */
                    if (have && tri >= n_tri_lim) {
                        atomicOr(sc.overflow, 2);
                        ok = false;
                    }
                }
                /* all coordinate loads and the ray's LDS reads are issued before anything is waited for:
                 * unconditional (a lane without a triangle reads triangle 0), fenced, then pinned */
                Tri3 tv;
                int32_t key = 0; /* what the per-ray minimum is broken ties with: position in the reference's order / objectId */
                V3 rc = mk(w->ray[0][rl], w->ray[1][rl], w->ray[2][rl]);
                V3 ru = mk(w->ray[3][rl], w->ray[4][rl], w->ray[5][rl]);
                if (SPHP) {
                    /* Sphere::intersect, raytracer.cpp:883-918, on sphere n_sph_big + tri of the list; a hit counts below
                     * Scene::intersect's initial t = 1E10 (:932, strict '<' at :938) */
                    const rt_sphere *sp = in_global(sc.spheres) + (size_t)sc.n_sph_big + (ok ? tri : 0u);
                    const V3 O = ld3(sp->O);
                    const double R = sp->R;
                    key = sp->object_id;
                    if (ok) {
                        if (CNT) cnt.sph_tests++;
                        const double a = 1;
                        const V3 CO = sub(rc, O);
                        const double b = 2 * dot(ru, CO);
                        const double c = norm2(CO) - R * R;
                        const double delta = b * b - 4 * a * c;
                        if (!(delta < 0)) {
                            const double sq = sqrt(delta);
                            const double t2 = (-b + sq) / (2 * a);
                            if (!(t2 < 0)) {
                                const double t1 = (-b - sq) / (2 * a);
                                lt = (t1 > 0) ? t1 : t2;
                                acc = lt < 1E10;
                                if (lt == 0) lt = 0.0; /* -0.0: the bit pattern must order like the value */
                            }
                        }
                    }
                    (void)beta; (void)gamma; (void)tv;
                } else {
                tv = tri_load(tri_base, ok ? tri : 0u);
                asm volatile("" ::: "memory");
                pin3(tv.A); pin3(tv.e1); pin3(tv.e2); pin3(rc); pin3(ru);
                if (CNT && RT_STAMP_STEPS) {
                    uint64_t t = RT_STAMP(); cnt.dbg[4] += t - st0;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    uint64_t t2 = RT_STAMP(); cnt.dbg[5] += t2 - t;
                }
                if (ok) {
                    if (CNT) cnt.tris++;
                    acc = tri_eval(tv, rc, ru, beta, gamma, lt);
                    if (__builtin_expect(acc && !(lt < 1E9), 0)) {
                        /* raytracer.cpp:689-690: has_inter is set, t stays 1E9 and P, N, colour stay as the caller
                         * initialised them. Remembered as the lowest such mesh; any triangle below 1E9 overrides it. */
                        acc = false;
                        atomicMin(&w->best_pos[rl], RT_PHANTOM_BASE + (own ? 0 : sc.tris[tri].mesh));
                    }
                }
                key = own ? tv.id : (int32_t)tri;
                }
                if (__ballot(acc) != 0ull) {
                    /* keep (min t, then min position) per ray. LDS executes one wave's instructions in
                     * order, so every lane's atomic of one statement is done before the next statement's
                     * reads: no barrier is needed inside a wave, only the compiler must keep the order. */
                    const unsigned long long tb = (unsigned long long)__double_as_longlong(lt);
                    unsigned long long old = 0;
                    if (acc) old = atomicMin(&w->best_t[rl], tb);
                    wave_sync();
                    bool mine = false;
                    if (acc) {
                        const unsigned long long cur = w->best_t[rl];
                        mine = (tb == cur);
                        if (mine && old > cur) { /* this lane lowered the minimum */
                            w->best_pos[rl] = INT32_MAX;
                            if (prune) w->thrf[rl] = (__float_as_int(w->tolf[rl]) < 0) ? 0.f : upper_float(lt * RT_PRUNE_SLACK);
                        }
                    }
                    wave_sync();
                    if (mine) atomicMin(&w->best_pos[rl], key);
                }
                wave_sync();
                if (CNT) { uint64_t t = RT_STAMP(); cnt.cyc_leaf += t - st0; st0 = t; cnt.leaf_steps++; cnt.leaf_lanes += (uint64_t)k; }
            } else if (own) {
                /* Filter tree: up to 64 four-child records per step, one per lane. A child is pushed unless the
                 * ray certainly misses its box or the box starts beyond the ray's threshold. A step pops k
                 * records and pushes at most 4 k, so k is held to a third of the room below node_soft; above
                 * it one record per step (depth-first: three more entries per level at most). */
                int k = n_nodes < RT_WAVE ? n_nodes : RT_WAVE;
                const int room = (node_soft - n_nodes) / 3;
                if (k > room) k = room > 1 ? room : 1;
                if (RT_NARROW_SPLIT > RT_WAVE / 2 && k > RT_WAVE / 2 && k <= RT_NARROW_SPLIT) k = RT_WAVE / 2;
                /* Half a step's worth of records or fewer: two lanes share a record, each tests two of its children (their
                 * boxes are 48 consecutive bytes, their codes 8); a quarter or fewer: four lanes, one child each (24 + 4
                 * bytes) - half or a quarter of the arithmetic and pushes per lane in the steps that end a query. */
                if (RT_NARROW_STEP > 1 && k <= RT_WAVE / 4) own_node_step<CNT, SPHP, 4>(sc, w, tree4, n_recs, lane, k, n_nodes, n_leaves, cnt, st0);
                else if (RT_NARROW_STEP > 0 && k <= RT_WAVE / 2) own_node_step<CNT, SPHP, 2>(sc, w, tree4, n_recs, lane, k, n_nodes, n_leaves, cnt, st0);
                else own_node_step<CNT, SPHP, 1>(sc, w, tree4, n_recs, lane, k, n_nodes, n_leaves, cnt, st0);
            } else {
                /* Up to 64 BVH records per step, one per lane: both child boxes are tested (raytracer.cpp:656-665)
                 * and the children that are hit are pushed. 64 records are popped while the stack is below
                 * node_soft (a step can add one entry per record popped); above it one record per step,
                 * a plain depth-first descent that needs at most max_depth + 1 entries more.
                 * Branch-free up to the box decisions, so that all loads of the step are in flight before the
                 * first is waited for: a lane without a record reads slot 0 / record 0 and is masked out. */
                int k = n_nodes < RT_WAVE ? n_nodes : RT_WAVE;
                const int room = node_soft - n_nodes;
                if (k > room) k = room > 1 ? room : 1;
                bool have = lane < k;
                const uint2 it = w->node[have ? n_nodes - 1 - lane : 0];
                n_nodes -= k;
                wave_sync();
                const int rl = (int)(it.x >> RT_NODE_BITS);
                uint32_t rec = it.x & RT_LOW_MASK;
                if (have && rec >= n_recs) { /* never fetch outside the arrays */
                    have = false;
                    atomicOr(sc.overflow, 2);
                }
                if (!have) rec = 0;
                const float4 *rp = reinterpret_cast<const float4 *>(recs + rec);
                const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
                const float thr = w->thrf[rl];
                const RayKf rk = ray_k_from_lds(w, rl);
                have = have && !(__uint_as_float(it.y) > thr); /* fallen behind the threshold since it was pushed */
                const bool ext = rec >= sc.n_ref_nodes; /* a filter of this implementation, not a record of the reference */
                if (CNT && have && (!ext || sc.count_own)) { cnt.nodes++; cnt.box_tests += 2; } /* the reference has no extension records */
                int32_t L = __float_as_int(q3.x), R = __float_as_int(q3.y);
                const int32_t nL = __float_as_int(q3.z), nR = __float_as_int(q3.w);
                bool hl, hr, ul, ur;
                float tl, tr;
                box_fast_f(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, rk, hl, ul, tl);
                box_fast_f(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, rk, hr, ur, tr);
                if (ext) { /* kept out only by a certain miss; with RT_F_NO_PRUNE every triangle the reference tests is tested */
                    hl = hl || ul || !prune;
                    hr = hr || ur || !prune;
                    if (!prune) tl = tr = -__builtin_huge_valf();
                } else if (__builtin_expect(have && (ul || ur), 0)) { /* rare: the double boxes and true divisions, as the reference does them */
                    const double cx = w->ray[0][rl], cy = w->ray[1][rl], cz = w->ray[2][rl];
                    const double ux = w->ray[3][rl], uy = w->ray[4][rl], uz = w->ray[5][rl];
                    if (ul) { hl = box_hit_exact(sc.halves[(size_t)rec * 2].box, cx, cy, cz, ux, uy, uz); tl = -__builtin_huge_valf(); }
                    if (ur) { hr = box_hit_exact(sc.halves[(size_t)rec * 2 + 1].box, cx, cy, cz, ux, uy, uz); tr = -__builtin_huge_valf(); }
                    if (CNT) cnt.box_exact += (uint64_t)((ul ? 1 : 0) + (ur ? 1 : 0));
                }
                hl = have && hl && !(tl > thr);
                hr = have && hr && !(tr > thr);
                const bool pnl = hl && nL == 0, pll = hl && nL > 0;
                const bool pnr = hr && nR == 0, plr = hr && nR > 0;
                float fl = lower_float(tl), fr = lower_float(tr);
                /* the nearer child goes on top of the stack: it is popped first */
                if (pnl && pnr && fl < fr) {
                    const int32_t ti = L; L = R; R = ti;
                    const float tf = fl; fl = fr; fr = tf;
                }
                const unsigned long long b0 = __ballot(pnl), b1 = __ballot(pnr), b2 = __ballot(pll), b3 = __ballot(plr);
                const int c0 = __popcll(b0), c1 = __popcll(b1);
                if (n_nodes + c0 + c1 > sc.node_cap) { /* cannot happen for a validated scene; never write past the stack */
                    if (lane == 0) atomicOr(sc.overflow, 1);
                } else {
                    if (pnl) w->node[n_nodes + rank_in(b0)] = make_uint2((uint32_t)L | ((uint32_t)rl << RT_NODE_BITS), __float_as_uint(fl));
                    if (pnr) w->node[n_nodes + c0 + rank_in(b1)] = make_uint2((uint32_t)R | ((uint32_t)rl << RT_NODE_BITS), __float_as_uint(fr));
                    n_nodes = uni(n_nodes + c0 + c1);
                }
                const int c2 = __popcll(b2);
                if (pll) {
                    const int q = n_leaves + rank_in(b2);
                    w->leaf[q] = leaf_item((uint32_t)L, (uint32_t)nL, fl, (uint32_t)rl);
                }
                if (plr) {
                    const int q = n_leaves + c2 + rank_in(b3);
                    w->leaf[q] = leaf_item((uint32_t)R, (uint32_t)nR, fr, (uint32_t)rl);
                }
                n_leaves = uni(n_leaves + c2 + __popcll(b3));
                wave_sync();
                if (CNT) { uint64_t t = RT_STAMP(); cnt.cyc_node += t - st0; st0 = t; cnt.node_steps++; cnt.node_lanes += (uint64_t)k; }
            }
        }
    }
}

/* The pool's tables for one ray before a traversal: the ray, the constants of the single-precision box test, and the
 * hit known so far - t0 (nothing at or beyond it can matter) with `pos` as the object / position that holds it, or
 * INT32_MAX for none. */
__device__ __forceinline__ void table_begin(const DScene &sc, WaveLDS *w, int slot, const RayQ &q, bool prune, double t0, int32_t pos, bool never_skip) {
    const RayK k = ray_constants(sc, q.C, q.u);
    w->ray[0][slot] = q.C.x; w->ray[1][slot] = q.C.y; w->ray[2][slot] = q.C.z;
    w->ray[3][slot] = q.u.x; w->ray[4][slot] = q.u.y; w->ray[5][slot] = q.u.z;
    w->rayf[0][slot] = (float)k.iu.x; w->rayf[1][slot] = (float)k.iu.y; w->rayf[2][slot] = (float)k.iu.z;
    w->rayf[3][slot] = (float)k.nc.x; w->rayf[4][slot] = (float)k.nc.y; w->rayf[5][slot] = (float)k.nc.z;
    const float tf = (float)(k.tol * RT_TOLF_SCALE);
    w->tolf[slot] = (q.anyhit && prune) ? -tf : tf; /* the sign bit of a NaN is kept too */
    w->best_t[slot] = (unsigned long long)__double_as_longlong(t0);
    w->best_pos[slot] = pos;
    w->thrf[slot] = (prune && !never_skip) ? upper_float(t0 * RT_PRUNE_SLACK) : __builtin_huge_valf();
}

/* Long sphere lists: the closest sphere of a lane's two rays through the wave's pool (pool_traverse<CNT, true>). The few
 * spheres far larger than the rest (the walls) are tested by every lane first, which gives the traversal a distance to
 * prune with; the rest are reached through the tree over their boxes, 64 (ray, record) or (ray, sphere) pairs per step
 * whichever rays they belong to. The winner is Scene::intersect's: the choice among tested spheres (smallest t, then
 * smallest objectId) does not depend on their order, and a sphere that is skipped - box certainly missed, or wholly
 * behind what is known, or beyond tmax - cannot be it. Rays outside the tree's reach are left to ray_begin (done_*
 * stays false). */
template <bool CNT, typename SP>
__device__ __forceinline__ void spheres_pool(const DScene &sc, const DScene &cold, WaveLDS *w, SP sph, const RayQ &qa, const RayQ &qb, HitRec &ha, HitRec &hb,
                                             bool &done_a, bool &done_b, Cnt &cnt, uint64_t &st0) {
    const int lane = lane_id();
    hit_none(ha);
    hit_none(hb);
    done_a = qa.active && sphere_walk_ok(cold, qa.C);
    done_b = qb.active && sphere_walk_ok(cold, qb.C);
    const unsigned long long mask_a = wave_ballot(done_a), mask_b = wave_ballot(done_b);
    if ((mask_a | mask_b) == 0ull) return;
    const int n_big = uni(cold.n_sph_big);
    if (CNT) {
        for (int s = 0; s < n_big; s++) {
            if (done_a) sphere_test_counted(sph, s, qa.C, qa.u, ha);
            if (done_b) sphere_test_counted(sph, s, qb.C, qb.u, hb);
        }
    } else { /* both rays per sphere in one branch-free piece of code */
        RayQ wa = qa, wb = qb;
        wa.active = done_a;
        wb.active = done_b;
        for (int s = 0; s < n_big; s++) sphere_test2(sph, s, wa, wb, ha, hb);
    }
    const RayQ *const q[2] = {&qa, &qb};
    HitRec *const h[2] = {&ha, &hb};
    const bool done[2] = {done_a, done_b};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (!done[r]) continue;
        /* spheres count below t0 = min(1E10, the hit so far, tmax); the hit so far is entered as the holder of t0 when it is */
        double t0 = 1E10;
        if (h[r]->t < t0) t0 = h[r]->t;
        if (q[r]->tmax < t0) t0 = q[r]->tmax;
        if (!(t0 > 0)) t0 = 0;
        const bool holds = h[r]->obj != INT32_MAX && h[r]->t == t0;
        table_begin(cold, w, lane + r * RT_WAVE, *q[r], true, t0, holds ? h[r]->obj : INT32_MAX, false);
    }
    wave_sync();
    pool_traverse<CNT, true>(sc, cold, w, nullptr, true, true, mask_a, mask_b, cnt, st0);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (!done[r]) continue;
        const int slot = lane + r * RT_WAVE;
        const int32_t oid = w->best_pos[slot];
        if (oid != INT32_MAX && oid != h[r]->obj) { /* a sphere of the tree is closer (or as close, with a lower objectId) */
            h[r]->t = __longlong_as_double((long long)w->best_t[slot]);
            h[r]->obj = oid;
            h[r]->prim = in_global(cold.sph_of_obj)[oid];
            h[r]->mesh = -1;
        }
    }
    wave_sync();
}

/* Scene::intersect, raytracer.cpp:930-951, for up to two rays per lane (slots lane and 64 + lane). Called by all
 * 64 lanes together; active rays get their closest hit. */
/* SPH: compiled with the walk through the sphere tree (scenes with long sphere lists); without it the code for short
 * lists is exactly what it was before the tree existed. */
template <bool CNT, bool SPH, typename SP>
__device__ __forceinline__ void scene_closest_wave(const DScene &sc, const DScene &cold, WaveLDS *w, SP sph, const RootRec *roots, const RayQ &qa0, const RayQ &qb0,
                                                   bool prune, HitRec &ha, HitRec &hb, Cnt &cnt) {
    const int lane = lane_id();
    uint64_t st0 = 0;
    if (CNT) { st0 = RT_STAMP(); cnt.queries++; cnt.query_lanes += (uint64_t)(__popcll(__ballot(qa0.active)) + __popcll(__ballot(qb0.active))); }
    RayQ qa = qa0, qb = qb0;
    /* First on the filter tree (one tree over all meshes; not for the counted variant or RT_F_NO_PRUNE, which
     * reproduce the reference's traversal); the rays whose winner could not be confirmed go round again on the
     * reference's trees. */
    bool own = (!CNT || sc.count_own) && prune && sc.own_rec4 != nullptr;
    HitRec kept_a = {}, kept_b = {};
    bool kept = false, redo_a = false, redo_b = false;
  for (;;) {
    bool sa = false, sb = false;
    if (SPH && cold.sph_rec4 != nullptr && prune) spheres_pool<CNT>(sc, cold, w, sph, qa, qb, ha, hb, sa, sb, cnt, st0);
    if (SPH) {
        ray_begin<CNT, SPH>(cold, w, sph, lane, qa, prune, sa, ha, cnt);
        ray_begin<CNT, SPH>(cold, w, sph, RT_WAVE + lane, qb, prune, sb, hb, cnt);
    } else {
        rays_begin2<CNT>(cold, w, sph, lane, qa, qb, prune, ha, hb, cnt);
    }
    if (CNT) { const uint64_t t = RT_STAMP(); cnt.cyc_roots += t - st0; st0 = t; } /* spheres and ray constants count with the roots */
    if (sc.n_meshes == 0) return;
    {
        /* a shadow ray that a sphere already blocks before tmax is settled: whatever the meshes hold, "hit && t < tmax" stays
         * true (its table entry keeps "no triangle", so rays_end leaves the sphere hit in place) */
        const bool settled_b = prune && qb.anyhit && hb.obj != INT32_MAX && hb.t < qb.tmax;
        const unsigned long long mask_a = wave_ballot(qa.active), mask_b = wave_ballot(qb.active && !settled_b);
        /* inlined on purpose: as a real call (scene fields from the LDS copy, the path state saved around it) the kernel
         * ran 9 % slower at the same occupancy, and compiled for three waves per SIMD 24 % slower */
#if RT_TRAV_PRIO != RT_SHADE_PRIO
        __builtin_amdgcn_s_setprio(RT_TRAV_PRIO);
#endif
        pool_traverse<CNT>(sc, cold, w, roots, own, prune, mask_a, mask_b, cnt, st0);
#if RT_TRAV_PRIO != RT_SHADE_PRIO
        __builtin_amdgcn_s_setprio(RT_SHADE_PRIO);
#endif
    }
    bool da, db;
    rays_end(cold, w, lane, qa, qb, own, ha, hb, da, db, CNT ? &cnt : nullptr);
    wave_sync();
    if (!own || __ballot(da || db) == 0ull) break;
    if (CNT) { cnt.dbg[6] += (uint64_t)(__popcll(__ballot(da)) + __popcll(__ballot(db))); cnt.dbg[7]++; }
    /* second round, on the reference's trees, for the unconfirmed rays only; the others keep their answers */
    own = false;
    kept_a = ha; kept_b = hb;
    kept = true; redo_a = da; redo_b = db;
    qa.active = da;
    qb.active = db;
  }
    if (kept) {
        if (!redo_a) ha = kept_a;
        if (!redo_b) hb = kept_b;
    }
    if (CNT) cnt.cyc_finish += RT_STAMP() - st0;
}

struct Surf {
    V3 P, N, albedo;
    int32_t flags;
};

/* what Scene::intersect hands back for the winning object: P, N, colour, material */
template <typename SP>
__device__ __forceinline__ void hit_surface(const DScene &sc, SP sph, V3 C, V3 u, const HitRec &h, Surf &s) {
    /* the normal before normalisation: get_normalized (a square root and three divisions) is then one piece of code
     * for sphere and mesh hits instead of one in each branch, which a wave with both kinds of lanes would run twice */
    V3 nraw = mk(0, 0, 0);
    if (h.mesh < 0) { /* Sphere::intersect :914-915 */
        V3 O = ld3(sph[h.prim].O);
        s.P = add(C, smul(h.t, u));
        nraw = sub(s.P, O);
        s.albedo = ld3(sph[h.prim].albedo);
        s.flags = sph[h.prim].flags;
    } else if (h.prim < 0) { /* a mesh hit without a triangle: Scene::intersect's zero-initialised locals (:936) */
        s.P = s.albedo = mk(0, 0, 0);
        s.flags = in_global(sc.meshes)[h.mesh].flags;
    } else { /* TriangleMesh::intersect :693-720 */
        const rt_mesh &m = in_global(sc.meshes)[h.mesh];
        const rt_tri_attr *at = in_global(sc.attrs) + h.prim;
        double beta = h.beta, gamma = h.gamma, alpha = 1 - beta - gamma;
        V3 nn = add(add(smul(alpha, ld3(at->n[0])), smul(beta, ld3(at->n[1]))), smul(gamma, ld3(at->n[2])));
        nraw = nn;
        s.P = add(C, u); /* raytracer.cpp:696: C + u, not C + t u */
        s.flags = m.flags;
        if (m.n_tex > 0) {
            const rt_texture tx = in_global(sc.tex)[m.first_tex + h.group];
            int H = tx.H, W = tx.W;
            double U = (alpha * at->uv[0][0] + beta * at->uv[1][0]) + gamma * at->uv[2][0];
            double V = (alpha * at->uv[0][1] + beta * at->uv[1][1]) + gamma * at->uv[2][1];
            U = U * (double)W;
            V = V * (double)H;
            int uvx = (int)(U + 0.5);
            int uvy = (int)(V + 0.5);
            /* uvx % W, uvy % H with negative results wrapped (:706-712). Texture coordinates inside [0, 1) - the usual case -
             * are in range already; the division is only run by a wave in which some lane needs it. */
            if (__builtin_expect(wave_ballot((unsigned)uvx >= (unsigned)W || (unsigned)uvy >= (unsigned)H) != 0ull, 0)) {
                uvx = uvx % W;
                uvy = uvy % H;
                if (uvx < 0) uvx += W;
                if (uvy < 0) uvy += H;
            }
            uvy = H - uvy - 1;
            const uint8_t *px = in_global(sc.texels) + tx.offset + ((int64_t)uvy * W + uvx) * 3;
            const double *lut = in_global(sc.lut);
            s.albedo = mk(lut[px[0]], lut[px[1]], lut[px[2]]);
        } else {
            s.albedo = ld3(m.albedo);
        }
    }
    const V3 nunit = unit(nraw);
    s.N = (h.mesh >= 0 && h.prim < 0) ? mk(0, 0, 0) : nunit;
}

/* random_cos, raytracer.cpp:112-138; sn, cs = sin and cos of 2 pi u1 (rt_math.h: the same bits as the CPU checker computes
 * with the same function) */
__device__ __forceinline__ V3 random_cos_sc(double sn, double cs, double u2, V3 N) {
    double x = cs * sqrt(1 - u2);
    double y = sn * sqrt(1 - u2);
    double z = sqrt(u2);
    V3 T1;
    if (N.x < N.y && N.x < N.z) T1 = mk(0, N.z, -N.y);
    else if (N.y < N.z && N.y < N.x) T1 = mk(N.z, 0, -N.x);
    else T1 = mk(N.y, -N.x, 0);
    T1 = unit(T1);
    V3 T2 = cross(N, T1);
    return add(add(smul(z, N), smul(x, T1)), smul(y, T2));
}
__device__ __forceinline__ V3 random_cos(double u1, double u2, V3 N) {
    double sn, cs;
    RT_SINCOS(2 * RT_PI * u1, &sn, &cs);
    return random_cos_sc(sn, cs, u2, N);
}

/* State of one pixel's sample loop between two queries: the rays to trace next and what the recursion of
 * Scene::getColor would hold on its stack. A diffuse vertex leaves two rays, its shadow ray and the indirect
 * ray (raytracer.cpp:1073, :1087); neither depends on the other's answer, so both go into the same query. */
struct Path {
    V3 rC, ru;      /* path segment to trace (primary, mirror, refraction or indirect ray) */
    V3 shC, shU;    /* shadow ray of the pending diffuse vertex */
    V3 T, Lsum;     /* throughput and radiance of the current path */
    V3 acc;         /* sum over the finished samples of this pixel (raytracer.cpp:1463, :1485) */
    V3 nee;         /* direct light of the pending diffuse vertex, times the throughput up to it */
    double sh_limit;
    uint64_t key;   /* random stream of the current sample */
    uint32_t ctr;   /* draws used so far */
    int32_t k;      /* current sample index */
    int32_t rebond;
    bool lastDiffuse, shadow, segment, fresh, finished; /* shadow / segment: which of the two rays are to be traced */
};

__device__ __forceinline__ void path_begin(Path &q, int k0, bool live) {
    q.rC = mk(0, 0, 0); q.ru = mk(0, 0, 1);
    q.shC = mk(0, 0, 0); q.shU = mk(0, 0, 1);
    q.T = mk(1, 1, 1); q.Lsum = mk(0, 0, 0); q.acc = mk(0, 0, 0);
    q.nee = mk(0, 0, 0);
    q.sh_limit = 0; q.key = 0; q.ctr = 0; q.k = k0; q.rebond = 0;
    q.lastDiffuse = false; q.shadow = false; q.segment = true; q.fresh = true; q.finished = !live;
}

/* Start the next sample of pixel (i, j) if the previous one has ended: one iteration of the k loop
 * up to the getColor call, raytracer.cpp:1464-1483. */
/* PLAIN (here and below): the caller knows that no mode flag is set (the estimator as the reference has it committed), so
 * the code of the other modes is left out at compile time. */
/* The camera ray of sample `key` of pixel (i, j) with its jitter and aperture sample: raytracer.cpp:1467-1483. Used by the
 * render kernels and by the kernel that makes a launch's camera rays ahead of it; the arithmetic is the same
 * instructions in both, so the rays are the same bits. */
__device__ __forceinline__ void camera_ray(const rt_camera &cam, const rt_params &p, int i, int j, uint64_t key, V3 &rC, V3 &ru) {
    const V3 camC = ld3(cam.C), right = ld3(cam.right), up = ld3(cam.up), view = ld3(cam.view);
    double u1 = rng_uniform(key, 0), u2 = rng_uniform(key, 1);
    double u3 = rng_uniform(key, 2), u4 = rng_uniform(key, 3);
    const double4 sc13 = dev_sincos2(2 * RT_PI * u1, 2 * RT_PI * u3);
    const double2 l24 = dev_log2(u2, u4);
    const double s1 = sc13.x, c1 = sc13.y, s3 = sc13.z, c3 = sc13.w, l2 = l24.x, l4 = l24.y;
    double x1 = cam.aa_sigma * c1 * sqrt(-2 * l2),
           x2 = cam.aa_sigma * s1 * sqrt(-2 * l2);
    double x3 = cam.dof_sigma * c3 * sqrt(-2 * l4),
           x4 = cam.dof_sigma * s3 * sqrt(-2 * l4);
    V3 d = mk(j - p.W / 2 + x2 + 0.5, i - p.H / 2 + x1 + 0.5, cam.zdist);
    d = unit(d);
    d = add(add(smul(d.x, right), smul(d.y, up)), smul(d.z, view));
    V3 target = add(camC, smul(cam.focus, d));
    rC = add(camC, mk(x3, x4, 0));
    ru = unit(sub(target, rC));
}

__device__ __forceinline__ auto camera_ray_at(const RenderArgs &ra, int i, int j, int k) {
    const int64_t at = ((int64_t)(k - ra.cam_k0) * (ra.i1 - ra.i0) + (i - ra.i0)) * (ra.j1 - ra.j0) + (j - ra.j0);
    return in_global(ra.cam_rays) + at * RT_CAM_RAY_DOUBLES;
}

/* PREGEN: the launch's camera rays were made ahead of it (ra.cam_rays): a lane reads its ray instead of computing it with
 * the few lanes that start a sample in the same iteration. */
template <bool CNT, bool PLAIN = false, bool PREGEN = false>
__device__ __forceinline__ void path_fresh(const RenderArgs &ra, Path &q, int i, int j, int kend, Cnt &cnt) {
    if (!q.fresh || q.finished) return;
    if (q.k >= kend) { q.finished = true; return; }
    const rt_params &p = ra.p;
    const rt_camera &cam = ra.cam;
    if (CNT) cnt.paths++;
    q.key = rng_key(ra.seed_mixed, (uint32_t)(i * p.W + j), (uint32_t)q.k);
    q.ctr = 0;
    if (PREGEN) {
        const auto *r = camera_ray_at(ra, i, j, q.k);
        const double r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
        q.ctr = 4;
        q.rC = mk(r0, r1, cam.C[2] + 0.0); /* the aperture sample moves the origin in x and y only */
        q.ru = mk(r2, r3, r4);
    } else if (!PLAIN && (p.flags & RT_F_NO_JITTER)) { /* Ray r(C, u), raytracer.cpp:1482 */
        const V3 camC = ld3(cam.C), right = ld3(cam.right), up = ld3(cam.up), view = ld3(cam.view);
        V3 d = mk(j - p.W / 2 + 0.5, i - p.H / 2 + 0.5, cam.zdist);
        d = unit(d);
        const V3 dir = add(add(smul(d.x, right), smul(d.y, up)), smul(d.z, view));
        q.rC = camC; /* both branches assign rC, then ru: stores in a different order get merged into stores through a
                        selected address, and the path state then stays in scratch memory instead of registers */
        q.ru = dir;
    } else {
        V3 rC, ru;
        camera_ray(cam, p, i, j, q.key, rC, ru);
        q.ctr = 4;
        q.rC = rC;
        q.ru = ru;
    }
    q.Lsum = mk(0, 0, 0);
    q.T = mk(1, 1, 1);
    q.rebond = 0;
    q.lastDiffuse = false;
    q.shadow = false;
    q.segment = true;
    q.fresh = false;
}

/* One step of Scene::getColor (raytracer.cpp:953-1093) given the answers of the rays that were traced: first the
 * shadow ray of the pending diffuse vertex (hs), then the path segment (h), in the order the recursion adds their
 * terms. Leaves the next ray(s) in the state or ends the sample (fresh = true). */
/* lean: rays whose answer cannot change the sample are not traced (the reference traces them and drops the answer):
 * - the shadow ray of a vertex whose direct term is exactly zero (surface or light sample facing away: nee == 0, so the
 *   sum gets + 0 whether or not the ray is occluded) - marked by sh_limit = 0, see shadow_wanted();
 * - the ray that leaves a vertex at depth max_rebond: Scene::getColor intersects it first and then returns zero because
 *   of its depth (raytracer.cpp:960-963).
 * The sample's value keeps every bit; only the work counters and the path trace differ, so the callers set lean only with
 * pruning on and no trace asked for. */
template <bool CNT, bool PLAIN = false, typename SP>
__device__ __forceinline__ void path_advance(const DScene &sc, SP sph, const RenderArgs &ra, Path &q, bool inter,
                                             const HitRec &h, bool inter_s, const HitRec &hs, Cnt &cnt, int &branch, bool lean) {
    const rt_params &p = ra.p;
    const uint32_t fl = PLAIN ? 0u : p.flags;
    const double epsilon = p.epsilon;
    bool done = false;
    branch = RT_TRACE_NONE; /* what is done with the segment's hit (the probe's path trace; unused by the render kernels) */
    if (q.shadow) {
        /* raytracer.cpp:1073-1083 (or :1048-1055 for the point light): occluded iff hit before the light */
        if (!(inter_s && hs.t < q.sh_limit)) q.Lsum = add(q.Lsum, q.nee); /* nee: throughput x direct light, see the diffuse branch */
        q.shadow = false;
        if (q.segment) {
            /* color += albedo * getColor(wiRay, rebond + 1, true), raytracer.cpp:1086-1088: (h) is that ray's hit */
            q.rebond++;
            q.lastDiffuse = true;
        }
    }
    if (!q.segment) { /* RT_F_NO_INDIRECT: the vertex had no indirect ray */
        done = true;
    } else if (q.rebond > p.max_rebond || !inter) {
        /* the query comes before the depth test (:960-963) */
        done = true;
    } else if (h.obj == 0 && !(fl & RT_F_POINT_LIGHT)) { /* the light, :966-973 */
        if (q.rebond == 0 || !q.lastDiffuse) {
            double e = sc.I / (4 * RT_PI * RT_PI * sqr(sc.lightR));
            q.Lsum = add(q.Lsum, cmul(q.T, mk(e, e, e)));
        }
        branch = RT_TRACE_LIGHT;
        done = true;
    } else {
        Surf s;
        hit_surface(sc, sph, q.rC, q.ru, h, s);
        const V3 P = s.P, N = s.N, ru = q.ru;
        if (s.flags & RT_MAT_MIRROR) { /* :974-979 */
            if (CNT) cnt.specular++;
            branch = RT_TRACE_MIRROR;
            V3 rd = sub(ru, smul(2 * dot(ru, N), N));
            q.rC = add(P, smul(epsilon, N));
            q.ru = rd;
            q.rebond++;
            q.lastDiffuse = false;
        } else if (s.flags & RT_MAT_TRANSPARENT) { /* :982-1021 */
            if (CNT) cnt.specular++;
            double n1 = 1, n2 = p.ior;
            V3 N2 = N;
            if (dot(ru, N) > 0) {
                double tmp = n1; n1 = n2; n2 = tmp;
                N2 = neg(N);
            }
            double rad = 1 - sqr(n1 / n2) * (1 - sqr(dot(ru, N2)));
            bool reflect;
            if (rad < 0) {
                reflect = true;
            } else if (fl & RT_F_NO_FRESNEL) {
                reflect = false;
            } else {
                double k0 = sqr(n1 - n2) / sqr(n1 + n2);
                double R = k0 + (1 - k0) * sqr(sqr(1 - fabs(dot(N2, ru)))) * (1 - fabs(dot(N2, ru)));
                double random = (double)((int)(rng_draw(q.key, q.ctr) >> 33) % 100); /* rand() % 100, :1006 */
                q.ctr++;
                reflect = (random <= 100 * R);
            }
            branch = rad < 0 ? RT_TRACE_TOTAL_REFLECTION : (reflect ? RT_TRACE_REFLECT : RT_TRACE_REFRACT);
            if (reflect) { /* offset along N, not N2: :997, :1011 */
                V3 rd = sub(ru, smul(2 * dot(ru, N), N));
                q.rC = add(P, smul(epsilon, N));
                q.ru = rd;
            } else {
                V3 tT = smul(n1 / n2, sub(ru, smul(dot(ru, N2), N2)));
                V3 tN = smul(-sqrt(rad), N2);
                q.rC = sub(P, smul(epsilon, N2));
                q.ru = add(tT, tN);
            }
            q.rebond++;
            q.lastDiffuse = false;
        } else { /* diffuse: the shadow ray and the indirect ray, traced together */
            if (CNT) cnt.diffuse++;
            branch = RT_TRACE_DIFFUSE_LIT; /* until the shadow ray's answer says otherwise */
            const V3 Lpos = ld3(sc.L);
            V3 shC = add(P, smul(0.00001, N)), shU;
            /* The draws of the vertex in the reference's order - two for the light sample (:1062), then two for the indirect
             * ray (:1087) - and the sine / cosine pairs of both directions from one call when both are needed. */
            const bool area = !(fl & RT_F_POINT_LIGHT);
            const bool indirect = !(fl & RT_F_NO_INDIRECT) && !(lean && q.rebond >= p.max_rebond);
            double uL1 = 0, uL2 = 0, uI1 = 0, uI2 = 0, sL = 0, cL = 0, sI = 0, cI = 0;
            /* The indirect ray is made whether or not it will be traced (a sample that does not trace it ends at this vertex
             * and draws nothing more): no branch then separates its arithmetic from the light sample's, and the two
             * directions - a square root and three divisions each - are issued interleaved. */
            if (area) { uL1 = rng_uniform(q.key, q.ctr); uL2 = rng_uniform(q.key, q.ctr + 1); q.ctr += 2; }
            uI1 = rng_uniform(q.key, q.ctr); uI2 = rng_uniform(q.key, q.ctr + 1); q.ctr += 2;
            if (area) {
                const double4 r4 = dev_sincos2(2 * RT_PI * uL1, 2 * RT_PI * uI1);
                sL = r4.x; cL = r4.y; sI = r4.z; cI = r4.w;
            } else {
                RT_SINCOS(2 * RT_PI * uI1, &sI, &cI);
            }
            if (!area) { /* :1040-1055 */
                V3 PL = sub(Lpos, P);
                double d = sqrt(norm2(PL));
                shU = divs(PL, d);
                q.nee = muls(smul(sc.I / (4 * RT_PI * d * d), divs(s.albedo, RT_PI)), smax(0., dot(N, divs(PL, d))));
                q.sh_limit = d;
            } else { /* :1059-1083 */
                const V3 lightO = ld3(sc.lightO);
                const double lightR = sc.lightR;
                V3 PL = unit(sub(Lpos, P));
                V3 wv = random_cos_sc(sL, cL, uL2, neg(PL));
                V3 xprime = add(muls(wv, lightR), lightO);
                V3 Px = sub(xprime, P);
                double d = sqrt(norm2(Px));
                Px = divs(Px, d);
                shU = Px;
                double R2 = sqr(lightR);
                double proba = smax(1E-8, dot(neg(PL), wv)) / (RT_PI * R2);
                double J = smax(0., dot(wv, neg(Px))) / (d * d);
                q.nee = divs(muls(muls(smul(sc.I / (4 * RT_PI * RT_PI * R2), divs(s.albedo, RT_PI)), smax(0., dot(N, Px))), J), proba);
                q.sh_limit = d - p.shadow_bias;
            }
            if (lean && q.nee.x == 0 && q.nee.y == 0 && q.nee.z == 0) q.sh_limit = 0;
            /* :1086-1087 */
            q.ru = random_cos_sc(sI, cI, uI2, N);
            q.rC = add(P, smul(epsilon, N));
            /* The recursion adds T * direct light when the shadow ray is back and multiplies T by the albedo before the indirect
             * ray's terms (raytracer.cpp:1082-1088). Both products have all their operands now: forming them here gives the
             * same bits and the albedo need not be kept through the query. */
            q.nee = cmul(q.T, q.nee);
            q.T = cmul(q.T, s.albedo);
            q.segment = indirect;
            q.shC = shC;
            q.shU = shU;
            q.shadow = true;
        }
    }
    if (lean && !q.shadow && q.rebond > p.max_rebond) done = true; /* a mirror or refraction ray beyond the depth limit */
    if (done) {
        q.acc = add(q.acc, q.Lsum);
        q.k++;
        q.fresh = true;
    }
}

/* Whether the pending vertex's shadow ray has to be traced. Its answer is used only as "hit && t < sh_limit"
 * (raytracer.cpp:1073), and no hit has a negative t (Sphere::intersect returns t2 >= 0 or t1 > 0, a triangle needs
 * localt > 0), so a limit that is not positive - too close to the light sample, or set to zero by path_advance because the
 * vertex's direct term is zero - makes the test false whatever is hit. */
__device__ __forceinline__ bool shadow_wanted(const Path &q, bool lean) { return q.shadow && !(lean && !(q.sh_limit > 0)); }

/* Single-kernel form: sum of the radiance of samples k0 .. k0+nk-1 of pixel (i, j), added in that
 * order (the k loop of raytracer.cpp:1464-1486). Called by all 64 lanes of the wave; lanes with
 * live == false only help in the pool. */
template <bool CNT, bool SPH, typename SP>
__device__ __forceinline__ V3 trace_pixel(const DScene &sc, const DScene &cold, WaveLDS *w, SP sph, const RootRec *roots, const RenderArgs &ra, bool live, int i,
                                          int j, int k0, int nk, Cnt &cnt, int32_t *trace = nullptr) {
    Path q;
    path_begin(q, k0, live);
    const int kend = k0 + nk;
    const bool prune = !(ra.p.flags & RT_F_NO_PRUNE);
    const bool lean = prune && trace == nullptr; /* the path trace lists every ray the reference traces */
    uint64_t sp0 = 0;
    while (true) {
        if (CNT) sp0 = RT_STAMP();
        path_fresh<CNT>(ra, q, i, j, kend, cnt);
        if (__ballot(!q.finished) == 0ull) break;
        if (CNT) cnt.cyc_fresh += RT_STAMP() - sp0;
        HitRec h, hs;
        const RayQ qa = RayQ{!q.finished && q.segment, q.rC, q.ru, 1E10, false};
        const RayQ qb = RayQ{!q.finished && shadow_wanted(q, lean), q.shC, q.shU, q.sh_limit, true};
        scene_closest_wave<CNT, SPH>(sc, cold, w, sph, roots, qa, qb, prune, h, hs, cnt);
        if (CNT) sp0 = RT_STAMP();
        if (!q.finished) {
            /* the probe's path trace (rt_sample_trace; nk == 1): entry d belongs to the query at recursion depth d */
            const bool inter = h.obj != INT32_MAX, inter_s = hs.obj != INT32_MAX;
            const int d_vertex = q.rebond, d_seg = q.rebond + ((q.shadow && q.segment) ? 1 : 0);
            const bool had_shadow = q.shadow, had_seg = q.segment;
            const uint32_t draws = q.ctr;
            const bool occluded = inter_s && hs.t < q.sh_limit;
            int branch;
            path_advance<CNT>(cold, sph, ra, q, inter, h, inter_s, hs, cnt, branch, lean);
            if (trace) {
                if (had_shadow && occluded && d_vertex < RT_TRACE_LEN) trace[4 * d_vertex + 3] = RT_TRACE_DIFFUSE_SHADOWED;
                if (had_seg && d_seg < RT_TRACE_LEN) {
                    int32_t *e = trace + 4 * d_seg;
                    e[0] = inter ? h.obj : -1;
                    e[1] = (inter && h.mesh >= 0 && h.prim >= 0) ? sc.tris[h.prim].orig_id : -1;
                    e[2] = (int32_t)draws;
                    e[3] = branch;
                }
            }
        }
        if (CNT) cnt.cyc_shade += RT_STAMP() - sp0;
    }
    return q.acc;
}

/* Slots run over 8x8 pixel tiles of the render rectangle (64 consecutive slots = one tile), so the camera rays a wave
 * starts together stay together. */
/* The render loop of a persistent wave. Lanes take pixels ("slots") from a global counter, one 64-slot tile
 * per request; a lane keeps its pixel for all nk samples (added in order, raytracer.cpp:1464-1486), writes the
 * sum and takes the next free slot. Because lanes re-fill independently, the wave's pool stays full until the
 * whole frame runs out of pixels, instead of idling while the longest of 64 fixed pixels finishes. */
template <bool CNT, bool SPH, bool PLAIN, bool PREGEN = false, typename SP>
__device__ __forceinline__ void render_persistent(const DScene &sc, const DScene &cold, WaveLDS *w, SP sph, const RootRec *roots,
                                                  const RenderArgs &ra, int k0, int nk, int chunk, double *accum, double *unit_sums, int *next_slot, int n_slots,
                                                  Cnt &cnt) {
    Path q;
    path_begin(q, k0, false); /* finished: no pixel yet */
    const int kend = k0 + nk;
    /* A unit of work is `chunk` consecutive samples of one pixel; the units are numbered chunk by chunk over the frame
     * (n_slots pixel slots each). Short units keep the end of a launch short: once the counter runs out, a wave only has
     * to finish the units its lanes hold. With more than one unit per pixel a unit's sum is stored in its own place,
     * unit_sums[unit of the pixel][pixel], and a second kernel adds them to the accumulator in the order of the units,
     * so the result does not depend on which unit ends first. */
    const int n_chunks = (nk + chunk - 1) / chunk;
    const int n_total = n_slots * n_chunks;
    int kend_l = kend, unit_c = 0;
    bool have = false, exhausted = false;
#ifdef RT_PROBE_LANES /* timing probe: only every RT_PROBE_LANES-th lane of a wave takes pixels (how a query's cost depends on its ray count) */
    exhausted = (lane_id() % RT_PROBE_LANES) != 0;
#endif
    int i = 0, j = 0;
    int cur = 0, end = 0; /* wave-uniform: the part of the last tile request not handed out yet */
    int tile_c = 0, tile_i = 0, tile_j = 0; /* wave-uniform: that request's chunk and the pixel of its tile's corner */
    const bool prune = PLAIN ? true : !(ra.p.flags & RT_F_NO_PRUNE); /* also: rays without effect are not traced (path_advance) */
    uint64_t sp0 = 0;
    while (true) {
        if (CNT) sp0 = RT_STAMP();
        if (have && q.finished) { /* all samples of this lane's pixel are in q.acc */
            double *px = accum + ((int64_t)i * ra.p.W + j) * 3;
            if (n_chunks > 1) {
                /* [unit][row of the rectangle][column of the rectangle][3] */
                double *us = unit_sums + (((int64_t)unit_c * (ra.i1 - ra.i0) + (i - ra.i0)) * (ra.j1 - ra.j0) + (j - ra.j0)) * 3;
                us[0] = q.acc.x;
                us[1] = q.acc.y;
                us[2] = q.acc.z;
            } else {
                px[0] = px[0] + q.acc.x;
                px[1] = px[1] + q.acc.y;
                px[2] = px[2] + q.acc.z;
            }
            have = false;
        }
        const unsigned long long need = wave_ballot(!have && !exhausted);
        if (need) {
            if (cur == end) { /* one atomic per tile */
                int base = 0;
                if (lane_id() == 0) base = atomicAdd(next_slot, RT_WAVE);
                cur = uni(base);
                end = cur + RT_WAVE < n_total ? cur + RT_WAVE : n_total;
                if (cur >= n_total) { cur = end = n_total; }
                /* The 64 slots of a request are one tile of one chunk (requests and the slots per chunk are multiples of
                 * 64): the chunk and the tile's corner are found once per request, with wave-uniform divisions, and a lane
                 * only adds its place inside the tile. */
                tile_c = uni(cur / n_slots);
                const int tile = (cur - tile_c * n_slots) >> 6, tiles_x = (ra.j1 - ra.j0 + 7) >> 3;
                const int ty = uni(tile / tiles_x);
                tile_i = ra.i0 + ty * 8;
                tile_j = ra.j0 + (tile - ty * tiles_x) * 8;
            }
            if (!have && !exhausted) {
                const int mine = cur + rank_in(need);
                if (cur >= n_total) {
                    exhausted = true;
                } else if (mine < end) {
                    const int ln = mine & 63;
                    j = tile_j + (ln & 7);
                    i = tile_i + (ln >> 3);
                    const bool live = (j < ra.j1) && (i < ra.i1);
                    const int ks = k0 + tile_c * chunk;
                    unit_c = tile_c;
                    kend_l = ks + chunk < kend ? ks + chunk : kend;
                    path_begin(q, ks, live);
                    have = live; /* a slot outside the rectangle is skipped */
                }
            }
            const int taken = __popcll(need);
            cur = (cur + taken < end) ? cur + taken : end;
        }
        path_fresh<CNT, PLAIN, PREGEN>(ra, q, i, j, kend_l, cnt);
        const bool tracing = !q.finished;
        if (wave_ballot(tracing) == 0ull) {
            if (wave_ballot(!exhausted || have) == 0ull) break;
            continue; /* nothing to trace in this iteration: only bookkeeping (pixels ended or were just assigned) */
        }
        if (CNT) cnt.cyc_fresh += RT_STAMP() - sp0;
        HitRec h, hs;
        const RayQ qa = RayQ{tracing && q.segment, q.rC, q.ru, 1E10, false};
        const RayQ qb = RayQ{tracing && shadow_wanted(q, prune), q.shC, q.shU, q.sh_limit, true};
        scene_closest_wave<CNT, SPH>(sc, cold, w, sph, roots, qa, qb, prune, h, hs, cnt);
        if (CNT) sp0 = RT_STAMP();
        int branch; /* unused here */
        if (tracing) path_advance<CNT, PLAIN>(cold, sph, ra, q, h.obj != INT32_MAX, h, hs.obj != INT32_MAX, hs, cnt, branch, prune);
        if (CNT) cnt.cyc_shade += RT_STAMP() - sp0;
    }
}

/* wave-reduce a counter set and add it to global memory with one atomic per field per wave */
__device__ __forceinline__ void flush_counters(const Cnt &c, rt_counters *out) {
    uint64_t v[8] = {c.paths, c.rays, c.box_tests, c.nodes, c.tris, c.mesh_calls, c.diffuse, c.specular};
    unsigned long long *o = reinterpret_cast<unsigned long long *>(out);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        uint64_t x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor((unsigned long long)x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(o + q, (unsigned long long)x);
    }
    /* wave-level diagnostics: identical in every lane, lane 0 adds them behind the public counters */
    uint64_t d[12] = {c.cyc_fresh, c.cyc_roots, c.cyc_node, c.cyc_leaf, c.cyc_finish, c.cyc_shade,
                      c.node_steps, c.node_lanes, c.leaf_steps, c.leaf_lanes, c.queries, c.query_lanes};
    if ((threadIdx.x & 63) == 0) {
        for (int q = 0; q < 12; q++) atomicAdd(o + 8 + q, (unsigned long long)d[q]);
        for (int q = 0; q < 8; q++) atomicAdd(o + 20 + q, (unsigned long long)c.dbg[q]);
    }
    uint64_t e[3] = {c.sph_tests, c.sph_nodes, c.box_exact};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        uint64_t x = e[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor((unsigned long long)x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(o + 28 + q, (unsigned long long)x);
    }
}

} // namespace rtd
