/*
 * rt_fast.hip.h - the path-tracing loop in single precision (rt_params.flags & RT_F_FAST) for gfx950.
 *
 * The same algorithm as rt_device.hip.h - the estimator of Scene::getColor as committed (raytracer.cpp:953-1093), the same
 * counter-based random stream and draw order, the same wave-cooperative pool traversal of the filter tree, the same work
 * units - with the arithmetic in float and FMA contraction allowed (this header is compiled by rt_fast.hip with
 * -ffp-contract=fast). It gives up what the double-precision kernel exists for: hit identities and radiance are no longer
 * bit-equal to the reference's, only statistically (the converged image stays inside the stated bias / RMSE bars of
 * tests/test_gpu_fast.py). What it leaves out: the second pass on the reference's own trees and the confirmation of the
 * winner (nothing is promised bit for bit), the reproduction of a mesh hit at t >= 1E9, long sphere lists, the noise-off
 * estimator modes, the counters. What it keeps in double: the sphere tests - a ray inside wall spheres of radius ~1000 loses
 * five digits in the discriminant, more than a float has to spare - and the accumulator.
 *
 * Offsets: the reference leaves a surface by epsilon = 1e-5 along the normal and accepts a shadow hit below d - 1e-4
 * (raytracer.cpp:955, :1073), sized for doubles. A float position at |P| ~ 100 is known to a few 1e-6 and rays that re-hit
 * the surface they leave darken the image (measured: 9.4 % of the light lost at 2^-20 |P|, 1.2 % at 2^-17), so epsilon here
 * is the larger of the parameter and 2^-15 of the hit point's largest coordinate (fast_eps), and the shadow bias keeps the
 * reference's ratio to it (see the diffuse branch of fast_path_advance).
 */
#pragma once
#include "rt_device.hip.h"

namespace rtf {
using namespace rtd;

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 mkf(float x, float y, float z) { return F3{x, y, z}; }
template <typename P>
__device__ __forceinline__ F3 ldf(P p) { return F3{(float)p[0], (float)p[1], (float)p[2]}; } /* three floats or doubles, any address space */
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return mkf(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return mkf(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return mkf(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(float s, F3 b) { return mkf(s * b.x, s * b.y, s * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return mkf(a.x * b.x, a.y * b.y, a.z * b.z); } /* component-wise, raytracer.cpp:87-90 */
__device__ __forceinline__ float dotf(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 crossf(F3 a, F3 b) { return mkf(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* hardware estimates (1 ulp) instead of correctly rounded division and square root: nothing here is compared bit for bit */
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ F3 unitf(F3 a) { /* get_normalized: no zero guard, a zero vector gives NaNs (0 x inf) as the reference's 0 / 0 does (raytracer.cpp:44-48) */
    const float r = __builtin_amdgcn_rsqf(dotf(a, a));
    return mkf(a.x * r, a.y * r, a.z * r);
}
#define RTF_PI 3.14159265358979323846f

/* a uniform in (0, 1): 24 bits of the stream's draw, centred in its cell so that log() never sees zero */
__device__ __forceinline__ float uniform_f(uint64_t key, uint32_t n) { return ((float)(uint32_t)(rng_draw(key, n) >> 40) + 0.5f) * 0x1p-24f; }

/* triangle of the filter tree's order in single precision: A, e1 = B - A, e2 = C - A; pos = position in the reference's
 * visiting order (breaks exact ties in t like the double-precision kernel does, and finds the attributes) */
struct FastTri {
    float A[3], e1[3], e2[3];
    int32_t pos;
};

#ifndef RTF_LOOKAHEAD
#define RTF_LOOKAHEAD 0 /* 1: full-width node steps request the next batch's records before the waiting batch is tested (two batches in flight) */
#endif

struct FastLDS {
    float ray[6][RT_RAY_SLOTS];   /* C, u of each ray */
    float rayf[6][RT_RAY_SLOTS];  /* 1/u, -(C/u): the box test's constants */
    float tolf[RT_RAY_SLOTS];     /* |tolf|: below this the box test does not say "missed" (NaN: never); sign bit set for a shadow ray */
    float thrf[RT_RAY_SLOTS];     /* not below the distance beyond which nothing can change the ray's answer */
    unsigned long long best[RT_RAY_SLOTS]; /* bits of the smallest accepted t (> 0: bit order = value order) << 32 | lowest position with it */
    int32_t mark[RT_WAVE];        /* scratch words (the branch-free push option of children_step) */
#ifdef RTF_LDS_PAD /* diagnostic: fewer resident waves per CU */
    float lds_pad[RTF_LDS_PAD / 4];
#endif
    DScene sc;
    RenderArgs ra;
    uint2 leaf[RT_LEAF_CAP];
    uint2 node[1];                /* node_cap entries, then the staged spheres */
};
__device__ __forceinline__ const rt_sphere *fast_lds_spheres(const FastLDS *w, int node_cap) {
    size_t off = (offsetof(FastLDS, node) + (size_t)node_cap * 8 + 15) & ~(size_t)15;
    return reinterpret_cast<const rt_sphere *>(reinterpret_cast<const unsigned char *>(w) + off);
}

struct FHit {
    float t;      /* +inf while nothing is hit */
    int32_t obj;  /* objectId, INT32_MAX while nothing is hit */
    int32_t prim; /* sphere index, or triangle position in the reference's order */
    int32_t mesh; /* mesh index or -1 for a sphere */
};

/* how far to step off a surface at P / how much earlier than the light sample a shadow hit must be: the parameters of the
 * reference, or a few units in the last place of the position if that is more */
__device__ __forceinline__ float fast_eps(float param, F3 P) {
    const float m = fmaxf(fmaxf(__builtin_fabsf(P.x), __builtin_fabsf(P.y)), __builtin_fabsf(P.z));
#ifndef RTF_EPS_REL
#define RTF_EPS_REL 0x1p-15f /* 256 ulp of the largest coordinate */
#endif
    return fmaxf(param, RTF_EPS_REL * m);
}

/* Sphere::intersect (raytracer.cpp:883-918) and Scene::intersect's choice (:938) for a lane's two rays, in double on the
 * float rays: both chains side by side, no branches (sphere_test2 of rt_device.hip.h) */
template <typename SP>
__device__ __forceinline__ void fast_spheres(SP sph, int n_sph, bool act_a, F3 Ca, F3 ua, bool act_b, F3 Cb, F3 ub, FHit &ha, FHit &hb) {
    double best[2] = {__builtin_huge_val(), __builtin_huge_val()};
    int32_t obj[2] = {INT32_MAX, INT32_MAX}, prim[2] = {-1, -1};
    const double C[2][3] = {{Ca.x, Ca.y, Ca.z}, {Cb.x, Cb.y, Cb.z}}, u[2][3] = {{ua.x, ua.y, ua.z}, {ub.x, ub.y, ub.z}};
    const bool act[2] = {act_a, act_b};
    for (int s = 0; s < n_sph; s++) {
        const double Ox = sph[s].O[0], Oy = sph[s].O[1], Oz = sph[s].O[2], R = sph[s].R;
        const int32_t oid = sph[s].object_id;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const double cx = C[r][0] - Ox, cy = C[r][1] - Oy, cz = C[r][2] - Oz;
            const double b = 2 * (u[r][0] * cx + u[r][1] * cy + u[r][2] * cz);
            const double c = (cx * cx + cy * cy + cz * cz) - R * R;
            const double delta = b * b - 4 * c;
            const double sq = sqrt(delta);
            const double t2 = (-b + sq) * 0.5, t1 = (-b - sq) * 0.5;
            const double t = (t1 > 0) ? t1 : t2;
            const bool upd = act[r] & !(delta < 0) & !(t2 < 0) & ((t < best[r]) | ((t == best[r]) & (oid < obj[r])));
            best[r] = upd ? t : best[r];
            obj[r] = upd ? oid : obj[r];
            prim[r] = upd ? s : prim[r];
        }
    }
    ha = FHit{(float)best[0], obj[0], prim[0], -1};
    hb = FHit{(float)best[1], obj[1], prim[1], -1};
}

/* the pool's table entries of one ray once its closest sphere is known (ray_tables of rt_device.hip.h) */
__device__ __forceinline__ void fast_tables(const DScene &sc, FastLDS *w, int slot, bool active, F3 C, F3 u, float tmax, bool anyhit, float t_sphere) {
    if (!active) return;
    const float iux = __builtin_amdgcn_rcpf(u.x), iuy = __builtin_amdgcn_rcpf(u.y), iuz = __builtin_amdgcn_rcpf(u.z);
    w->ray[0][slot] = C.x; w->ray[1][slot] = C.y; w->ray[2][slot] = C.z;
    w->ray[3][slot] = u.x; w->ray[4][slot] = u.y; w->ray[5][slot] = u.z;
    w->rayf[0][slot] = iux; w->rayf[1][slot] = iuy; w->rayf[2][slot] = iuz;
    w->rayf[3][slot] = -(C.x * iux); w->rayf[4][slot] = -(C.y * iuy); w->rayf[5][slot] = -(C.z * iuz);
    /* the tolerance of the single-precision slab test, as ray_constants derives it: 7 x 2^-24 M with M the largest slab
     * parameter any box of the scene can give; rays it cannot be trusted for enter every box */
    const float ax = __builtin_fabsf(iux), ay = __builtin_fabsf(iuy), az = __builtin_fabsf(iuz);
    const float mx = ((float)sc.bmax[0] + __builtin_fabsf(C.x)) * ax, my = ((float)sc.bmax[1] + __builtin_fabsf(C.y)) * ay, mz = ((float)sc.bmax[2] + __builtin_fabsf(C.z)) * az;
    const bool ok = (mx < 1e30f) && (my < 1e30f) && (mz < 1e30f) && (ax > 1e-25f) && (ay > 1e-25f) && (az > 1e-25f) && (ax < 1e25f) && (ay < 1e25f) && (az < 1e25f);
    const float tf = ok ? 4.2e-7f * fmaxf(mx, fmaxf(my, mz)) : __builtin_nanf("");
    w->tolf[slot] = anyhit ? -tf : tf;
    float t0 = fminf(t_sphere, tmax); /* mesh hits count below the closest sphere and below what the caller looks at */
    if (!(t0 > 0)) t0 = 0;
    w->best[slot] = ((unsigned long long)__float_as_uint(t0) << 32) | 0xFFFFFFFFull;
    w->thrf[slot] = upper_float((double)t0 * RT_PRUNE_SLACK); /* +inf stays +inf */
}

/* One query of the wave on the filter tree: everything between "the rays' tables are in LDS" and "the rays' best
 * triangles are in LDS" (pool_traverse of rt_device.hip.h, filter-tree branch only). */
__device__ __forceinline__ void fast_pool(const DScene &sc, FastLDS *w, const FastTri *ftris, unsigned long long mask_a, unsigned long long mask_b) {
    const int lane = lane_id();
    const bool act_b = (mask_b >> lane) & 1ull;
    const NodeRec4 *const tree4 = sc.own_rec4;
    const uint32_t n_recs = sc.n_own_nodes, n_tri_lim = sc.n_own_tris;
    const int node_soft = sc.node_soft;
    const int quad_shift = sc.own_leaf_shift;
    const int leaf_trigger = RT_WAVE >> quad_shift;
    int n_nodes = 0, n_leaves = 0;
    unsigned steps = 0;
    Cnt cnt_unused = {};
    uint64_t st_unused = 0;
    {
        /* every active ray starts at record 0, whose children are tested here right away (wave-uniform addresses) */
        const float4 *rp = reinterpret_cast<const float4 *>(tree4);
        const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3], q4 = rp[4], q5 = rp[5];
        const uint4 ch = *reinterpret_cast<const uint4 *>(rp + 6);
        const float ninf = -__builtin_huge_valf();
        wide_children(sc, w, lane, mask_a, lane, ninf, q0, q1, q2, q3, q4, q5, ch, n_nodes, n_leaves);
        const int nb = __popcll(mask_b);
        if (n_nodes + 4 * nb <= node_soft && n_leaves + 4 * nb <= RT_LEAF_CAP) {
            wide_children(sc, w, lane, mask_b, RT_WAVE + lane, ninf, q0, q1, q2, q3, q4, q5, ch, n_nodes, n_leaves);
        } else {
            if (act_b) w->node[n_nodes + rank_in(mask_b)] = make_uint2((uint32_t)(RT_WAVE + lane) << RT_NODE_BITS, __float_as_uint(ninf));
            n_nodes += nb;
        }
        wave_sync();
    }
#if RTF_LOOKAHEAD
    /* a full-width batch of records whose loads are in flight while the batch before it is tested and pushed */
    bool pend = false;
    unsigned long long p_have = 0ull;
    int p_rl = 0, p_k = 0;
    float p_item_t = 0.f;
    float p_bx[24];
    uint32_t p_code[4];
#pragma unroll
    for (int a = 0; a < 24; a++) p_bx[a] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; a++) p_code[a] = RT_CHILD_EMPTY;
#else
    const bool pend = false;
    const int p_k = 0;
#endif
    while (n_nodes + n_leaves != 0 || pend) {
        if (++steps > RT_STEP_LIMIT) {
            if (lane == 0) atomicOr(sc.overflow, 8);
            break;
        }
        const int reserved = pend ? 4 * p_k : 0; /* pushes the waiting batch may still make, on either stack */
        const bool do_leaf = n_leaves > 0 && ((n_leaves >= leaf_trigger) || (n_nodes == 0 && !pend) || (n_leaves + reserved + 4 * RT_WAVE > RT_LEAF_CAP));
        if (do_leaf) {
            /* lane r of an item's group takes the item's r-th triangle (leaves hold at most 1 << quad_shift) */
            const int mq = n_leaves < (RT_WAVE >> quad_shift) ? n_leaves : (RT_WAVE >> quad_shift);
            const int iq = lane >> quad_shift, rq = lane & ((1 << quad_shift) - 1);
            uint2 itq = make_uint2(0u, 0u);
            int cq = 0;
            if (iq < mq) {
                itq = w->leaf[n_leaves - 1 - iq];
                cq = leaf_count(itq);
                if (leaf_entry(itq) > w->thrf[leaf_slot(itq)]) cq = 0; /* fallen behind its ray's threshold since it was pushed */
            }
            n_leaves -= mq;
            const int rl = leaf_slot(itq);
            const uint32_t tri = leaf_first(itq) + (uint32_t)rq;
            bool ok = rq < cq;
            if (__builtin_expect(wave_ballot(ok && tri >= n_tri_lim) != 0ull, 0)) { /* never fetch outside the arrays */
                if (ok && tri >= n_tri_lim) { atomicOr(sc.overflow, 2); ok = false; }
            }
            const FastTri *tp = in_global(ftris) + (ok ? tri : 0u);
            const F3 A = ldf(tp->A), e1 = ldf(tp->e1), e2 = ldf(tp->e2);
            const uint32_t pos = (uint32_t)tp->pos;
            const F3 C = mkf(w->ray[0][rl], w->ray[1][rl], w->ray[2][rl]), u = mkf(w->ray[3][rl], w->ray[4][rl], w->ray[5][rl]);
            /* raytracer.cpp:673-687 */
            const F3 N = crossf(e1, e2), AO = C - A, AOu = crossf(AO, u);
            const float inv = frcp(dotf(u, N));
            const float beta = -dotf(e2, AOu) * inv, gamma = dotf(e1, AOu) * inv, alpha = 1.f - beta - gamma, t = -dotf(AO, N) * inv;
            const bool acc = ok && beta >= 0 && gamma >= 0 && beta <= 1 && gamma <= 1 && alpha >= 0 && t > 0 && t < __builtin_huge_valf();
            if (wave_ballot(acc) != 0ull) {
                if (acc) {
                    /* (min t, then min position) per ray in one 64-bit minimum */
                    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | pos;
                    const unsigned long long old = atomicMin(&w->best[rl], key);
                    if (key < old) w->thrf[rl] = (__float_as_int(w->tolf[rl]) < 0) ? 0.f : upper_float((double)t * RT_PRUNE_SLACK); /* a later, larger bound from another lane of this step is still a bound */
                }
            }
            wave_sync();
        } else {
            int k = n_nodes < RT_WAVE ? n_nodes : RT_WAVE;
            const int room = (node_soft - n_nodes - reserved) / 3;
            if (k > room) k = room > 1 ? room : 1;
#if RTF_LOOKAHEAD
            if (k > RT_WAVE / 2 && room >= k && n_leaves + reserved + 4 * k <= RT_LEAF_CAP) {
                /* pop the next full-width batch and request its records; then test and push the waiting batch while they arrive */
                const unsigned long long in_k = wave_ballot(lane < k);
                const uint2 it = w->node[__builtin_amdgcn_inverse_ballot_w64(in_k) ? n_nodes - 1 - lane : 0];
                n_nodes -= k;
                const int rl = (int)(it.x >> RT_NODE_BITS);
                uint32_t rec = it.x & RT_LOW_MASK;
                const unsigned long long outside = in_k & wave_ballot(rec >= n_recs);
                if (__builtin_expect(outside != 0ull, 0)) {
                    if (lane == 0) atomicOr(sc.overflow, 2);
                }
                const unsigned long long have_mask = in_k & ~outside;
                if (!__builtin_amdgcn_inverse_ballot_w64(have_mask)) rec = 0;
                float bx[24];
                uint32_t code[4];
                load_children<4>(tree4 + rec, 0, bx, code);
                if (pend) children_step<4>(sc, w, lane, p_have, p_rl, p_item_t, p_bx, p_code, n_nodes, n_leaves);
                pend = true;
                p_have = have_mask; p_rl = rl; p_k = k; p_item_t = __uint_as_float(it.y);
#pragma unroll
                for (int a = 0; a < 24; a++) p_bx[a] = bx[a];
#pragma unroll
                for (int a = 0; a < 4; a++) p_code[a] = code[a];
                wave_sync();
                continue;
            }
            if (pend) { /* no full batch to start: the waiting one is finished first, its children may make one */
                children_step<4>(sc, w, lane, p_have, p_rl, p_item_t, p_bx, p_code, n_nodes, n_leaves);
                pend = false;
                wave_sync();
                continue;
            }
            if (n_nodes == 0) continue;
#endif
            if (RT_NARROW_STEP > 1 && k <= RT_WAVE / 4) own_node_step<false, false, 4>(sc, w, tree4, n_recs, lane, k, n_nodes, n_leaves, cnt_unused, st_unused);
            else if (RT_NARROW_STEP > 0 && k <= RT_WAVE / 2) own_node_step<false, false, 2>(sc, w, tree4, n_recs, lane, k, n_nodes, n_leaves, cnt_unused, st_unused);
            else own_node_step<false, false, 1>(sc, w, tree4, n_recs, lane, k, n_nodes, n_leaves, cnt_unused, st_unused);
        }
    }
}

struct FSurf {
    F3 P, N, albedo;
    int32_t flags;
};

/* Scene::intersect for a lane's two rays (slots lane and 64 + lane). qa: a path segment (closest hit, with the winner's
 * barycentric coordinates); qb: a shadow ray (only "hit before tmax" matters). */
template <typename SP>
__device__ __forceinline__ void fast_closest(const DScene &sc, const DScene &cold, FastLDS *w, SP sph, const FastTri *ftris, bool act_a, F3 Ca, F3 ua, bool act_b, F3 Cb, F3 ub,
                                             float tmax_b, FHit &ha, FHit &hb, float &beta, float &gamma) {
    const int lane = lane_id();
    fast_spheres(sph, uni(cold.n_spheres), act_a, Ca, ua, act_b, Cb, ub, ha, hb);
    beta = gamma = 0.f;
    if (sc.n_meshes == 0 || sc.own_rec4 == nullptr) return; /* no mesh, or none with a triangle that can be reached */
    /* a shadow ray that a sphere blocks before the light is settled whatever the meshes hold */
    const bool settled_b = hb.obj != INT32_MAX && hb.t < tmax_b;
    fast_tables(cold, w, lane, act_a, Ca, ua, __builtin_huge_valf(), false, ha.t);
    fast_tables(cold, w, RT_WAVE + lane, act_b && !settled_b, Cb, ub, tmax_b, true, hb.t);
    wave_sync();
    const unsigned long long mask_a = wave_ballot(act_a), mask_b = wave_ballot(act_b && !settled_b);
    if ((mask_a | mask_b) != 0ull) {
        __builtin_amdgcn_s_setprio(RT_TRAV_PRIO);
        fast_pool(sc, w, ftris, mask_a, mask_b);
        __builtin_amdgcn_s_setprio(RT_SHADE_PRIO);
    }
    /* the mesh hits, and Scene::intersect's choice between each and the sphere hit (strict '<', lowest objectId on a tie) */
    const unsigned long long ka = act_a ? w->best[lane] : ~0ull, kb = (act_b && !settled_b) ? w->best[RT_WAVE + lane] : ~0ull;
    const uint32_t pa = (uint32_t)ka, pb = (uint32_t)kb;
    const bool tri_a = pa != 0xFFFFFFFFu, tri_b = pb != 0xFFFFFFFFu;
    const rt_tri *ta = in_global(cold.tris) + (tri_a ? pa : 0u), *tb = in_global(cold.tris) + (tri_b ? pb : 0u);
    const F3 A = ldf(ta->A), e1 = ldf(ta->e1), e2 = ldf(ta->e2);
    const int32_t mesh_a = ta->mesh, mesh_b = tb->mesh;
    const int32_t oid_a = in_global(cold.meshes)[mesh_a].object_id, oid_b = in_global(cold.meshes)[mesh_b].object_id;
    if (tri_a) {
        const float t = __uint_as_float((uint32_t)(ka >> 32));
        if (t < ha.t || (t == ha.t && oid_a < ha.obj)) {
            ha = FHit{t, oid_a, (int32_t)pa, mesh_a};
            const F3 AO = Ca - A, AOu = crossf(AO, ua), N = crossf(e1, e2);
            const float inv = frcp(dotf(ua, N));
            beta = -dotf(e2, AOu) * inv;
            gamma = dotf(e1, AOu) * inv;
        }
    }
    if (tri_b) {
        const float t = __uint_as_float((uint32_t)(kb >> 32));
        if (t < hb.t || (t == hb.t && oid_b < hb.obj)) hb = FHit{t, oid_b, (int32_t)pb, mesh_b};
    }
    wave_sync();
}

/* what Scene::intersect hands back for the winning object (hit_surface of rt_device.hip.h) */
template <typename SP>
__device__ __forceinline__ void fast_surface(const DScene &sc, SP sph, F3 C, F3 u, const FHit &h, float beta, float gamma, FSurf &s) {
    F3 nraw = mkf(0, 0, 0);
    if (h.mesh < 0) { /* Sphere::intersect :914-915 */
        s.P = C + h.t * u;
        nraw = s.P - ldf(sph[h.prim].O);
        s.albedo = ldf(sph[h.prim].albedo);
        s.flags = sph[h.prim].flags;
    } else { /* TriangleMesh::intersect :693-720 */
        const rt_mesh &m = in_global(sc.meshes)[h.mesh];
        const rt_tri_attr *at = in_global(sc.attrs) + h.prim;
        const float alpha = 1.f - beta - gamma;
        nraw = alpha * ldf(at->n[0]) + beta * ldf(at->n[1]) + gamma * ldf(at->n[2]);
        s.P = C + u; /* raytracer.cpp:696: C + u, not C + t u */
        s.flags = m.flags;
        if (m.n_tex > 0) {
            const rt_texture tx = in_global(sc.tex)[m.first_tex + at->group];
            const int H = tx.H, W = tx.W;
            const float U = ((alpha * (float)at->uv[0][0] + beta * (float)at->uv[1][0]) + gamma * (float)at->uv[2][0]) * (float)W;
            const float V = ((alpha * (float)at->uv[0][1] + beta * (float)at->uv[1][1]) + gamma * (float)at->uv[2][1]) * (float)H;
            int uvx = (int)(U + 0.5f), uvy = (int)(V + 0.5f);
            if (__builtin_expect(wave_ballot((unsigned)uvx >= (unsigned)W || (unsigned)uvy >= (unsigned)H) != 0ull, 0)) {
                uvx = uvx % W;
                uvy = uvy % H;
                if (uvx < 0) uvx += W;
                if (uvy < 0) uvy += H;
            }
            uvy = H - uvy - 1;
            const uint8_t *px = in_global(sc.texels) + tx.offset + ((int64_t)uvy * W + uvx) * 3;
            const double *lut = in_global(sc.lut);
            s.albedo = mkf((float)lut[px[0]], (float)lut[px[1]], (float)lut[px[2]]);
        } else {
            s.albedo = ldf(m.albedo);
        }
    }
    s.N = unitf(nraw);
}

/* random_cos, raytracer.cpp:112-138 (tangent by the smallest signed component; a zero tangent gives NaNs as there) */
__device__ __forceinline__ F3 fast_random_cos(float u1, float u2, F3 N) {
    float sn, cs;
    __sincosf(2.f * RTF_PI * u1, &sn, &cs);
    const float r = fsqrt(1.f - u2), x = cs * r, y = sn * r, z = fsqrt(u2);
    F3 T1;
    if (N.x < N.y && N.x < N.z) T1 = mkf(0, N.z, -N.y);
    else if (N.y < N.z && N.y < N.x) T1 = mkf(N.z, 0, -N.x);
    else T1 = mkf(N.y, -N.x, 0);
    T1 = unitf(T1);
    const F3 T2 = crossf(N, T1);
    return z * N + x * T1 + y * T2;
}

struct FPath {
    F3 rC, ru;   /* path segment to trace */
    F3 shC, shU; /* shadow ray of the pending diffuse vertex */
    F3 T, Lsum, acc, nee;
    float sh_limit;
    uint64_t key;
    uint32_t ctr;
    int32_t k, rebond;
    bool lastDiffuse, shadow, segment, fresh, finished;
};
__device__ __forceinline__ void fast_path_begin(FPath &q, int k0, bool live) {
    q.rC = q.shC = mkf(0, 0, 0);
    q.ru = q.shU = mkf(0, 0, 1);
    q.T = mkf(1, 1, 1);
    q.Lsum = q.acc = q.nee = mkf(0, 0, 0);
    q.sh_limit = 0; q.key = 0; q.ctr = 0; q.k = k0; q.rebond = 0;
    q.lastDiffuse = false; q.shadow = false; q.segment = true; q.fresh = true; q.finished = !live;
}

/* one iteration of the k loop up to the getColor call, raytracer.cpp:1464-1483 */
__device__ __forceinline__ void fast_path_fresh(const RenderArgs &ra, FPath &q, int i, int j, int kend) {
    if (!q.fresh || q.finished) return;
    if (q.k >= kend) { q.finished = true; return; }
    const rt_params &p = ra.p;
    const rt_camera &cam = ra.cam;
    const F3 camC = ldf(cam.C), right = ldf(cam.right), up = ldf(cam.up), view = ldf(cam.view);
    q.key = rng_key(ra.seed_mixed, (uint32_t)(i * p.W + j), (uint32_t)q.k);
    const float u1 = uniform_f(q.key, 0), u2 = uniform_f(q.key, 1), u3 = uniform_f(q.key, 2), u4 = uniform_f(q.key, 3);
    q.ctr = 4;
    float s1, c1, s3, c3;
    __sincosf(2.f * RTF_PI * u1, &s1, &c1);
    __sincosf(2.f * RTF_PI * u3, &s3, &c3);
    const float r2 = fsqrt(-2.f * __logf(u2)), r4 = fsqrt(-2.f * __logf(u4));
    const float aa = (float)cam.aa_sigma, dof = (float)cam.dof_sigma;
    const float x1 = aa * c1 * r2, x2 = aa * s1 * r2, x3 = dof * c3 * r4, x4 = dof * s3 * r4;
    F3 d = unitf(mkf((float)(j - p.W / 2) + x2 + 0.5f, (float)(i - p.H / 2) + x1 + 0.5f, (float)cam.zdist));
    d = d.x * right + d.y * up + d.z * view;
    const F3 target = camC + (float)cam.focus * d;
    q.rC = camC + mkf(x3, x4, 0);
    q.ru = unitf(target - q.rC);
    q.Lsum = mkf(0, 0, 0);
    q.T = mkf(1, 1, 1);
    q.rebond = 0;
    q.lastDiffuse = false;
    q.shadow = false;
    q.segment = true;
    q.fresh = false;
}

/* one step of Scene::getColor (raytracer.cpp:953-1093) given the answers of the rays that were traced (path_advance of
 * rt_device.hip.h, the committed estimator, rays without effect not traced) */
template <typename SP>
__device__ __forceinline__ void fast_path_advance(const DScene &sc, SP sph, const RenderArgs &ra, FPath &q, const FHit &h, float beta, float gamma, const FHit &hs) {
    const rt_params &p = ra.p;
    bool done = false;
    if (q.shadow) {
        if (!(hs.obj != INT32_MAX && hs.t < q.sh_limit)) q.Lsum = q.Lsum + q.nee; /* nee: throughput x direct light */
        q.shadow = false;
        if (q.segment) { q.rebond++; q.lastDiffuse = true; }
    }
    if (!q.segment) {
        done = true;
    } else if (q.rebond > p.max_rebond || h.obj == INT32_MAX) {
        done = true;
    } else if (h.obj == 0) { /* the light, :966-973 */
        if (q.rebond == 0 || !q.lastDiffuse) {
            const float R = (float)sc.lightR, e = (float)sc.I * frcp(4.f * RTF_PI * RTF_PI * R * R);
            q.Lsum = q.Lsum + e * q.T;
        }
        done = true;
    } else {
        FSurf s;
        fast_surface(sc, sph, q.rC, q.ru, h, beta, gamma, s);
        const F3 P = s.P, N = s.N, ru = q.ru;
        const float eps = fast_eps((float)p.epsilon, P);
        if (s.flags & RT_MAT_MIRROR) { /* :974-979 */
            q.rC = P + eps * N;
            q.ru = ru - (2.f * dotf(ru, N)) * N;
            q.rebond++;
            q.lastDiffuse = false;
        } else if (s.flags & RT_MAT_TRANSPARENT) { /* :982-1021 */
            float n1 = 1.f, n2 = (float)p.ior;
            F3 N2 = N;
            if (dotf(ru, N) > 0) { const float tmp = n1; n1 = n2; n2 = tmp; N2 = -N; }
            const float n12 = n1 * frcp(n2);
            const float rad = 1.f - n12 * n12 * (1.f - dotf(ru, N2) * dotf(ru, N2));
            bool reflect = rad < 0;
            if (!reflect) {
                const float k0 = ((n1 - n2) * (n1 - n2)) * frcp((n1 + n2) * (n1 + n2)), c = 1.f - __builtin_fabsf(dotf(N2, ru));
                const float Rf = k0 + (1.f - k0) * (c * c) * (c * c) * c;
                const float random = (float)((int)(rng_draw(q.key, q.ctr) >> 33) % 100); /* rand() % 100, :1006 */
                q.ctr++;
                reflect = (random <= 100.f * Rf);
            }
            if (reflect) { /* offset along N, not N2: :997, :1011 */
                q.rC = P + eps * N;
                q.ru = ru - (2.f * dotf(ru, N)) * N;
            } else {
                const F3 tT = n12 * (ru - dotf(ru, N2) * N2), tN = (-fsqrt(rad)) * N2;
                q.rC = P - eps * N2;
                q.ru = tT + tN;
            }
            q.rebond++;
            q.lastDiffuse = false;
        } else { /* diffuse, :1059-1088: the shadow ray and the indirect ray are traced together */
            const F3 Lpos = ldf(sc.L), lightO = ldf(sc.lightO);
            const float lightR = (float)sc.lightR, R2 = lightR * lightR;
            const float uL1 = uniform_f(q.key, q.ctr), uL2 = uniform_f(q.key, q.ctr + 1), uI1 = uniform_f(q.key, q.ctr + 2), uI2 = uniform_f(q.key, q.ctr + 3);
            q.ctr += 4;
            const F3 PL = unitf(Lpos - P);
            const F3 wv = fast_random_cos(uL1, uL2, -PL);
            const F3 xprime = lightR * wv + lightO;
            F3 Px = xprime - P;
            const float d = fsqrt(dotf(Px, Px));
            Px = frcp(d) * Px;
            const float proba = fmaxf(1e-8f, dotf(-PL, wv)) * frcp(RTF_PI * R2);
            const float J = fmaxf(0.f, dotf(wv, -Px)) * frcp(d * d);
            const float scale = ((float)sc.I * frcp(4.f * RTF_PI * RTF_PI * R2)) * (1.f / RTF_PI) * fmaxf(0.f, dotf(N, Px)) * J * frcp(proba);
            const F3 nee = scale * s.albedo;
            /* The shadow ray starts eps off the surface but aims as if from P, so it meets the light a little beside the sample;
             * on a grazing sample that moves the hit by many eps, and whether such a sample still counts as lit depends on the
             * ratio shadow_bias / epsilon (10 in the reference, raytracer.cpp:955, :1073). The raised offset keeps that ratio,
             * so the same share of grazing samples is lost as in the reference. */
            q.sh_limit = d - fmaxf((float)p.shadow_bias, (float)(p.shadow_bias / p.epsilon) * eps);
            if (nee.x == 0 && nee.y == 0 && nee.z == 0) q.sh_limit = 0; /* nothing to add: the shadow ray is not traced */
            q.shC = P + eps * N;
            q.shU = Px;
            q.shadow = true;
            q.nee = q.T * nee;
            q.T = q.T * s.albedo;
            q.ru = fast_random_cos(uI1, uI2, N);
            q.rC = P + eps * N;
            q.segment = q.rebond < p.max_rebond; /* the ray beyond the depth limit is intersected and dropped (:960-963): not traced */
        }
    }
    if (!q.shadow && q.rebond > p.max_rebond) done = true;
    if (done) {
        q.acc = q.acc + q.Lsum;
        q.k++;
        q.fresh = true;
    }
}

/* The render loop of a persistent wave (render_persistent of rt_device.hip.h): lanes take 8 x 8 tiles of pixel slots from a
 * global counter, a lane keeps its pixel for a unit of samples, adds them in order and stores the unit's sum. */
template <typename SP>
__device__ __forceinline__ void fast_render_persistent(const DScene &sc, const DScene &cold, FastLDS *w, SP sph, const FastTri *ftris, const RenderArgs &ra, int k0, int nk,
                                                       int chunk, double *accum, double *unit_sums, int *next_slot, int n_slots) {
    FPath q;
    fast_path_begin(q, k0, false);
    const int kend = k0 + nk;
    const int n_chunks = (nk + chunk - 1) / chunk;
    const int n_total = n_slots * n_chunks;
    int kend_l = kend, unit_c = 0;
    bool have = false, exhausted = false;
    int i = 0, j = 0;
    int cur = 0, end = 0, tile_c = 0, tile_i = 0, tile_j = 0;
    while (true) {
        if (have && q.finished) {
            if (n_chunks > 1) {
                double *us = unit_sums + (((int64_t)unit_c * (ra.i1 - ra.i0) + (i - ra.i0)) * (ra.j1 - ra.j0) + (j - ra.j0)) * 3;
                us[0] = (double)q.acc.x;
                us[1] = (double)q.acc.y;
                us[2] = (double)q.acc.z;
            } else {
                double *px = accum + ((int64_t)i * ra.p.W + j) * 3;
                px[0] = px[0] + (double)q.acc.x;
                px[1] = px[1] + (double)q.acc.y;
                px[2] = px[2] + (double)q.acc.z;
            }
            have = false;
        }
        const unsigned long long need = wave_ballot(!have && !exhausted);
        if (need) {
            if (cur == end) {
                int base = 0;
                if (lane_id() == 0) base = atomicAdd(next_slot, RT_WAVE);
                cur = uni(base);
                end = cur + RT_WAVE < n_total ? cur + RT_WAVE : n_total;
                if (cur >= n_total) { cur = end = n_total; }
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
                    fast_path_begin(q, ks, live);
                    have = live;
                }
            }
            const int taken = __popcll(need);
            cur = (cur + taken < end) ? cur + taken : end;
        }
        fast_path_fresh(ra, q, i, j, kend_l);
        const bool tracing = !q.finished;
        if (wave_ballot(tracing) == 0ull) {
            if (wave_ballot(!exhausted || have) == 0ull) break;
            continue;
        }
        FHit h, hs;
        float beta, gamma;
        const bool act_a = tracing && q.segment, act_b = tracing && q.shadow && (q.sh_limit > 0);
        fast_closest(sc, cold, w, sph, ftris, act_a, q.rC, q.ru, act_b, q.shC, q.shU, q.sh_limit, h, hs, beta, gamma);
        if (!act_a) h.obj = INT32_MAX;
        if (!act_b) hs.obj = INT32_MAX;
        if (tracing) fast_path_advance(cold, sph, ra, q, h, beta, gamma, hs);
    }
}

} // namespace rtf
