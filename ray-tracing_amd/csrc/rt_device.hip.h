/*
 * rt_device.hip.h - device code of the path-tracing hot path for gfx950.
 *
 * FP64 throughout, compiled with -ffp-contract=off: every expression keeps
 * the operation order of the reference expression it replaces (cited per
 * function), so hit parameters, hit identities and noise-free radiance equal
 * the CPU results bit for bit up to libm differences in cos/sin/log/pow.
 *
 * Structure (one lane = one pixel, persistent over its samples):
 *   - the recursive Scene::getColor (raytracer.cpp:953-1093) becomes an
 *     iterative state machine with exactly ONE closest-hit query per loop
 *     iteration. A lane is either on a path segment or on the shadow ray of
 *     its last diffuse vertex; both kinds of ray go through the same query, so
 *     the lanes of a wave stay busy whatever mix of segments they are on, and a
 *     lane whose path ends starts its next sample at once.
 *   - radiance is carried as (sum, throughput): sum += throughput * direct,
 *     throughput *= albedo, which re-associates the reference's
 *     color += albedo * getColor(...) (raytracer.cpp:1088) by rounding only.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"

namespace rtd {

#define RT_STACK_MAX 64
#define RT_LDS_SPHERES_MAX 1024

struct DScene {
    const rt_sphere *spheres;
    const rt_mesh *meshes;
    const rt_bvh_node *nodes;
    const rt_tri *tris;
    const rt_tri_attr *attrs;
    const rt_texture *tex;
    const uint8_t *texels;
    const double *lut;
    int32_t n_spheres, n_meshes;
    double I;
    double L[3];
    double lightO[3];
    double lightR;
};

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
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
};

/* BoudingBox::intersect, raytracer.cpp:163-186 (six true divisions, no bound on t) */
__device__ __forceinline__ bool box_hit(const double *b, V3 C, V3 u) {
    double t1x = (b[0] - C.x) / u.x, t2x = (b[3] - C.x) / u.x;
    double txMin = smin(t1x, t2x), txMax = smax(t1x, t2x);
    double t1y = (b[1] - C.y) / u.y, t2y = (b[4] - C.y) / u.y;
    double tyMin = smin(t1y, t2y), tyMax = smax(t1y, t2y);
    double t1z = (b[2] - C.z) / u.z, t2z = (b[5] - C.z) / u.z;
    double tzMin = smin(t1z, t2z), tzMax = smax(t1z, t2z);
    double tMax = smin(txMax, smin(tyMax, tzMax));
    double tMin = smax(txMin, smax(tyMin, tzMin));
    if (tMax < 0) return false;
    return tMax > tMin;
}

/* closest hit found so far, in the form that lets the attributes be computed once at the end */
struct HitRec {
    double t;
    int32_t obj;  /* objectId, INT32_MAX while nothing is hit */
    int32_t prim; /* sphere index, or global permuted triangle index */
    int32_t mesh; /* mesh index or -1 for a sphere */
    double beta, gamma;
};

/* TriangleMesh::intersect traversal + triangle test, raytracer.cpp:642-692.
 * Same visiting order as the reference (the child pushed last, fd, is popped first; ascending i in
 * a leaf; strict '<' keeps the first of equal t). */
template <bool CNT>
__device__ __forceinline__ bool mesh_traverse(const DScene &sc, const rt_mesh &m, V3 C, V3 u, double &t_out,
                                              int32_t &tri_out, double &beta_out, double &gamma_out, Cnt &cnt) {
    int32_t stack_idx[RT_STACK_MAX];
    int32_t stack_cnt[RT_STACK_MAX];
    int sp = 0;
    stack_idx[0] = m.root;
    stack_cnt[0] = m.root_count;
    sp = 1;
    double t = 1E9;
    int32_t tri = -1;
    double bb = 0, gg = 0;
    bool has_inter = false;
    while (sp > 0) {
        sp--;
        int32_t idx = stack_idx[sp];
        int32_t n = stack_cnt[sp];
        if (n == 0) {
            const rt_bvh_node *nd = sc.nodes + idx;
            if (CNT) { cnt.nodes++; cnt.box_tests += 2; }
            bool hl = box_hit(nd->boxL, C, u);
            bool hr = box_hit(nd->boxR, C, u);
            if (hl && sp < RT_STACK_MAX) { stack_idx[sp] = nd->L; stack_cnt[sp] = nd->nL; sp++; }
            if (hr && sp < RT_STACK_MAX) { stack_idx[sp] = nd->R; stack_cnt[sp] = nd->nR; sp++; }
        } else {
            for (int32_t i = idx; i < idx + n; i++) {
                if (CNT) cnt.tris++;
                const rt_tri *tr = sc.tris + i;
                V3 A = ld3(tr->A), e1 = ld3(tr->e1), e2 = ld3(tr->e2);
                V3 N = cross(e1, e2), AO = sub(C, A), AOu = cross(AO, u);
                double invUN = 1. / dot(u, N);
                double beta = -dot(e2, AOu) * invUN, gamma = dot(e1, AOu) * invUN, alpha = 1 - beta - gamma,
                       localt = -dot(AO, N) * invUN;
                if (beta >= 0 && gamma >= 0 && beta <= 1 && gamma <= 1 && alpha >= 0 && localt > 0) {
                    has_inter = true;
                    if (localt < t) { t = localt; tri = i; bb = beta; gg = gamma; }
                }
            }
        }
    }
    t_out = t; tri_out = tri; beta_out = bb; gamma_out = gg;
    return has_inter;
}

/* Scene::intersect, raytracer.cpp:930-951, over spheres (Sphere::intersect :883-918) and meshes.
 * Objects are visited spheres first, then meshes; "lowest objectId wins an exact tie" is what the
 * reference's in-order loop with strict '<' produces, and is applied explicitly here. */
template <bool CNT>
__device__ __forceinline__ bool scene_closest(const DScene &sc, const rt_sphere *sph, V3 C, V3 u, HitRec &h, Cnt &cnt) {
    if (CNT) cnt.rays++;
    h.t = 1E10;
    h.obj = INT32_MAX;
    h.prim = -1;
    h.mesh = -1;
    h.beta = h.gamma = 0;
    for (int s = 0; s < sc.n_spheres; s++) {
        V3 O = ld3(sph[s].O);
        double R = sph[s].R;
        int32_t oid = sph[s].object_id;
        double a = 1;
        V3 CO = sub(C, O);
        double b = 2 * dot(u, CO);
        double c = norm2(CO) - R * R;
        double delta = b * b - 4 * a * c;
        if (delta < 0) continue;
        double sq = sqrt(delta);
        double t2 = (-b + sq) / (2 * a);
        if (t2 < 0) continue;
        double t1 = (-b - sq) / (2 * a);
        double t = (t1 > 0) ? t1 : t2;
        if (t < h.t || (t == h.t && oid < h.obj)) { h.t = t; h.obj = oid; h.prim = s; h.mesh = -1; }
    }
    for (int mi = 0; mi < sc.n_meshes; mi++) {
        const rt_mesh &m = sc.meshes[mi];
        if (CNT) cnt.box_tests++;
        if (!box_hit(m.root_box, C, u)) continue;
        if (CNT) cnt.mesh_calls++;
        double t, bb, gg;
        int32_t tri;
        bool ok = mesh_traverse<CNT>(sc, m, C, u, t, tri, bb, gg, cnt);
        /* ok with tri < 0 needs an accepted triangle with t >= 1e9 (raytracer.cpp:689-690); not a usable hit */
        if (ok && tri >= 0 && (t < h.t || (t == h.t && m.object_id < h.obj))) {
            h.t = t; h.obj = m.object_id; h.prim = tri; h.mesh = mi; h.beta = bb; h.gamma = gg;
        }
    }
    return h.obj != INT32_MAX;
}

struct Surf {
    V3 P, N, albedo;
    int32_t flags;
};

/* what Scene::intersect hands back for the winning object: P, N, colour, material */
__device__ __forceinline__ void hit_surface(const DScene &sc, const rt_sphere *sph, V3 C, V3 u, const HitRec &h, Surf &s) {
    if (h.mesh < 0) { /* Sphere::intersect :914-915 */
        V3 O = ld3(sph[h.prim].O);
        s.P = add(C, smul(h.t, u));
        s.N = unit(sub(s.P, O));
        s.albedo = ld3(sph[h.prim].albedo);
        s.flags = sph[h.prim].flags;
    } else { /* TriangleMesh::intersect :693-720 */
        const rt_mesh &m = sc.meshes[h.mesh];
        const rt_tri_attr *at = sc.attrs + h.prim;
        double beta = h.beta, gamma = h.gamma, alpha = 1 - beta - gamma;
        V3 nn = add(add(smul(alpha, ld3(at->n[0])), smul(beta, ld3(at->n[1]))), smul(gamma, ld3(at->n[2])));
        s.N = unit(nn);
        s.P = add(C, u); /* raytracer.cpp:696: C + u, not C + t u */
        s.flags = m.flags;
        if (m.n_tex > 0) {
            const rt_texture tx = sc.tex[m.first_tex + at->group];
            int H = tx.H, W = tx.W;
            double U = (alpha * at->uv[0][0] + beta * at->uv[1][0]) + gamma * at->uv[2][0];
            double V = (alpha * at->uv[0][1] + beta * at->uv[1][1]) + gamma * at->uv[2][1];
            U = U * (double)W;
            V = V * (double)H;
            int uvx = (int)(U + 0.5);
            int uvy = (int)(V + 0.5);
            uvx = uvx % W;
            uvy = uvy % H;
            if (uvx < 0) uvx += W;
            if (uvy < 0) uvy += H;
            uvy = H - uvy - 1;
            const uint8_t *px = sc.texels + tx.offset + ((int64_t)uvy * W + uvx) * 3;
            s.albedo = mk(sc.lut[px[0]], sc.lut[px[1]], sc.lut[px[2]]);
        } else {
            s.albedo = ld3(m.albedo);
        }
    }
}

/* random_cos, raytracer.cpp:112-138 */
__device__ __forceinline__ V3 random_cos(double u1, double u2, V3 N) {
    double x = cos(2 * RT_PI * u1) * sqrt(1 - u2);
    double y = sin(2 * RT_PI * u1) * sqrt(1 - u2);
    double z = sqrt(u2);
    V3 T1;
    if (N.x < N.y && N.x < N.z) T1 = mk(0, N.z, -N.y);
    else if (N.y < N.z && N.y < N.x) T1 = mk(N.z, 0, -N.x);
    else T1 = mk(N.y, -N.x, 0);
    T1 = unit(T1);
    V3 T2 = cross(N, T1);
    return add(add(smul(z, N), smul(x, T1)), smul(y, T2));
}

struct RenderArgs {
    rt_camera cam;
    rt_params p;
    uint64_t seed_mixed; /* mix64(seed + golden) */
    int32_t j0, i0, j1, i1;
};

/* Sum of the radiance of samples k0 .. k0+nk-1 of pixel (i, j), added in that order
 * (the k loop of raytracer.cpp:1464-1486). */
template <bool CNT>
__device__ __forceinline__ V3 trace_pixel(const DScene &sc, const rt_sphere *sph, const RenderArgs &ra, int i, int j,
                                          int k0, int nk, Cnt &cnt) {
    const rt_params &p = ra.p;
    const rt_camera &cam = ra.cam;
    const uint32_t fl = p.flags;
    const double epsilon = p.epsilon;
    const V3 camC = ld3(cam.C), right = ld3(cam.right), up = ld3(cam.up), view = ld3(cam.view);
    const V3 Lpos = ld3(sc.L);
    const V3 lightO = ld3(sc.lightO);
    const double lightR = sc.lightR;
    const uint32_t pixel = (uint32_t)(i * p.W + j);

    V3 acc = mk(0, 0, 0);
    int k = k0;
    const int kend = k0 + nk;

    /* per-path state */
    uint64_t key = 0;
    uint32_t ctr = 0;
    V3 rC = mk(0, 0, 0), ru = mk(0, 0, 0); /* ray being traced */
    V3 Lsum = mk(0, 0, 0), T = mk(1, 1, 1);
    int rebond = 0;
    bool lastDiffuse = false;
    bool shadow = false; /* the ray being traced is the shadow ray of a diffuse vertex */
    /* pending state of the diffuse vertex while its shadow ray is traced */
    V3 nee = mk(0, 0, 0), nextC = mk(0, 0, 0), nextU = mk(0, 0, 0), vAlbedo = mk(0, 0, 0);
    double sh_limit = 0;
    bool fresh = true;

    while (true) {
        if (fresh) {
            if (k >= kend) break;
            if (CNT) cnt.paths++;
            key = rng_key(ra.seed_mixed, pixel, (uint32_t)k);
            ctr = 0;
            if (fl & RT_F_NO_JITTER) { /* Ray r(C, u), raytracer.cpp:1482 */
                V3 d = mk(j - p.W / 2 + 0.5, i - p.H / 2 + 0.5, cam.zdist);
                d = unit(d);
                ru = add(add(smul(d.x, right), smul(d.y, up)), smul(d.z, view));
                rC = camC;
            } else { /* raytracer.cpp:1467-1483 */
                double u1 = rng_uniform(key, 0), u2 = rng_uniform(key, 1);
                double x1 = cam.aa_sigma * cos(2 * RT_PI * u1) * sqrt(-2 * log(u2)),
                       x2 = cam.aa_sigma * sin(2 * RT_PI * u1) * sqrt(-2 * log(u2));
                double u3 = rng_uniform(key, 2), u4 = rng_uniform(key, 3);
                double x3 = cam.dof_sigma * cos(2 * RT_PI * u3) * sqrt(-2 * log(u4)),
                       x4 = cam.dof_sigma * sin(2 * RT_PI * u3) * sqrt(-2 * log(u4));
                ctr = 4;
                V3 d = mk(j - p.W / 2 + x2 + 0.5, i - p.H / 2 + x1 + 0.5, cam.zdist);
                d = unit(d);
                d = add(add(smul(d.x, right), smul(d.y, up)), smul(d.z, view));
                V3 target = add(camC, smul(cam.focus, d));
                rC = add(camC, mk(x3, x4, 0));
                ru = unit(sub(target, rC));
            }
            Lsum = mk(0, 0, 0);
            T = mk(1, 1, 1);
            rebond = 0;
            lastDiffuse = false;
            shadow = false;
            fresh = false;
        }

        HitRec h;
        bool inter = scene_closest<CNT>(sc, sph, rC, ru, h, cnt);

        if (shadow) {
            /* raytracer.cpp:1073-1083 (or :1048-1055 for the point light): occluded iff hit before the light */
            if (!(inter && h.t < sh_limit)) Lsum = add(Lsum, cmul(T, nee));
            shadow = false;
            if (fl & RT_F_NO_INDIRECT) {
                acc = add(acc, Lsum);
                k++;
                fresh = true;
                continue;
            }
            /* color += albedo * getColor(wiRay, rebond + 1, true), raytracer.cpp:1086-1088 */
            T = cmul(T, vAlbedo);
            rC = nextC;
            ru = nextU;
            rebond++;
            lastDiffuse = true;
            continue;
        }

        /* Scene::getColor, raytracer.cpp:953-1093; the query above comes before the depth test (:960-963) */
        bool done = false;
        if (rebond > p.max_rebond || !inter) {
            done = true;
        } else if (h.obj == 0 && !(fl & RT_F_POINT_LIGHT)) { /* the light, :966-973 */
            if (rebond == 0 || !lastDiffuse) {
                double e = sc.I / (4 * RT_PI * RT_PI * sqr(lightR));
                Lsum = add(Lsum, cmul(T, mk(e, e, e)));
            }
            done = true;
        } else {
            Surf s;
            hit_surface(sc, sph, rC, ru, h, s);
            const V3 P = s.P, N = s.N;
            if (s.flags & RT_MAT_MIRROR) { /* :974-979 */
                if (CNT) cnt.specular++;
                V3 rd = sub(ru, smul(2 * dot(ru, N), N));
                rC = add(P, smul(epsilon, N));
                ru = rd;
                rebond++;
                lastDiffuse = false;
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
                    double random = (double)((int)(rng_draw(key, ctr) >> 33) % 100); /* rand() % 100, :1006 */
                    ctr++;
                    reflect = (random <= 100 * R);
                }
                if (reflect) { /* offset along N, not N2: :997, :1011 */
                    V3 rd = sub(ru, smul(2 * dot(ru, N), N));
                    rC = add(P, smul(epsilon, N));
                    ru = rd;
                } else {
                    V3 tT = smul(n1 / n2, sub(ru, smul(dot(ru, N2), N2)));
                    V3 tN = smul(-sqrt(rad), N2);
                    rC = sub(P, smul(epsilon, N2));
                    ru = add(tT, tN);
                }
                rebond++;
                lastDiffuse = false;
            } else { /* diffuse: direct light now, indirect ray after the shadow query */
                if (CNT) cnt.diffuse++;
                V3 shC = add(P, smul(0.00001, N)), shU;
                if (fl & RT_F_POINT_LIGHT) { /* :1040-1055 */
                    V3 PL = sub(Lpos, P);
                    double d = sqrt(norm2(PL));
                    shU = divs(PL, d);
                    nee = muls(smul(sc.I / (4 * RT_PI * d * d), divs(s.albedo, RT_PI)), smax(0., dot(N, divs(PL, d))));
                    sh_limit = d;
                } else { /* :1059-1083 */
                    V3 PL = unit(sub(Lpos, P));
                    double u1 = rng_uniform(key, ctr), u2 = rng_uniform(key, ctr + 1);
                    ctr += 2;
                    V3 w = random_cos(u1, u2, neg(PL));
                    V3 xprime = add(muls(w, lightR), lightO);
                    V3 Px = sub(xprime, P);
                    double d = sqrt(norm2(Px));
                    Px = divs(Px, d);
                    shU = Px;
                    double R2 = sqr(lightR);
                    double proba = smax(1E-8, dot(neg(PL), w)) / (RT_PI * R2);
                    double J = smax(0., dot(w, neg(Px))) / (d * d);
                    nee = divs(muls(muls(smul(sc.I / (4 * RT_PI * RT_PI * R2), divs(s.albedo, RT_PI)), smax(0., dot(N, Px))), J), proba);
                    sh_limit = d - p.shadow_bias;
                }
                if (!(fl & RT_F_NO_INDIRECT)) { /* :1086-1087; its two draws follow the light's two */
                    double u1 = rng_uniform(key, ctr), u2 = rng_uniform(key, ctr + 1);
                    ctr += 2;
                    nextU = random_cos(u1, u2, N);
                    nextC = add(P, smul(epsilon, N));
                    vAlbedo = s.albedo;
                }
                rC = shC;
                ru = shU;
                shadow = true;
            }
        }
        if (done) {
            acc = add(acc, Lsum);
            k++;
            fresh = true;
        }
    }
    return acc;
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
}

} // namespace rtd
