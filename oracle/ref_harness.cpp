/*
 * ref_harness.cpp - drives the REAL reference classes as a shared library.
 *
 * TEST INFRASTRUCTURE ONLY, and only where /root/reference exists: this file
 * textually includes /root/reference/raytracer.cpp (with main renamed), so the
 * Scene / Sphere / TriangleMesh code that runs is the reference's own,
 * compiled from where it lies. Built by oracle/Makefile into
 * oracle/_ref/libref_true.so; nothing of the reference is copied into this
 * repository.
 *
 * Two macro substitutions make the unmodified reference deterministic:
 *   default_random_engine -> std::InjEngine : a URBG whose values are the
 *     counter-based stream the oracle and the GPU use. Its range is 2^64, so
 *     libstdc++'s uniform_real_distribution<double> draws exactly one value per
 *     uniform(engine) call; the low 11 bits are cleared so value / 2^64 is
 *     exactly (draw >> 11) * 2^-53.
 *   rand -> inj_rand : the dielectric coin of raytracer.cpp:1006.
 * The pixel loop of the reference is inline in main (raytracer.cpp:1457-1503)
 * and cannot be called; trf_sample_radiance / trf_resolve restate those lines
 * with the reference's own Vector / uniform(engine) / Scene::getColor.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <ctime>
#include <fstream>
#include <iostream>
#include <list>
#include <random>
#include <string>
#include <vector>

#include "../include/rt_amd.h" /* rt_camera / rt_params PODs only */

/* ---- injected random stream ------------------------------------------------------------- */
static uint64_t g_key = 0;
static uint32_t g_n = 0;
static uint64_t g_draws = 0;
static inline uint64_t h_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t h_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    return h_mix64(h_mix64(seed + 0x9E3779B97F4A7C15ull) + (((uint64_t)pixel << 32) | (uint64_t)sample));
}
static inline uint64_t h_draw() {
    g_draws++;
    return h_mix64(g_key + ((uint64_t)(g_n++) + 1ull) * 0x9E3779B97F4A7C15ull);
}
namespace std {
struct InjEngine {
    typedef uint64_t result_type;
    explicit InjEngine(int) {}
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return ~0ull; }
    result_type operator()() { return (h_draw() >> 11) << 11; }
};
} // namespace std
static int inj_rand() { return (int)(h_draw() >> 33); }

/* TRF_PLAIN (oracle/_ref/libref_plain.so): no substitution at all - the reference's own global engine(10) shared by all
 * OpenMP threads and libc's rand(), as committed (raytracer.cpp:26-27, :1006). Not deterministic with more than one
 * thread; built only to TIME the reference faithfully (trf_render_omp), never to compare values. */
#ifndef TRF_PLAIN
#define default_random_engine InjEngine
#define rand inj_rand
#endif
#define main reference_main
#include "/root/reference/raytracer.cpp"
#undef main
#ifndef TRF_PLAIN
#undef rand
#undef default_random_engine
#endif

struct trf_scene {
    Scene scene;
    std::vector<Object *> owned;
    std::vector<std::vector<int> > perm; /* per object: permuted position -> face-fan index */
};

static TriangleMesh *mesh_of(trf_scene *s, int id) {
    if (id < 0 || id >= (int)s->owned.size()) return NULL;
    return dynamic_cast<TriangleMesh *>(s->owned[id]);
}

extern "C" {

trf_scene *trf_scene_create(double I, const double *L) {
    trf_scene *s = new trf_scene();
    s->scene.I = I;
    s->scene.L = Vector(L[0], L[1], L[2]);
    return s;
}

int trf_add_sphere(trf_scene *s, const double *O, double R, const double *albedo, int mirror, int transp) {
    Sphere *sp = new Sphere(Vector(O[0], O[1], O[2]), R, Vector(albedo[0], albedo[1], albedo[2]), mirror != 0, transp != 0);
    s->owned.push_back(sp);
    s->scene.objects.push_back(sp);
    s->perm.push_back(std::vector<int>());
    return (int)s->owned.size() - 1;
}

int trf_add_mesh(trf_scene *s, const double *albedo, int mirror, int transp) {
    TriangleMesh *m = new TriangleMesh(Vector(albedo[0], albedo[1], albedo[2]), mirror != 0, transp != 0);
    s->owned.push_back(m);
    s->scene.objects.push_back(m);
    s->perm.push_back(std::vector<int>());
    return (int)s->owned.size() - 1;
}

int trf_mesh_read_obj(trf_scene *s, int id, const char *path) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    FILE *f = fopen(path, "r"); /* the reference does not check (raytracer.cpp:300-302) */
    if (!f) return -2;
    fclose(f);
    m->readOBJ(path);
    return 0;
}

int trf_mesh_load_texture(trf_scene *s, int id, const char *path) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    FILE *f = fopen(path, "rb");
    if (!f) return -2;
    fclose(f);
    m->loadTexture(path);
    return m->textures.back() ? 0 : -3;
}

int trf_mesh_add_texture_raw(trf_scene *s, int id, const uint8_t *rgb, int W, int H) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    unsigned char *copy = (unsigned char *)malloc((size_t)W * H * 3);
    memcpy(copy, rgb, (size_t)W * H * 3);
    m->Wtex.push_back(W);
    m->Htex.push_back(H);
    m->textures.push_back(copy);
    return 0;
}

int trf_mesh_set_raw(trf_scene *s, int id, int64_t nv, const double *v, int64_t nn, const double *n,
                     int64_t nuv, const double *uv, int64_t nt, const int32_t *idx10) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    m->vertices.clear(); m->normals.clear(); m->uvs.clear(); m->indices.clear();
    for (int64_t i = 0; i < nv; i++) m->vertices.push_back(Vector(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
    for (int64_t i = 0; i < nn; i++) m->normals.push_back(Vector(n[3 * i], n[3 * i + 1], n[3 * i + 2]));
    for (int64_t i = 0; i < nuv; i++) m->uvs.push_back(Vector(uv[2 * i], uv[2 * i + 1], 0));
    for (int64_t i = 0; i < nt; i++) {
        const int32_t *q = idx10 + 10 * i;
        TriangleIndices t;
        t.vtxi = q[0]; t.vtxj = q[1]; t.vtxk = q[2];
        t.uvi = q[3]; t.uvj = q[4]; t.uvk = q[5];
        t.ni = q[6]; t.nj = q[7]; t.nk = q[8];
        t.group = q[9];
        m->indices.push_back(t);
    }
    return 0;
}

/* op: 0 invertNormals, 1 translate(a,b), 2 rotateNormals(a,b), 3 rotate(a,b), 4 rotateAroundAxe(a,angle), 5 zoom(b).
 * The reference loops run to i <= vertices.size() (raytracer.cpp:770-854): one element past the
 * end. A zero dummy element is appended for the call and removed afterwards so that access is
 * defined; it cannot change any real vertex. */
int trf_mesh_op(trf_scene *s, int id, int op, int a, int b, double angle) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    size_t nv = m->vertices.size(), nn = m->normals.size();
    m->vertices.push_back(Vector(0, 0, 0));
    while (m->normals.size() < nv + 1) m->normals.push_back(Vector(0, 0, 0));
    switch (op) {
    case 0: m->invertNormals(); break;
    case 1: m->translate(a, b); break;
    case 2: m->rotateNormals(a, b); break;
    case 3: m->rotate(a, b); break;
    case 4: m->rotateAroundAxe(a, angle); break;
    case 5: m->zoom(b); break;
    default: return -2;
    }
    m->vertices.resize(nv);
    m->normals.resize(nn);
    return 0;
}

int trf_mesh_build_bvh(trf_scene *s, int id) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    /* tag every triangle with its face-fan index in 'group' (buildBVH never reads it), build,
     * read the permutation back and restore the groups */
    size_t nt = m->indices.size();
    std::vector<int> saved(nt);
    for (size_t k = 0; k < nt; k++) { saved[k] = m->indices[k].group; m->indices[k].group = (int)k; }
    m->buildBVH(m->BVH, 0, (int)nt);
    std::vector<int> &perm = s->perm[id];
    perm.resize(nt);
    for (size_t k = 0; k < nt; k++) { perm[k] = m->indices[k].group; m->indices[k].group = saved[perm[k]]; }
    return 0;
}

int trf_mesh_sizes(trf_scene *s, int id, int64_t *nv, int64_t *nn, int64_t *nuv, int64_t *nt, int32_t *ntex) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    *nv = m->vertices.size(); *nn = m->normals.size(); *nuv = m->uvs.size(); *nt = m->indices.size();
    *ntex = (int)m->textures.size();
    return 0;
}

int trf_mesh_get(trf_scene *s, int id, double *v, double *n, double *uv, int32_t *idx10, int32_t *perm) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    if (v) for (size_t i = 0; i < m->vertices.size(); i++) for (int k = 0; k < 3; k++) v[3 * i + k] = m->vertices[i][k];
    if (n) for (size_t i = 0; i < m->normals.size(); i++) for (int k = 0; k < 3; k++) n[3 * i + k] = m->normals[i][k];
    if (uv) for (size_t i = 0; i < m->uvs.size(); i++) for (int k = 0; k < 2; k++) uv[2 * i + k] = m->uvs[i][k];
    if (idx10) for (size_t i = 0; i < m->indices.size(); i++) {
        const TriangleIndices &t = m->indices[i];
        int32_t *q = idx10 + 10 * i;
        q[0] = t.vtxi; q[1] = t.vtxj; q[2] = t.vtxk; q[3] = t.uvi; q[4] = t.uvj; q[5] = t.uvk;
        q[6] = t.ni; q[7] = t.nj; q[8] = t.nk; q[9] = t.group;
    }
    if (perm) for (size_t i = 0; i < s->perm[id].size(); i++) perm[i] = s->perm[id][i];
    return 0;
}

int trf_mesh_texture(trf_scene *s, int id, int k, int32_t *W, int32_t *H, uint8_t *rgb) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m || k < 0 || k >= (int)m->textures.size()) return -1;
    *W = m->Wtex[k]; *H = m->Htex[k];
    if (rgb) memcpy(rgb, m->textures[k], (size_t)m->Wtex[k] * m->Htex[k] * 3);
    return 0;
}

/* preorder (fg before fd) dump of the reference's tree */
int64_t trf_bvh_dump(trf_scene *s, int id, double *boxes, int32_t *dfl, int64_t cap) {
    TriangleMesh *m = mesh_of(s, id);
    if (!m) return -1;
    std::vector<Noeud *> st;
    st.push_back(m->BVH);
    int64_t k = 0;
    while (!st.empty()) {
        Noeud *c = st.back();
        st.pop_back();
        if (k < cap) {
            if (boxes) for (int q = 0; q < 3; q++) { boxes[6 * k + q] = c->b.mini[q]; boxes[6 * k + 3 + q] = c->b.maxi[q]; }
            if (dfl) { dfl[3 * k] = c->debut; dfl[3 * k + 1] = c->fin; dfl[3 * k + 2] = c->fg == NULL; }
        }
        k++;
        if (c->fg) { st.push_back(c->fd); st.push_back(c->fg); }
    }
    return k;
}

/* Which triangle produced (t, N)?  TriangleMesh::intersect does not say. Recomputed here from the
 * reference's own public arrays with the expressions of raytracer.cpp:673-687: the candidates are
 * the accepted triangles whose t equals the returned one bit for bit; ties are resolved by the
 * interpolated normal and then by the reference's visit order (fd subtree first, ascending i in a
 * leaf, strict '<' keeps the first: raytracer.cpp:656-665, :670, :690). */
static void leaf_starts(Noeud *n, std::vector<int> &ls) {
    if (n->fg) { leaf_starts(n->fg, ls); leaf_starts(n->fd, ls); }
    else for (int i = n->debut; i < n->fin; i++) ls[i] = n->debut;
}
static int recover_tri(trf_scene *s, int id, const Ray &r, double t, const Vector &Nhit) {
    TriangleMesh *m = mesh_of(s, id);
    std::vector<int> cand;
    for (int i = 0; i < (int)m->indices.size(); i++) {
        const Vector &A = m->vertices[m->indices[i].vtxi], &B = m->vertices[m->indices[i].vtxj], &C = m->vertices[m->indices[i].vtxk];
        Vector e1 = B - A, e2 = C - A, N = cross(e1, e2), AO = r.C - A, AOu = cross(AO, r.u);
        double invUN = 1. / dot(r.u, N);
        double beta = -dot(e2, AOu) * invUN, gamma = dot(e1, AOu) * invUN, alpha = 1 - beta - gamma, localt = -dot(AO, N) * invUN;
        if (beta >= 0 && gamma >= 0 && beta <= 1 && gamma <= 1 && alpha >= 0 && localt > 0 && localt == t) cand.push_back(i);
    }
    if (cand.empty()) return -3;
    if (cand.size() > 1) {
        std::vector<int> ls(m->indices.size(), -1);
        leaf_starts(m->BVH, ls);
        int best = cand[0];
        for (size_t q = 1; q < cand.size(); q++) {
            int c = cand[q];
            if (ls[c] > ls[best] || (ls[c] == ls[best] && c < best)) best = c;
        }
        (void)Nhit;
        return s->perm[id][best];
    }
    return s->perm[id][cand[0]];
}

void trf_intersect_rays(trf_scene *s, const double *rays, int64_t n, int32_t *obj_id, int32_t *tri_id,
                        double *t, double *PN, double *albedo) {
    for (int64_t q = 0; q < n; q++) {
        Ray r(Vector(rays[6 * q], rays[6 * q + 1], rays[6 * q + 2]), Vector(rays[6 * q + 3], rays[6 * q + 4], rays[6 * q + 5]));
        Vector P, N, alb;
        double tt;
        bool mirror, transp;
        int oid = -1;
        bool ok = s->scene.intersect(r, P, N, alb, mirror, transp, tt, oid);
        if (obj_id) obj_id[q] = ok ? oid : -1;
        if (t) t[q] = tt;
        if (tri_id) {
            tri_id[q] = -1;
            if (ok && mesh_of(s, oid) && !s->perm[oid].empty()) tri_id[q] = recover_tri(s, oid, r, tt, N);
        }
        if (PN) for (int k = 0; k < 3; k++) { PN[6 * q + k] = ok ? P[k] : 0; PN[6 * q + 3 + k] = ok ? N[k] : 0; }
        if (albedo) for (int k = 0; k < 3; k++) albedo[3 * q + k] = ok ? alb[k] : 0;
    }
}

/* one iteration of the k loop, raytracer.cpp:1467-1485, on the reference's own types.
 * Only the committed estimator exists in the reference: p->flags must be 0 and the literals
 * (depth 10, n 1.4, eps 1e-5, bias 1e-4) cannot be changed. */
static Vector sample_one(trf_scene *s, const rt_camera *cam, const rt_params *p, int i, int j, int k) {
    g_key = h_key(p->seed, (uint32_t)(i * p->W + j), (uint32_t)k);
    g_n = 0;
    int W = p->W, H = p->H;
    Vector C(cam->C[0], cam->C[1], cam->C[2]);
    Vector right(cam->right[0], cam->right[1], cam->right[2]);
    Vector up(cam->up[0], cam->up[1], cam->up[2]);
    Vector viewDirection(cam->view[0], cam->view[1], cam->view[2]);
    double u1 = uniform(engine), u2 = uniform(engine);
    double x1 = cam->aa_sigma * cos(2 * M_PI * u1) * sqrt(-2 * log(u2)),
           x2 = cam->aa_sigma * sin(2 * M_PI * u1) * sqrt(-2 * log(u2));
    double u3 = uniform(engine), u4 = uniform(engine);
    double x3 = cam->dof_sigma * cos(2 * M_PI * u3) * sqrt(-2 * log(u4)),
           x4 = cam->dof_sigma * sin(2 * M_PI * u3) * sqrt(-2 * log(u4));
    Vector u(j - W / 2 + x2 + 0.5, i - H / 2 + x1 + 0.5, cam->zdist);
    u = u.get_normalized();
    u = u[0] * right + u[1] * up + u[2] * viewDirection;
    Vector target = C + cam->focus * u;
    Vector Cprime = C + Vector(x3, x4, 0);
    Vector uprime = (target - Cprime).get_normalized();
    Ray r(Cprime, uprime);
    return s->scene.getColor(r, 0, false);
}

int trf_sample_radiance(trf_scene *s, const rt_camera *cam, const rt_params *p, const int32_t *ijk, int64_t n, double *out) {
    if (p->flags != 0 || p->max_rebond != 10 || p->ior != 1.4 || p->epsilon != 0.00001 || p->shadow_bias != 0.0001) return -1;
    for (int64_t q = 0; q < n; q++) {
        Vector c = sample_one(s, cam, p, ijk[3 * q], ijk[3 * q + 1], ijk[3 * q + 2]);
        out[3 * q] = c[0]; out[3 * q + 1] = c[1]; out[3 * q + 2] = c[2];
    }
    return 0;
}

/* The pixel loop as the reference runs it: rows in parallel, schedule(dynamic, 1) (raytracer.cpp:1456-1457), every thread
 * drawing from whatever random source this build has. With TRF_PLAIN that is the reference's single shared engine. */
int trf_render_omp(trf_scene *s, const rt_camera *cam, const rt_params *p, double *accum, int n_threads) {
    if (p->flags != 0 || p->max_rebond != 10 || p->ior != 1.4 || p->epsilon != 0.00001 || p->shadow_bias != 0.0001) return -1;
#ifndef TRF_PLAIN
    if (n_threads != 1) return -2; /* the injected stream is one global cursor: single-threaded only */
#endif
    int j0 = p->rect[0], i0 = p->rect[1], j1 = p->rect[2], i1 = p->rect[3];
    if (!j0 && !i0 && !j1 && !i1) { j1 = p->W; i1 = p->H; }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int i = i0; i < i1; i++)
        for (int j = j0; j < j1; j++) {
            double *px = accum + ((int64_t)i * p->W + j) * 3;
            Vector color(px[0], px[1], px[2]);
            for (int k = 0; k < p->spp; k++) color += sample_one(s, cam, p, i, j, p->spp_offset + k);
            px[0] = color[0]; px[1] = color[1]; px[2] = color[2];
        }
    return 0;
}

int trf_render(trf_scene *s, const rt_camera *cam, const rt_params *p, double *accum) {
    if (p->flags != 0 || p->max_rebond != 10 || p->ior != 1.4 || p->epsilon != 0.00001 || p->shadow_bias != 0.0001) return -1;
    int j0 = p->rect[0], i0 = p->rect[1], j1 = p->rect[2], i1 = p->rect[3];
    if (!j0 && !i0 && !j1 && !i1) { j1 = p->W; i1 = p->H; }
    for (int i = i0; i < i1; i++)
        for (int j = j0; j < j1; j++) {
            double *px = accum + ((int64_t)i * p->W + j) * 3;
            Vector color(px[0], px[1], px[2]);
            for (int k = 0; k < p->spp; k++) color += sample_one(s, cam, p, i, j, p->spp_offset + k);
            px[0] = color[0]; px[1] = color[1]; px[2] = color[2];
        }
    return 0;
}

/* raytracer.cpp:1488-1501 */
void trf_resolve(const double *accum, int W, int H, int nbrays, uint8_t *rgb8) {
    std::vector<unsigned char> image(W * H * 3, 0);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            Vector color(accum[((int64_t)i * W + j) * 3], accum[((int64_t)i * W + j) * 3 + 1], accum[((int64_t)i * W + j) * 3 + 2]);
            color = color / nbrays;
            image[((H - i - 1) * W + j) * 3 + 0] = std::min(255., std::pow(color[0], 0.45));
            image[((H - i - 1) * W + j) * 3 + 1] = std::min(255., std::pow(color[1], 0.45));
            image[((H - i - 1) * W + j) * 3 + 2] = std::min(255., std::pow(color[2], 0.45));
        }
    memcpy(rgb8, &image[0], image.size());
}

/* camera basis exactly as main() builds it, raytracer.cpp:1441-1453 */
void trf_camera_basis(double fov_deg, double angle_v_deg, double angle_h_deg, int W, double *right3, double *up3, double *view3, double *zdist) {
    double fov = fov_deg * M_PI / 180;
    double angleVertical = angle_v_deg * M_PI / 180, angleHorizontal = angle_h_deg * M_PI / 180;
    Vector up(0, cos(angleVertical), sin(angleVertical));
    Vector right(cos(angleHorizontal), 0, sin(angleHorizontal));
    double up0 = up[0];
    up[0] = cos(angleHorizontal) * up[0] - sin(angleHorizontal) * up[2];
    up[2] = sin(angleHorizontal) * up0 + cos(angleHorizontal) * up[2];
    Vector viewDirection = cross(up, right);
    for (int k = 0; k < 3; k++) { right3[k] = right[k]; up3[k] = up[k]; view3[k] = viewDirection[k]; }
    *zdist = W / (2. * tan(fov / 2));
}

uint64_t trf_draws(void) { return g_draws; }

void trf_scene_destroy(trf_scene *s) {
    if (!s) return;
    for (size_t i = 0; i < s->owned.size(); i++) {
        /* Object has no virtual destructor: delete through the concrete type */
        if (TriangleMesh *m = dynamic_cast<TriangleMesh *>(s->owned[i])) delete m;
        else delete dynamic_cast<Sphere *>(s->owned[i]);
    }
    delete s;
}

} /* extern "C" */
