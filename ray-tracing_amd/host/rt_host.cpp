/*
 * rt_host.cpp - host C++ API: scene objects, OBJ/texture loading, mesh transforms, BVH build,
 * flattening into the C-ABI arrays, and the Renderer that calls the C ABI.
 * Reference behaviour followed per function (file:line = /root/reference/raytracer.cpp).
 */
#include "rt_host.hpp"
#include "rt_threads.hpp"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <thread>

namespace rt {

static const double kPi = 3.14159265358979323846;

/* ---- Vector algebra, raytracer.cpp:29-110 ---------------------------------------------------- */
Vector Vector::get_normalized() const {
    double n = sqrt(sqrNorm());
    return Vector(c[0] / n, c[1] / n, c[2] / n);
}
Vector operator+(const Vector &a, const Vector &b) { return Vector(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
Vector operator-(const Vector &a, const Vector &b) { return Vector(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
Vector operator-(const Vector &a) { return Vector(-a[0], -a[1], -a[2]); }
Vector operator*(double a, const Vector &b) { return Vector(a * b[0], a * b[1], a * b[2]); }
Vector operator*(const Vector &a, double b) { return Vector(a[0] * b, a[1] * b, a[2] * b); }
Vector operator*(const Vector &a, const Vector &b) { return Vector(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
Vector operator/(const Vector &a, double b) { return Vector(a[0] / b, a[1] / b, a[2] / b); }
Vector cross(const Vector &a, const Vector &b) {
    return Vector(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
double dot(const Vector &a, const Vector &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* ---- OBJ reader: same results as raytracer.cpp:293-640 on well-formed files ------------------ */
namespace {
struct FaceRef {
    int v, vt, vn;
    bool has_vt, has_vn;
};
/* one "a", "a/b", "a/b/c" or "a//c" reference starting at *pp; advances *pp */
bool parse_ref(const char **pp, FaceRef &r) {
    const char *p = *pp;
    while (*p == ' ' || *p == '\t' || *p == '\r') p++;
    char *end;
    long a = strtol(p, &end, 10);
    if (end == p) return false;
    r.v = (int)a;
    r.vt = r.vn = 0;
    r.has_vt = r.has_vn = false;
    p = end;
    if (*p == '/') {
        p++;
        if (*p != '/') {
            long b = strtol(p, &end, 10);
            if (end != p) { r.vt = (int)b; r.has_vt = true; p = end; }
        }
        if (*p == '/') {
            p++;
            long c = strtol(p, &end, 10);
            if (end != p) { r.vn = (int)c; r.has_vn = true; p = end; }
        }
    }
    *pp = p;
    return true;
}
/* 1-based, or negative = relative to the current end (raytracer.cpp:365-368) */
inline int resolve(int idx, size_t count) { return idx < 0 ? (int)count + idx : idx - 1; }
} // namespace

void TriangleMesh::readOBJ(const char *obj) {
    FILE *f = fopen(obj, "r");
    if (!f) throw std::runtime_error(std::string("readOBJ: cannot open ") + obj);
    int curGroup = -1;
    char line[255]; /* the reference reads 254-character pieces (:304-305); longer lines split the same way */
    while (fgets(line, 255, f)) {
        if (line[0] == 'u' && line[1] == 's') curGroup++; /* usemtl, :312-316 */
        if (line[0] == 'v' && line[1] == ' ') {           /* :318-337 */
            double x = 0, y = 0, z = 0, r = 0, g = 0, b = 0;
            int got = sscanf(line, "v %lf %lf %lf %lf %lf %lf\n", &x, &y, &z, &r, &g, &b);
            vertices.push_back(Vector(x, y, z));
            if (got == 6) {
                r = r < 0 ? 0 : (r > 1 ? 1 : r);
                g = g < 0 ? 0 : (g > 1 ? 1 : g);
                b = b < 0 ? 0 : (b > 1 ? 1 : b);
                vertexcolors.push_back(Vector(r, g, b));
            }
        }
        if (line[0] == 'v' && line[1] == 'n') { /* :338-343 */
            double x = 0, y = 0, z = 0;
            sscanf(line, "vn %lf %lf %lf\n", &x, &y, &z);
            normals.push_back(Vector(x, y, z));
        }
        if (line[0] == 'v' && line[1] == 't') { /* :344-349: two values, third stays 0 */
            double x = 0, y = 0;
            sscanf(line, "vt %lf %lf\n", &x, &y);
            uvs.push_back(Vector(x, y, 0));
        }
        if (line[0] == 'f') { /* :350-637: first three references, then a fan (0, previous, next) */
            const char *p = line + 1;
            FaceRef r0, r1, r2;
            if (!parse_ref(&p, r0) || !parse_ref(&p, r1) || !parse_ref(&p, r2)) continue;
            FaceRef prev = r2;
            FaceRef cur = r2;
            bool first = true;
            while (true) {
                if (!first) {
                    /* skip anything that is not a reference, one character at a time (:631) */
                    bool got = false;
                    while (*p != '\n' && *p != '\0') {
                        const char *q = p;
                        if (parse_ref(&q, cur)) { p = q; got = true; break; }
                        p++;
                    }
                    if (!got) break;
                }
                const FaceRef &a = r0, &b = first ? r1 : prev, &c = cur;
                TriangleIndices t;
                t.group = curGroup;
                t.vtxi = resolve(a.v, vertices.size());
                t.vtxj = resolve(b.v, vertices.size());
                t.vtxk = resolve(c.v, vertices.size());
                if (c.has_vt && (first ? (a.has_vt && b.has_vt) : true)) {
                    t.uvi = resolve(a.vt, uvs.size());
                    t.uvj = resolve(b.vt, uvs.size());
                    t.uvk = resolve(c.vt, uvs.size());
                }
                if (c.has_vn && (first ? (a.has_vn && b.has_vn) : true)) {
                    t.ni = resolve(a.vn, normals.size());
                    t.nj = resolve(b.vn, normals.size());
                    t.nk = resolve(c.vn, normals.size());
                }
                indices.push_back(t);
                prev = cur;
                first = false;
            }
        }
    }
    fclose(f);
}

void TriangleMesh::addTexture(const uint8_t *rgb, int W, int H) {
    if (!rgb || W <= 0 || H <= 0) throw std::runtime_error("addTexture: bad texture");
    textures.push_back(std::vector<uint8_t>(rgb, rgb + (size_t)W * H * 3));
    Wtex.push_back(W);
    Htex.push_back(H);
}

void TriangleMesh::loadTexture(const char *filename) {
    FILE *f = fopen(filename, "rb");
    if (!f) throw std::runtime_error(std::string("loadTexture: cannot open ") + filename);
    {
        /* JPEG, what the reference's scenes use (raytracer.cpp:1223, :1244, ...): decoded with the arithmetic of the
         * reference's stbi_load(..., 3) call (:762), texel for texel (rt_jpeg.cpp) */
        uint8_t head[3] = {0, 0, 0};
        const size_t got = fread(head, 1, 3, f);
        if (looksLikeJPEG(head, got)) {
            std::vector<uint8_t> file;
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            bool ok = sz > 0;
            if (ok) {
                file.resize((size_t)sz);
                ok = fread(file.data(), 1, file.size(), f) == file.size();
            }
            fclose(f);
            if (!ok) throw std::runtime_error(std::string("loadTexture: cannot read ") + filename);
            std::vector<uint8_t> rgb;
            int W = 0, H = 0;
            try {
                decodeJPEG(file.data(), file.size(), rgb, W, H);
            } catch (const std::exception &e) {
                throw std::runtime_error(std::string("loadTexture: ") + filename + ": " + e.what());
            }
            addTexture(rgb.data(), W, H);
            return;
        }
        fseek(f, 0, SEEK_SET);
    }
    int W = 0, H = 0, maxv = 0;
    char magic[3] = {0, 0, 0};
    bool ok = fscanf(f, "%2s", magic) == 1 && magic[0] == 'P' && magic[1] == '6';
    /* header fields may be separated by comments */
    int *fields[3] = {&W, &H, &maxv};
    for (int k = 0; ok && k < 3; k++) {
        int ch;
        while ((ch = fgetc(f)) != EOF) {
            if (ch == '#') { while ((ch = fgetc(f)) != EOF && ch != '\n') {} continue; }
            if (ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r') { ungetc(ch, f); break; }
        }
        ok = fscanf(f, "%d", fields[k]) == 1;
    }
    if (ok) fgetc(f); /* the single whitespace before the raster */
    std::vector<uint8_t> buf;
    if (ok && W > 0 && H > 0 && maxv == 255) {
        buf.resize((size_t)W * H * 3);
        ok = fread(buf.data(), 1, buf.size(), f) == buf.size();
    } else {
        ok = false;
    }
    fclose(f);
    if (!ok) throw std::runtime_error(std::string("loadTexture: neither a JPEG file nor a binary PPM (P6, maxval 255): ") + filename +
                                      " - decode other formats to RGB8 and use addTexture");
    addTexture(buf.data(), W, H);
}

/* ---- transforms, raytracer.cpp:768-860. The reference loops run one element past the end
 * (i <= vertices.size()); that element does not exist here. Everything else is kept, including
 * the integer arguments and the integer truncation inside rotateAroundAxe and zoom. ------------ */
void TriangleMesh::invertNormals() { /* :768-774: as many normals as there are vertices */
    size_t n = vertices.size() < normals.size() ? vertices.size() : normals.size();
    for (size_t i = 0; i < n; i++) normals[i] = -normals[i];
}
void TriangleMesh::translate(int dir, int dist) { /* :776-794: dist == 0 mirrors the axis */
    if (dist != 0) for (size_t i = 0; i < vertices.size(); i++) vertices[i][dir] += dist;
    else for (size_t i = 0; i < vertices.size(); i++) vertices[i][dir] = -vertices[i][dir];
}
void TriangleMesh::rotateNormals(int axe1, int axe2) { /* :795-801 */
    size_t n = vertices.size() < normals.size() ? vertices.size() : normals.size();
    for (size_t i = 0; i < n; i++) { double tmp = normals[i][axe1]; normals[i][axe1] = normals[i][axe2]; normals[i][axe2] = tmp; }
}
void TriangleMesh::rotate(int axe1, int axe2) { /* :802-808 */
    for (size_t i = 0; i < vertices.size(); i++) { double tmp = vertices[i][axe1]; vertices[i][axe1] = vertices[i][axe2]; vertices[i][axe2] = tmp; }
}
void TriangleMesh::rotateAroundAxe(int axe, double angle) { /* :810-838 */
    int axe1, axe2;
    if (axe == 2) { axe1 = 0; axe2 = 1; }
    else if (axe == 1) { axe1 = 0; axe2 = 2; }
    else { axe1 = 1; axe2 = 2; }
    for (size_t i = 0; i < vertices.size(); i++) {
        int axe1value = (int)vertices[i][axe1]; /* :831-834: the first coordinate is truncated to int */
        double v2 = vertices[i][axe2];
        vertices[i][axe1] = cos(angle * kPi / 180) * axe1value - sin(angle * kPi / 180) * v2;
        vertices[i][axe2] = sin(angle * kPi / 180) * axe1value + cos(angle * kPi / 180) * v2;
    }
}
void TriangleMesh::zoom(int zoomFacteur) {
    /* :840-860. The reference sums the coordinates into ints, divides each sum by itself (an
     * optimising compiler folds x / x to 1 whatever x is, zero included) and uses (1,1,1) as the
     * "centre": every vertex becomes v + zoomFacteur * (1 * v). The argument is an int, so
     * zoom(0.8) is zoom(0) (:1285). */
    const Vector center(1, 1, 1);
    for (size_t i = 0; i < vertices.size(); i++) {
        Vector cv = center * vertices[i];
        vertices[i][0] = vertices[i][0] + zoomFacteur * cv[0];
        vertices[i][1] = vertices[i][1] + zoomFacteur * cv[1];
        vertices[i][2] = vertices[i][2] + zoomFacteur * cv[2];
    }
}

/* ---- BVH, raytracer.cpp:220-291 --------------------------------------------------------------- */
namespace {
inline double smin(double a, double b) { return (b < a) ? b : a; } /* std::min / std::max */
inline double smax(double a, double b) { return (a < b) ? b : a; }
} // namespace

void TriangleMesh::buildNode(int node, int debut, int fin, int depth) {
    { /* buildBB, :220-242 */
        BVHNode &n = bvh[node];
        n.debut = debut; n.fin = fin; n.fg = n.fd = -1; n.depth = depth;
        for (int k = 0; k < 3; k++) { n.box[k] = 1E9; n.box[3 + k] = -1E9; }
        for (int i = debut; i < fin; i++) {
            const int vid[3] = {indices[i].vtxi, indices[i].vtxj, indices[i].vtxk};
            for (int k = 0; k < 3; k++)
                for (int q = 0; q < 3; q++) {
                    n.box[k] = smin(n.box[k], vertices[vid[q]][k]);
                    n.box[3 + k] = smax(n.box[3 + k], vertices[vid[q]][k]);
                }
        }
    }
    const double dx = bvh[node].box[3] - bvh[node].box[0], dy = bvh[node].box[4] - bvh[node].box[1], dz = bvh[node].box[5] - bvh[node].box[2];
    int dim;
    if (dx >= dy && dx >= dz) dim = 0;
    else if (dy >= dx && dy >= dz) dim = 1;
    else dim = 2;
    const double milieu = (bvh[node].box[dim] + bvh[node].box[3 + dim]) * 0.5;
    int pivot = debut;
    for (int i = debut; i < fin; i++) { /* :270-278 */
        double centre = (vertices[indices[i].vtxi][dim] + vertices[indices[i].vtxj][dim] + vertices[indices[i].vtxk][dim]) / 3;
        if (centre < milieu) {
            std::swap(indices[i], indices[pivot]);
            std::swap(order[i], order[pivot]);
            pivot++;
        }
    }
    if (pivot == debut || pivot == fin || fin - debut < 5) return; /* :283 */
    const int l = (int)bvh.size();
    bvh.push_back(BVHNode());
    const int r = (int)bvh.size();
    bvh.push_back(BVHNode());
    bvh[node].fg = l;
    bvh[node].fd = r;
    buildNode(l, debut, pivot, depth + 1);
    buildNode(r, pivot, fin, depth + 1);
}

void TriangleMesh::buildBVH() {
    for (size_t i = 0; i < indices.size(); i++) {
        const TriangleIndices &t = indices[i];
        const int nv = (int)vertices.size();
        if (t.vtxi < 0 || t.vtxi >= nv || t.vtxj < 0 || t.vtxj >= nv || t.vtxk < 0 || t.vtxk >= nv)
            throw std::runtime_error("buildBVH: a triangle refers to a vertex that does not exist");
    }
    bvh.clear();
    order.resize(indices.size());
    for (size_t i = 0; i < indices.size(); i++) order[i] = (int)i;
    bvh.reserve(2 * indices.size() + 1);
    bvh.push_back(BVHNode());
    buildNode(0, 0, (int)indices.size(), 0);
}

void TriangleMesh::buildBVH(rt_ctx *ctx) {
    if (!ctx) throw std::runtime_error("buildBVH: no context");
    const size_t nt = indices.size(), nv = vertices.size();
    if (nt == 0) throw std::runtime_error("buildBVH: the mesh has no triangles");
    std::vector<double> v(nv * 3);
    for (size_t i = 0; i < nv; i++)
        for (int k = 0; k < 3; k++) v[3 * i + k] = vertices[i][k];
    std::vector<int32_t> tv(nt * 3), ord(nt);
    for (size_t i = 0; i < nt; i++) { tv[3 * i] = indices[i].vtxi; tv[3 * i + 1] = indices[i].vtxj; tv[3 * i + 2] = indices[i].vtxk; }
    const size_t cap = 2 * nt + 1;
    std::unique_ptr<rt_build_node[]> nodes(new rt_build_node[cap]); /* not zero-filled: 72 bytes per possible node */
    int64_t n_nodes = 0;
    if (rt_build_bvh(ctx, v.data(), (int64_t)nv, tv.data(), (int64_t)nt, ord.data(), nodes.get(), (int64_t)cap, &n_nodes) != RT_OK)
        throw std::runtime_error(std::string("buildBVH: ") + rt_last_error(ctx));
    std::vector<TriangleIndices> moved(nt);
    order.resize(nt);
    for (size_t k = 0; k < nt; k++) { moved[k] = indices[(size_t)ord[k]]; order[k] = ord[k]; }
    indices.swap(moved);
    static_assert(sizeof(BVHNode) == sizeof(rt_build_node) && offsetof(BVHNode, debut) == offsetof(rt_build_node, debut) &&
                      offsetof(BVHNode, fg) == offsetof(rt_build_node, fg) && offsetof(BVHNode, depth) == offsetof(rt_build_node, depth),
                  "the two node records have the same layout");
    const BVHNode *first = reinterpret_cast<const BVHNode *>(nodes.get());
    bvh.assign(first, first + n_nodes);
}

int TriangleMesh::maxDepth() const {
    int d = 0;
    for (size_t i = 0; i < bvh.size(); i++) if (bvh[i].depth > d) d = bvh[i].depth;
    return d;
}

/* ---- camera / params ------------------------------------------------------------------------- */
rt_camera Camera::pod(int W) const { /* raytracer.cpp:1446-1453 */
    Vector up(0, cos(angleVertical), sin(angleVertical));
    Vector right(cos(angleHorizontal), 0, sin(angleHorizontal));
    double up0 = up[0];
    up[0] = cos(angleHorizontal) * up[0] - sin(angleHorizontal) * up[2];
    up[2] = sin(angleHorizontal) * up0 + cos(angleHorizontal) * up[2];
    Vector view = cross(up, right);
    rt_camera c;
    for (int k = 0; k < 3; k++) { c.C[k] = C[k]; c.right[k] = right[k]; c.up[k] = up[k]; c.view[k] = view[k]; }
    c.zdist = W / (2. * tan(fov / 2)); /* :1475 */
    c.focus = focus;
    c.aa_sigma = aa_sigma;
    c.dof_sigma = dof_sigma;
    return c;
}

rt_params RenderParams::pod() const {
    rt_params p;
    memset(&p, 0, sizeof p);
    p.W = W; p.H = H; p.spp = nbrays; p.spp_offset = spp_offset; p.seed = seed; p.flags = flags;
    p.max_rebond = 10;      /* :962 */
    p.ior = 1.4;            /* :984 */
    p.epsilon = 0.00001;    /* :955 */
    p.shadow_bias = 0.0001; /* :1073 */
    return p;
}

/* ---- flattening ------------------------------------------------------------------------------ */
rt_scene_desc FlatScene::desc() const {
    rt_scene_desc d;
    memset(&d, 0, sizeof d);
    d.n_spheres = (int32_t)spheres.size(); d.spheres = spheres.data();
    d.n_meshes = (int32_t)meshes.size(); d.meshes = meshes.data();
    d.n_nodes = (int64_t)nodes.size(); d.nodes = nodes.data();
    d.n_tris = (int64_t)tris.size(); d.tris = tris.data(); d.attrs = attrs.data();
    d.n_tex = (int32_t)tex.size(); d.tex = tex.data();
    d.texels = texels.data(); d.texel_bytes = (int64_t)texels.size();
    d.srgb_lut = lut;
    d.I = I;
    for (int k = 0; k < 3; k++) d.L[k] = L[k];
    return d;
}

size_t FlatScene::bytes() const {
    return spheres.size() * sizeof(rt_sphere) + meshes.size() * sizeof(rt_mesh) + nodes.size() * sizeof(rt_bvh_node) +
           tris.size() * (sizeof(rt_tri) + sizeof(rt_tri_attr)) + tex.size() * sizeof(rt_texture) + texels.size() + sizeof lut;
}

void flatten(const Scene &scene, FlatScene &out) {
    out = FlatScene();
    rtown::stage_tick(nullptr);
    rtown::PoolScope pool(rtown::host_threads()); /* the caller's, when it has one already */
    {
        size_t nt = 0, ntexel = 0;
        for (const Object *o : scene.objects)
            if (const TriangleMesh *m = dynamic_cast<const TriangleMesh *>(o)) {
                nt += m->indices.size();
                for (const auto &t : m->textures) ntexel += t.size();
            }
        out.tris.reserve(nt);
        out.attrs.reserve(nt);
        out.texels.reserve(ntexel);
    }
    out.I = scene.I;
    for (int k = 0; k < 3; k++) out.L[k] = scene.L[k];
    for (int i = 0; i < 256; i++) out.lut[i] = pow(i / 255., 2.2); /* :713-715 */
    for (size_t oi = 0; oi < scene.objects.size(); oi++) {
        const Object *o = scene.objects[oi];
        if (!o) throw std::runtime_error("flatten: NULL object");
        const int flags = (o->isMirror ? RT_MAT_MIRROR : 0) | (o->isTransparent ? RT_MAT_TRANSPARENT : 0);
        if (const Sphere *s = dynamic_cast<const Sphere *>(o)) {
            rt_sphere r;
            for (int k = 0; k < 3; k++) { r.O[k] = s->O[k]; r.albedo[k] = s->albedo[k]; }
            r.R = s->R;
            r.flags = flags;
            r.object_id = (int32_t)oi;
            out.spheres.push_back(r);
            continue;
        }
        const TriangleMesh *m = dynamic_cast<const TriangleMesh *>(o);
        if (!m) throw std::runtime_error("flatten: object is neither a Sphere nor a TriangleMesh");
        if (m->bvh.empty() || m->order.size() != m->indices.size()) throw std::runtime_error("flatten: mesh without BVH (call buildBVH after the last transform)");
        if (m->indices.empty()) throw std::runtime_error("flatten: mesh without triangles");
        rt_mesh rm;
        memset(&rm, 0, sizeof rm);
        for (int k = 0; k < 6; k++) rm.root_box[k] = m->bvh[0].box[k];
        for (int k = 0; k < 3; k++) rm.albedo[k] = m->albedo[k];
        rm.flags = flags;
        rm.object_id = (int32_t)oi;
        rm.first_tri = (int32_t)out.tris.size();
        rm.n_tris = (int32_t)m->indices.size();
        rm.first_tex = (int32_t)out.tex.size();
        rm.n_tex = (int32_t)m->textures.size();
        rm.max_depth = m->maxDepth();
        /* Triangle order in HBM = the order in which the reference's traversal would VISIT them if it
         * visited everything: it pops fd before fg (raytracer.cpp:656-665), so leaves come in descending
         * `debut`, and a leaf is scanned in ascending i (:670). With this layout "the first visited of
         * equal t wins" (strict '<' at :690) is "the lower position wins", whatever order the GPU
         * traverses in. */
        std::vector<int> leaf_new_first(m->indices.size() + 1, -1);
        std::vector<int> visit; /* visit[new position - first_tri] = position in m->indices */
        visit.reserve(m->indices.size());
        {
            /* the leaves by descending `debut`: their ranges tile [0, n), so a table over debut orders them in one pass
             * (two leaves with one debut - an empty range - fall back to sorting) */
            std::vector<int> leaves, leaf_at(m->indices.size() + 1, -1);
            bool tiled = true;
            for (size_t k = 0; k < m->bvh.size(); k++)
                if (m->bvh[k].fg < 0) {
                    const int d = m->bvh[k].debut;
                    if (d < 0 || (size_t)d > m->indices.size() || leaf_at[(size_t)d] >= 0) { tiled = false; break; }
                    leaf_at[(size_t)d] = (int)k;
                }
            if (tiled) {
                for (size_t d = leaf_at.size(); d-- > 0;)
                    if (leaf_at[d] >= 0) leaves.push_back(leaf_at[d]);
            } else {
                for (size_t k = 0; k < m->bvh.size(); k++) if (m->bvh[k].fg < 0) leaves.push_back((int)k);
                std::sort(leaves.begin(), leaves.end(), [&](int x, int y) { return m->bvh[x].debut > m->bvh[y].debut; });
            }
            for (size_t q = 0; q < leaves.size(); q++) {
                const BVHNode &lf = m->bvh[(size_t)leaves[q]];
                leaf_new_first[(size_t)lf.debut] = rm.first_tri + (int)visit.size();
                for (int i = lf.debut; i < lf.fin; i++) visit.push_back(i);
            }
        }
        if (visit.size() != m->indices.size()) throw std::runtime_error("flatten: BVH leaves do not cover the triangles");
        rtown::stage_tick("  flatten: visit order");
        /* A, e1 = B - A, e2 = C - A (:673-677) and the gathered attributes */
        const int nn = (int)m->normals.size(), nuv = (int)m->uvs.size();
        const int32_t mesh_index = (int32_t)out.meshes.size();
        const size_t base = out.tris.size();
        out.tris.resize(base + visit.size());
        out.attrs.resize(base + visit.size());
        /* the triangles are independent: large meshes are filled by several threads; the first error of each is kept */
        const int threads = visit.size() >= (1u << 12) ? rtown::host_threads() : 1;
        std::vector<const char *> err((size_t)threads, nullptr);
        auto fill = [&](int part, size_t vk0, size_t vk1) {
            for (size_t vk = vk0; vk < vk1; vk++) {
                const size_t k = (size_t)visit[vk];
                const TriangleIndices &t = m->indices[k];
                const Vector &A = m->vertices[t.vtxi], &B = m->vertices[t.vtxj], &C = m->vertices[t.vtxk];
                Vector e1 = B - A, e2 = C - A;
                rt_tri &rt_ = out.tris[base + vk];
                for (int q = 0; q < 3; q++) { rt_.A[q] = A[q]; rt_.e1[q] = e1[q]; rt_.e2[q] = e2[q]; }
                rt_.orig_id = m->order[k];
                rt_.mesh = mesh_index;
                rt_tri_attr &at = out.attrs[base + vk];
                memset(&at, 0, sizeof at);
                const int ni[3] = {t.ni, t.nj, t.nk}, ui[3] = {t.uvi, t.uvj, t.uvk};
                for (int q = 0; q < 3; q++) {
                    /* the reference reads normals[ni] unchecked (:694); a missing normal is an error here */
                    if (ni[q] < 0 || ni[q] >= nn) { err[(size_t)part] = "flatten: a triangle has no vertex normal (the reference interpolates normals[ni], raytracer.cpp:694)"; return; }
                    for (int d = 0; d < 3; d++) at.n[q][d] = m->normals[ni[q]][d];
                    if (rm.n_tex > 0) {
                        if (ui[q] < 0 || ui[q] >= nuv) { err[(size_t)part] = "flatten: a textured triangle has no uv (raytracer.cpp:702)"; return; }
                        at.uv[q][0] = m->uvs[ui[q]][0];
                        at.uv[q][1] = m->uvs[ui[q]][1];
                    }
                }
                at.group = t.group;
                if (rm.n_tex > 0 && (t.group < 0 || t.group >= rm.n_tex)) {
                    err[(size_t)part] = "flatten: triangle texture group outside the mesh's textures (raytracer.cpp:700-701 would read out of bounds)";
                    return;
                }
            }
        };
        rtown::parallel_parts((int64_t)visit.size(), threads, [&](int part, int64_t vk0, int64_t vk1) { fill(part, (size_t)vk0, (size_t)vk1); });
        for (const char *e : err)
            if (e) throw std::runtime_error(e);
        rtown::stage_tick("  flatten: triangles");
        /* nodes: one record per inner Noeud holding both child boxes; children get larger indices */
        if (m->bvh[0].fg < 0) {
            rm.root = leaf_new_first[(size_t)m->bvh[0].debut];
            rm.root_count = m->bvh[0].fin - m->bvh[0].debut;
        } else {
            std::vector<int> rec_of(m->bvh.size(), -1);
            int next = (int)out.nodes.size();
            for (size_t k = 0; k < m->bvh.size(); k++) /* bvh is in creation order: parents before children */
                if (m->bvh[k].fg >= 0) rec_of[k] = next++;
            out.nodes.resize((size_t)next);
            rtown::parallel_parts((int64_t)m->bvh.size(), m->bvh.size() >= (1u << 14) ? rtown::host_threads() : 1, [&](int, int64_t k0, int64_t k1) {
                for (int64_t k = k0; k < k1; k++) {
                    const BVHNode &n = m->bvh[(size_t)k];
                    if (n.fg < 0) continue;
                    rt_bvh_node &r = out.nodes[(size_t)rec_of[(size_t)k]];
                    memset(&r, 0, sizeof r);
                    const BVHNode &a = m->bvh[n.fg], &b = m->bvh[n.fd];
                    for (int q = 0; q < 6; q++) { r.boxL[q] = a.box[q]; r.boxR[q] = b.box[q]; }
                    if (a.fg >= 0) { r.L = rec_of[n.fg]; r.nL = 0; } else { r.L = leaf_new_first[(size_t)a.debut]; r.nL = a.fin - a.debut; }
                    if (b.fg >= 0) { r.R = rec_of[n.fd]; r.nR = 0; } else { r.R = leaf_new_first[(size_t)b.debut]; r.nR = b.fin - b.debut; }
                }
            });
            rm.root = rec_of[0];
            rm.root_count = 0;
        }
        for (size_t k = 0; k < m->textures.size(); k++) {
            rt_texture tx;
            tx.offset = (int64_t)out.texels.size();
            tx.W = m->Wtex[k];
            tx.H = m->Htex[k];
            out.tex.push_back(tx);
            const std::vector<uint8_t> &src = m->textures[k];
            out.texels.resize((size_t)tx.offset + src.size()); /* reserved above: no reallocation */
            rtown::parallel_parts((int64_t)src.size(), src.size() >= (1u << 20) ? rtown::host_threads() : 1, [&](int, int64_t b0, int64_t b1) {
                memcpy(out.texels.data() + tx.offset + b0, src.data() + b0, (size_t)(b1 - b0));
            });
        }
        out.meshes.push_back(rm);
        rtown::stage_tick("  flatten: records, textures");
    }
}

/* ---- Renderer -------------------------------------------------------------------------------- */
void Renderer::check(int rc, const char *what) {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx_));
}
Renderer::Renderer(int device) {
    int rc = rt_create(device, &ctx_);
    if (rc != RT_OK) throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
}
Renderer::~Renderer() { rt_destroy(ctx_); }
void Renderer::setScene(const Scene &scene) {
    if (scene.objects.empty() || !dynamic_cast<const Sphere *>(scene.objects[0]))
        throw std::runtime_error("setScene: objects[0] must be the light Sphere (raytracer.cpp:970)");
    rtown::PoolScope pool(rtown::host_threads()); /* one set of worker threads for flattening and the upload's builds */
    FlatScene fs;
    flatten(scene, fs);
    rt_scene_desc d = fs.desc();
    check(rt_upload_scene(ctx_, &d), "rt_upload_scene");
}
void Renderer::render(const Camera &cam, const RenderParams &p, std::vector<uint8_t> &image) {
    rt_camera c = cam.pod(p.W);
    rt_params rp = p.pod();
    check(rt_accum_clear(ctx_, p.W, p.H), "rt_accum_clear");
    check(rt_render(ctx_, &c, &rp, nullptr), "rt_render");
    image.resize((size_t)p.W * p.H * 3);
    check(rt_resolve(ctx_, nullptr, p.W, p.H, (double)p.nbrays, image.data()), "rt_resolve");
}
void Renderer::intersect(const std::vector<Ray> &rays, std::vector<int> &objectId, std::vector<int> &triangleId, std::vector<double> &t) {
    std::vector<double> flat(rays.size() * 6);
    for (size_t i = 0; i < rays.size(); i++)
        for (int k = 0; k < 3; k++) { flat[6 * i + k] = rays[i].C[k]; flat[6 * i + 3 + k] = rays[i].u[k]; }
    objectId.resize(rays.size());
    triangleId.resize(rays.size());
    t.resize(rays.size());
    check(rt_intersect_rays(ctx_, flat.data(), (int64_t)rays.size(), objectId.data(), triangleId.data(), t.data(), nullptr, nullptr), "rt_intersect_rays");
}
double Renderer::lastRenderMs() {
    double ms = 0;
    check(rt_last_render_ms(ctx_, &ms, nullptr), "rt_last_render_ms");
    return ms;
}
void render(const Scene &scene, const Camera &cam, const RenderParams &p, std::vector<uint8_t> &image, int device) {
    Renderer r(device);
    r.setScene(scene);
    r.render(cam, p, image);
}

void renderDevices(const rt_scene_desc &d, const rt_camera &cam, const rt_params &p, const int *devices, int n, uint8_t *rgb8) {
    if (!devices || n <= 0 || !rgb8) throw std::runtime_error("renderDevices: empty device list or no image buffer");
    std::vector<rt_ctx *> ctx((size_t)n, nullptr);
    std::vector<std::string> err((size_t)n);
    auto work = [&](int k) {
        /* one host thread per device: create, upload, render this device's samples [lo, hi) of every pixel */
        auto bad = [&](const char *what, rt_ctx *x) { err[(size_t)k] = std::string(what) + ": " + rt_last_error(x); };
        if (rt_create(devices[k], &ctx[(size_t)k]) != RT_OK) { bad("rt_create", nullptr); return; }
        rt_ctx *x = ctx[(size_t)k];
        if (rt_upload_scene(x, &d) != RT_OK) { bad("rt_upload_scene", x); return; }
        if (rt_accum_clear(x, p.W, p.H) != RT_OK) { bad("rt_accum_clear", x); return; }
        rt_params rp = p;
        const int64_t lo = (int64_t)p.spp * k / n, hi = (int64_t)p.spp * (k + 1) / n;
        rp.spp = (int32_t)(hi - lo);
        rp.spp_offset = p.spp_offset + (int32_t)lo;
        if (rp.spp > 0 && rt_render(x, &cam, &rp, nullptr) != RT_OK) { bad("rt_render", x); return; }
        if (rt_sync(x) != RT_OK) bad("rt_sync", x);
    };
    std::string first_err;
    {
        /* whatever fails - a device that cannot be opened, a thread that cannot be started - the threads already running
         * are joined and every context that was created is destroyed before the error is reported */
        std::vector<std::thread> th;
        try {
            for (int k = 1; k < n; k++) th.emplace_back(work, k);
            work(0);
        } catch (const std::exception &e) {
            first_err = std::string("renderDevices: ") + e.what();
        }
        for (auto &t : th) t.join();
    }
    for (int k = 0; k < n && first_err.empty(); k++) first_err = err[(size_t)k];
    for (int k = 1; k < n && first_err.empty(); k++) /* fixed order: the same bits from run to run */
        if (rt_accum_add(ctx[0], ctx[(size_t)k]) != RT_OK) first_err = std::string("rt_accum_add: ") + rt_last_error(ctx[0]);
    if (first_err.empty() && rt_resolve(ctx[0], nullptr, p.W, p.H, (double)p.spp, rgb8) != RT_OK) first_err = std::string("rt_resolve: ") + rt_last_error(ctx[0]);
    for (rt_ctx *x : ctx) rt_destroy(x);
    if (!first_err.empty()) throw std::runtime_error(first_err);
}

void render(const Scene &scene, const Camera &cam, const RenderParams &p, std::vector<uint8_t> &image, const std::vector<int> &devices) {
    if (scene.objects.empty() || !dynamic_cast<const Sphere *>(scene.objects[0]))
        throw std::runtime_error("render: objects[0] must be the light Sphere (raytracer.cpp:970)");
    FlatScene fs;
    flatten(scene, fs); /* once: every device gets the same arrays */
    image.resize((size_t)p.W * p.H * 3);
    renderDevices(fs.desc(), cam.pod(p.W), p.pod(), devices.data(), (int)devices.size(), image.data());
}

/* ---- image and timing output (raytracer.cpp:1505-1512) ------------------------------------------------------- */
namespace {
struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const uint8_t *p, size_t n) const {
        for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
        return crc;
    }
};
void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void png_chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
    static const Crc32 crc;
    put32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc.run(0xFFFFFFFFu, &out[at], out.size() - at) ^ 0xFFFFFFFFu);
}
} // namespace

void writePNG(const char *path, int W, int H, const uint8_t *rgb) {
    if (!path || !rgb || W <= 0 || H <= 0 || (int64_t)W * H > (int64_t)1 << 28) throw std::runtime_error("writePNG: bad arguments");
    /* scanlines, each behind a filter byte 0 (none) */
    const size_t row = (size_t)W * 3;
    std::vector<uint8_t> raw;
    raw.reserve((row + 1) * (size_t)H);
    for (int i = 0; i < H; i++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)i * row, rgb + (size_t)(i + 1) * row);
    }
    /* zlib stream of stored deflate blocks (at most 65535 bytes each) + Adler-32 */
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t at = 0; at < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        for (size_t i = at; i < at + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        at += n;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)W); put32(ihdr, (uint32_t)H);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); /* 8 bits, RGB, deflate, adaptive filters, no interlace */
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", std::vector<uint8_t>());
    FILE *f = fopen(path, "wb");
    if (!f) throw std::runtime_error(std::string("writePNG: cannot open ") + path);
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    if (fclose(f) != 0 || !ok) throw std::runtime_error(std::string("writePNG: write failed: ") + path);
}

void writeExecutionTime(const char *path, double seconds) {
    if (!path || !(seconds >= 0)) throw std::runtime_error("writeExecutionTime: bad arguments");
    FILE *f = fopen(path, "w");
    if (!f) throw std::runtime_error(std::string("writeExecutionTime: cannot open ") + path);
    const bool ok = fprintf(f, "%lld", (long long)seconds) > 0; /* (clock() - ini_time) / CLOCKS_PER_SEC is an integer division */
    if (fclose(f) != 0 || !ok) throw std::runtime_error(std::string("writeExecutionTime: write failed: ") + path);
}

} // namespace rt

/* ---------------------------------------------------------------------------------------------
 * C wrappers over the classes above (declared in include/rt_amd_host.h) for non-C++ callers
 * ------------------------------------------------------------------------------------------- */
#include "rt_amd_host.h"

struct rth_mesh {
    rt::TriangleMesh m;
    rth_mesh(const double *a, int mirror, int transp) : m(rt::Vector(a[0], a[1], a[2]), mirror != 0, transp != 0) {}
};
struct rth_scene {
    rt::Scene s;
    std::vector<rt::Sphere *> owned;
    ~rth_scene() { for (size_t i = 0; i < owned.size(); i++) delete owned[i]; }
};

static thread_local std::string g_host_err;
#define RTH_TRY(body)                                        \
    try { body; return 0; }                                  \
    catch (const std::exception &e) { g_host_err = e.what(); return -1; } \
    catch (...) { g_host_err = "unknown error"; return -1; }

extern "C" {

const char *rth_last_error(void) { return g_host_err.c_str(); }

rth_mesh *rth_mesh_create(const double albedo[3], int mirror, int transp) { return new rth_mesh(albedo, mirror, transp); }
void rth_mesh_destroy(rth_mesh *m) { delete m; }
int rth_mesh_read_obj(rth_mesh *m, const char *path) { RTH_TRY(m->m.readOBJ(path)) }
int rth_mesh_load_texture(rth_mesh *m, const char *path) { RTH_TRY(m->m.loadTexture(path)) }
int rth_mesh_add_texture(rth_mesh *m, const uint8_t *rgb, int32_t W, int32_t H) { RTH_TRY(m->m.addTexture(rgb, W, H)) }
int rth_mesh_texture(const rth_mesh *m, int32_t k, int32_t *W, int32_t *H, uint8_t *rgb) {
    RTH_TRY({
        if (!m || k < 0 || (size_t)k >= m->m.textures.size()) throw std::runtime_error("rth_mesh_texture: no such texture");
        if (W) *W = m->m.Wtex[(size_t)k];
        if (H) *H = m->m.Htex[(size_t)k];
        if (rgb) memcpy(rgb, m->m.textures[(size_t)k].data(), m->m.textures[(size_t)k].size());
    })
}
int rth_decode_jpeg(const uint8_t *file, int64_t n, int32_t *W, int32_t *H, uint8_t *rgb, int64_t rgb_cap) {
    RTH_TRY({
        if (!file || n <= 0 || !W || !H) throw std::runtime_error("rth_decode_jpeg: bad argument");
        std::vector<uint8_t> out;
        int w = 0;
        int h = 0;
        rt::decodeJPEG(file, (size_t)n, out, w, h);
        *W = w;
        *H = h;
        if (rgb) {
            if ((int64_t)out.size() > rgb_cap) throw std::runtime_error("rth_decode_jpeg: output buffer too small");
            memcpy(rgb, out.data(), out.size());
        }
    })
}

int rth_mesh_set_raw(rth_mesh *h, int64_t nv, const double *v, int64_t nn, const double *n, int64_t nuv, const double *uv, int64_t nt,
                     const int32_t *idx10) {
    RTH_TRY({
        rt::TriangleMesh &m = h->m;
        m.vertices.clear(); m.normals.clear(); m.uvs.clear(); m.indices.clear(); m.bvh.clear(); m.order.clear();
        m.vertices.reserve(nv); m.normals.reserve(nn); m.uvs.reserve(nuv); m.indices.reserve(nt);
        for (int64_t i = 0; i < nv; i++) m.vertices.push_back(rt::Vector(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
        for (int64_t i = 0; i < nn; i++) m.normals.push_back(rt::Vector(n[3 * i], n[3 * i + 1], n[3 * i + 2]));
        for (int64_t i = 0; i < nuv; i++) m.uvs.push_back(rt::Vector(uv[2 * i], uv[2 * i + 1], 0));
        for (int64_t i = 0; i < nt; i++) {
            const int32_t *q = idx10 + 10 * i;
            rt::TriangleIndices t;
            t.vtxi = q[0]; t.vtxj = q[1]; t.vtxk = q[2]; t.uvi = q[3]; t.uvj = q[4]; t.uvk = q[5];
            t.ni = q[6]; t.nj = q[7]; t.nk = q[8]; t.group = q[9];
            m.indices.push_back(t);
        }
    })
}

int rth_mesh_op(rth_mesh *h, int op, int a, int b, double angle) {
    RTH_TRY({
        switch (op) {
        case 0: h->m.invertNormals(); break;
        case 1: h->m.translate(a, b); break;
        case 2: h->m.rotateNormals(a, b); break;
        case 3: h->m.rotate(a, b); break;
        case 4: h->m.rotateAroundAxe(a, angle); break;
        case 5: h->m.zoom(b); break;
        default: throw std::runtime_error("rth_mesh_op: unknown op");
        }
    })
}

int rth_mesh_build_bvh(rth_mesh *h) { RTH_TRY(h->m.buildBVH()) }
int rth_mesh_build_bvh_device(rth_mesh *h, rt_ctx *ctx) { RTH_TRY(h->m.buildBVH(ctx)) }

int rth_mesh_sizes(const rth_mesh *h, int64_t *nv, int64_t *nn, int64_t *nuv, int64_t *nt, int32_t *ntex, int64_t *n_nodes, int32_t *max_depth) {
    if (nv) *nv = (int64_t)h->m.vertices.size();
    if (nn) *nn = (int64_t)h->m.normals.size();
    if (nuv) *nuv = (int64_t)h->m.uvs.size();
    if (nt) *nt = (int64_t)h->m.indices.size();
    if (ntex) *ntex = (int32_t)h->m.textures.size();
    if (n_nodes) *n_nodes = (int64_t)h->m.bvh.size();
    if (max_depth) *max_depth = h->m.maxDepth();
    return 0;
}

int rth_mesh_get(const rth_mesh *h, double *v, double *n, double *uv, int32_t *idx10, int32_t *perm) {
    const rt::TriangleMesh &m = h->m;
    if (v) for (size_t i = 0; i < m.vertices.size(); i++) for (int k = 0; k < 3; k++) v[3 * i + k] = m.vertices[i][k];
    if (n) for (size_t i = 0; i < m.normals.size(); i++) for (int k = 0; k < 3; k++) n[3 * i + k] = m.normals[i][k];
    if (uv) for (size_t i = 0; i < m.uvs.size(); i++) for (int k = 0; k < 2; k++) uv[2 * i + k] = m.uvs[i][k];
    if (idx10) for (size_t i = 0; i < m.indices.size(); i++) {
        const rt::TriangleIndices &t = m.indices[i];
        int32_t *q = idx10 + 10 * i;
        q[0] = t.vtxi; q[1] = t.vtxj; q[2] = t.vtxk; q[3] = t.uvi; q[4] = t.uvj; q[5] = t.uvk; q[6] = t.ni; q[7] = t.nj; q[8] = t.nk; q[9] = t.group;
    }
    if (perm) for (size_t i = 0; i < m.order.size(); i++) perm[i] = m.order[i];
    return 0;
}

/* preorder (fg before fd): box(6), then debut, fin, is_leaf per node */
int64_t rth_mesh_bvh_dump(const rth_mesh *h, double *boxes, int32_t *dfl, int64_t cap) {
    const rt::TriangleMesh &m = h->m;
    if (m.bvh.empty()) return 0;
    std::vector<int> st;
    st.push_back(0);
    int64_t k = 0;
    while (!st.empty()) {
        int c = st.back();
        st.pop_back();
        const rt::BVHNode &n = m.bvh[(size_t)c];
        if (k < cap) {
            if (boxes) for (int q = 0; q < 6; q++) boxes[6 * k + q] = n.box[q];
            if (dfl) { dfl[3 * k] = n.debut; dfl[3 * k + 1] = n.fin; dfl[3 * k + 2] = n.fg < 0; }
        }
        k++;
        if (n.fg >= 0) { st.push_back(n.fd); st.push_back(n.fg); }
    }
    return k;
}

rth_scene *rth_scene_create(double I, const double L[3]) {
    rth_scene *s = new rth_scene();
    s->s.I = I;
    s->s.L = rt::Vector(L[0], L[1], L[2]);
    return s;
}
void rth_scene_destroy(rth_scene *s) { delete s; }
int rth_scene_add_sphere(rth_scene *s, const double O[3], double R, const double albedo[3], int mirror, int transp) {
    rt::Sphere *sp = new rt::Sphere(rt::Vector(O[0], O[1], O[2]), R, rt::Vector(albedo[0], albedo[1], albedo[2]), mirror != 0, transp != 0);
    s->owned.push_back(sp);
    s->s.objects.push_back(sp);
    return (int)s->s.objects.size() - 1;
}
int rth_scene_add_mesh(rth_scene *s, rth_mesh *m) {
    s->s.objects.push_back(&m->m);
    return (int)s->s.objects.size() - 1;
}
int rth_scene_upload(rth_scene *s, rt_ctx *ctx, int64_t *flat_bytes) {
    try {
        rt::FlatScene fs;
        rtown::PoolScope pool(rtown::host_threads()); /* shared by flattening and the upload's builds */
        rt::flatten(s->s, fs);
        if (flat_bytes) *flat_bytes = (int64_t)fs.bytes();
        rt_scene_desc d = fs.desc();
        int rc = rt_upload_scene(ctx, &d);
        if (rc != RT_OK) { g_host_err = rt_last_error(ctx); return rc; }
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}
/* flattening only (no device): sizes of the arrays rt_upload_scene would receive */
int rth_scene_render_devices(rth_scene *s, const rt_camera *cam, const rt_params *p, const int32_t *devices, int32_t n_devices, uint8_t *rgb8) {
    RTH_TRY({
        if (!s || !cam || !p) throw std::runtime_error("rth_scene_render_devices: NULL argument");
        rt::FlatScene fs;
        rt::flatten(s->s, fs);
        rt::renderDevices(fs.desc(), *cam, *p, devices, n_devices, rgb8);
    })
}

int rth_scene_flat_info(rth_scene *s, int64_t *n_spheres, int64_t *n_meshes, int64_t *n_nodes, int64_t *n_tris, int64_t *bytes) {
    RTH_TRY({
        rt::FlatScene fs;
        rt::flatten(s->s, fs);
        if (n_spheres) *n_spheres = (int64_t)fs.spheres.size();
        if (n_meshes) *n_meshes = (int64_t)fs.meshes.size();
        if (n_nodes) *n_nodes = (int64_t)fs.nodes.size();
        if (n_tris) *n_tris = (int64_t)fs.tris.size();
        if (bytes) *bytes = (int64_t)fs.bytes();
    })
}

int rth_scene_flat_arrays(rth_scene *s, rt_sphere *spheres, rt_mesh *meshes, rt_bvh_node *nodes, rt_tri *tris, rt_tri_attr *attrs, rt_texture *tex,
                          uint8_t *texels, int64_t *n_tex, int64_t *texel_bytes) {
    RTH_TRY({
        rt::FlatScene fs;
        rt::flatten(s->s, fs);
        if (n_tex) *n_tex = (int64_t)fs.tex.size();
        if (texel_bytes) *texel_bytes = (int64_t)fs.texels.size();
        if (spheres) memcpy(spheres, fs.spheres.data(), fs.spheres.size() * sizeof(rt_sphere));
        if (meshes) memcpy(meshes, fs.meshes.data(), fs.meshes.size() * sizeof(rt_mesh));
        if (nodes) memcpy(nodes, fs.nodes.data(), fs.nodes.size() * sizeof(rt_bvh_node));
        if (tris) memcpy(tris, fs.tris.data(), fs.tris.size() * sizeof(rt_tri));
        if (attrs) memcpy(attrs, fs.attrs.data(), fs.attrs.size() * sizeof(rt_tri_attr));
        if (tex) memcpy(tex, fs.tex.data(), fs.tex.size() * sizeof(rt_texture));
        if (texels) memcpy(texels, fs.texels.data(), fs.texels.size());
    })
}

int rth_camera_make(const double C[3], double fov_deg, double angle_v_deg, double angle_h_deg, int32_t W, double focus, double aa_sigma,
                    double dof_sigma, rt_camera *out) {
    rt::Camera c;
    c.C = rt::Vector(C[0], C[1], C[2]);
    c.fov = fov_deg * 3.14159265358979323846 / 180;              /* raytracer.cpp:1441 */
    c.angleVertical = angle_v_deg * 3.14159265358979323846 / 180; /* :1444 */
    c.angleHorizontal = angle_h_deg * 3.14159265358979323846 / 180;
    c.focus = focus;
    c.aa_sigma = aa_sigma;
    c.dof_sigma = dof_sigma;
    *out = c.pod(W);
    return 0;
}

int rth_params_default(int32_t W, int32_t H, int32_t spp, rt_params *out) {
    rt::RenderParams p;
    p.W = W; p.H = H; p.nbrays = spp;
    *out = p.pod();
    return 0;
}

int rth_write_png(const char *path, int32_t W, int32_t H, const uint8_t *rgb) { RTH_TRY(rt::writePNG(path, W, H, rgb)) }
int rth_write_execution_time(const char *path, double seconds) { RTH_TRY(rt::writeExecutionTime(path, seconds)) }

} /* extern "C" */
