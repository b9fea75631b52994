/*
 * rt_host.hpp - host-side C++ surface of the MI355X path tracer.
 *
 * Same names, argument meaning and (deliberately reproduced) quirks as the
 * classes of the reference translation unit, so a scene written against
 * raytracer.cpp:1153-1435 reads the same here:
 *   Vector (:29-110), Ray (:140-148), Object (:150-158), TriangleIndices
 *   (:198-206), TriangleMesh (:208-872), Sphere (:874-921), Scene (:923-929).
 * What differs: Scene::intersect / Scene::getColor and the pixel loop of main
 * (:930-1093, :1441-1503) do not run here; rt::Renderer flattens the scene
 * into the POD arrays of include/rt_amd.h and calls the HIP kernels through
 * that C ABI. Errors the reference leaves undefined (missing file, objects[0]
 * not a sphere, texture group out of range) throw std::runtime_error.
 */
#pragma once
#include <stdint.h>

#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rt_amd.h"

namespace rt {

class Vector {
public:
    explicit Vector(double x = 0, double y = 0, double z = 0) { c[0] = x; c[1] = y; c[2] = z; }
    double operator[](int i) const { return c[i]; }
    double &operator[](int i) { return c[i]; }
    double sqrNorm() const { return c[0] * c[0] + c[1] * c[1] + c[2] * c[2]; }
    Vector get_normalized() const;
    Vector &operator+=(const Vector &a) { c[0] += a[0]; c[1] += a[1]; c[2] += a[2]; return *this; }
private:
    double c[3];
};
Vector operator+(const Vector &a, const Vector &b);
Vector operator-(const Vector &a, const Vector &b);
Vector operator-(const Vector &a);
Vector operator*(double a, const Vector &b);
Vector operator*(const Vector &a, double b);
Vector operator*(const Vector &a, const Vector &b); /* component-wise */
Vector operator/(const Vector &a, double b);
Vector cross(const Vector &a, const Vector &b);
double dot(const Vector &a, const Vector &b);
inline double sqr(double x) { return x * x; }

class Ray {
public:
    Ray(const Vector &C, const Vector &u) : C(C), u(u) {}
    Vector C, u;
};

class Object {
public:
    virtual ~Object() {}
    Vector albedo;
    bool isMirror = false, isTransparent = false;
};

class Sphere : public Object {
public:
    Sphere(const Vector &O, double R, const Vector &albedo, bool isMirror = false, bool isTransparent = false) : O(O), R(R) {
        this->albedo = albedo;
        this->isMirror = isMirror;
        this->isTransparent = isTransparent;
    }
    Vector O;
    double R;
};

class TriangleIndices {
public:
    TriangleIndices(int vtxi = -1, int vtxj = -1, int vtxk = -1, int ni = -1, int nj = -1, int nk = -1, int uvi = -1, int uvj = -1,
                    int uvk = -1, int group = -1)
        : vtxi(vtxi), vtxj(vtxj), vtxk(vtxk), uvi(uvi), uvj(uvj), uvk(uvk), ni(ni), nj(nj), nk(nk), group(group) {}
    int vtxi, vtxj, vtxk;
    int uvi, uvj, uvk;
    int ni, nj, nk;
    int group;
};

/* one node of the tree buildBVH produces, children by index (root = 0, -1 on a leaf) */
struct BVHNode {
    double box[6]; /* min xyz, max xyz */
    int debut, fin;
    int fg, fd;
    int depth;
};

class TriangleMesh : public Object {
public:
    TriangleMesh(const Vector &albedo, bool mirror = false, bool transp = false) {
        this->albedo = albedo;
        isMirror = mirror;
        isTransparent = transp;
    }
    void readOBJ(const char *obj);
    /* binary PPM (P6, maxval 255) only; decoded RGB8 from any other source goes through addTexture */
    void loadTexture(const char *filename);
    void addTexture(const uint8_t *rgb, int W, int H);
    void invertNormals();
    void translate(int dir, int dist);
    void rotateNormals(int axe1, int axe2);
    void rotate(int axe1, int axe2);
    void rotateAroundAxe(int axe, double angle);
    void zoom(int zoomFacteur);
    /* buildBVH(BVH, 0, indices.size()) of the reference: permutes `indices` and fills `bvh` */
    void buildBVH();
    /* the same on the GPU of `ctx` (rt_build_bvh): identical boxes, tree and triangle order */
    void buildBVH(rt_ctx *ctx);

    std::vector<TriangleIndices> indices;
    std::vector<Vector> vertices;
    std::vector<Vector> normals;
    std::vector<Vector> uvs;
    std::vector<Vector> vertexcolors;
    std::vector<std::vector<uint8_t> > textures;
    std::vector<int> Wtex, Htex;

    std::vector<BVHNode> bvh; /* empty until buildBVH */
    std::vector<int> order;   /* order[k] = face-fan index of the triangle now at indices[k] */
    int maxDepth() const;
private:
    void buildNode(int node, int debut, int fin, int depth);
};

class Scene {
public:
    Scene() {}
    std::vector<Object *> objects; /* non-owning, objects[0] = the light sphere */
    Vector L;
    double I = 0;
};

/* the literals of main(), raytracer.cpp:1162, :1441-1445, :1468, :1472, :1479 */
struct Camera {
    Vector C = Vector(0, 30, 55);
    double fov = 60 * 3.14159265358979323846 / 180;
    double angleVertical = -25 * 3.14159265358979323846 / 180;
    double angleHorizontal = 0;
    double focus = 55;
    double aa_sigma = 0.25;
    double dof_sigma = 0.01;
    rt_camera pod(int W) const; /* basis vectors as raytracer.cpp:1446-1453 builds them */
};

struct RenderParams {
    int W = 1024, H = 1024;
    int nbrays = 1;
    uint64_t seed = 10;
    uint32_t flags = 0;
    int spp_offset = 0;
    rt_params pod() const;
};

/* scene flattened into the arrays rt_upload_scene takes */
/* allocator whose vectors leave plain structs unwritten on resize(): flatten() sizes the triangle arrays and fills them
 * with several threads, and zeroing them first would take one thread as long as the fill */
template <typename T>
struct UninitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = UninitAlloc<U>; };
    UninitAlloc() = default;
    template <typename U> UninitAlloc(const UninitAlloc<U> &) {}
    template <typename U> void construct(U *p) { ::new ((void *)p) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};

struct FlatScene {
    std::vector<rt_sphere> spheres;
    std::vector<rt_mesh> meshes;
    std::vector<rt_bvh_node, UninitAlloc<rt_bvh_node>> nodes;
    std::vector<rt_tri, UninitAlloc<rt_tri>> tris;
    std::vector<rt_tri_attr, UninitAlloc<rt_tri_attr>> attrs;
    std::vector<rt_texture> tex;
    std::vector<uint8_t, UninitAlloc<uint8_t>> texels;
    double lut[256];
    double I;
    double L[3];
    rt_scene_desc desc() const;
    size_t bytes() const;
};
void flatten(const Scene &scene, FlatScene &out);

class Renderer {
public:
    explicit Renderer(int device = 0);
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;
    void setScene(const Scene &scene);
    /* the whole pixel loop + tone map of main(): image = W*H*3 bytes, top row first */
    void render(const Camera &cam, const RenderParams &p, std::vector<uint8_t> &image);
    /* Scene::intersect for a batch of rays */
    void intersect(const std::vector<Ray> &rays, std::vector<int> &objectId, std::vector<int> &triangleId, std::vector<double> &t);
    double lastRenderMs();
    rt_ctx *ctx() { return ctx_; }
private:
    void check(int rc, const char *what);
    rt_ctx *ctx_ = nullptr;
};

/* convenience: one call = what running the reference's main() does up to the PNG write */
void render(const Scene &scene, const Camera &cam, const RenderParams &p, std::vector<uint8_t> &image, int device = 0);

/* The same on several GPUs of this machine (the reference's one parallel construct, OpenMP threads over rows at
 * raytracer.cpp:1456, becomes threads over GPUs): device d of `devices` gets its own context and copy of the scene and
 * renders its share of the nbrays sample indices of every pixel on a host thread of its own; the accumulators are
 * then added onto the first device in the order of the list (rt_accum_add: peer copies over xGMI) and resolved there.
 * The image does not depend on the number of devices up to the order of these last additions. A device may be listed
 * more than once (two contexts then share it). */
void render(const Scene &scene, const Camera &cam, const RenderParams &p, std::vector<uint8_t> &image, const std::vector<int> &devices);
/* the same for callers that hold the flattened scene and the PODs: p.spp samples from p.spp_offset on, rgb8 = W*H*3 bytes */
void renderDevices(const rt_scene_desc &d, const rt_camera &cam, const rt_params &p, const int *devices, int n_devices, uint8_t *rgb8);

/* JPEG decoding with the arithmetic of the reference's stbi_load(file, &W, &H, &c, 3) call (raytracer.cpp:762): the same
 * texels as that library gives, baseline and progressive (rt_jpeg.cpp). rgb receives W*H*3 bytes, rows top to bottom.
 * Throws std::runtime_error on files it cannot decode. */
void decodeJPEG(const uint8_t *data, size_t n, std::vector<uint8_t> &rgb, int &W, int &H);
bool looksLikeJPEG(const uint8_t *data, size_t n);

/* The output step of main(), raytracer.cpp:1505-1512. writePNG stores an 8-bit RGB image (rows top to bottom, as
 * Renderer::render returns it) as a PNG file: the same pixels as the reference's stbi_write_png call, in an
 * uncompressed zlib stream (any decoder reads it; the encoder is this library's own). writeExecutionTime writes the
 * whole seconds of a duration the way the reference writes its execution_time.txt. Both throw on I/O errors. */
void writePNG(const char *path, int W, int H, const uint8_t *rgb);
void writeExecutionTime(const char *path, double seconds);

} // namespace rt
