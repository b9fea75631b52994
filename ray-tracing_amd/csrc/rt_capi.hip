/*
 * rt_capi.hip - kernels and the C ABI (include/rt_amd.h) of the gfx950 path tracer.
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "rt_amd.h"
#include "rt_device.hip.h"
#include <map>
#include <mutex>

#include "rt_render_kernels.hip.h"
#include "rt_own_bvh.hpp"

using namespace rtd;

/* Scratch of rt_render_static, the entry point that allocates nothing: the tile counter of the persistent waves
 * (first RT_STATIC_HEADER bytes) and the sums of one launch's work units behind it. One copy per GPU, held from the
 * moment the library is loaded until the process ends. In host code this name is only a handle for
 * hipGetSymbolAddress (rt_create asks for the address on its device and keeps it in the context): it is never
 * passed to a kernel or a copy itself. */
#define RT_STATIC_HEADER 256
#define RT_STATIC_UNIT_BYTES (64ull << 20)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_HEADER + RT_STATIC_UNIT_BYTES];

/* ---------------------------------------------------------------------------------------------
 * kernels
 * ------------------------------------------------------------------------------------------- */

/* one path per thread: radiance of sample (i, j, k) */
__global__ void __launch_bounds__(RT_BLOCK) rt_sample_kernel(DScene sc, RenderArgs ra, const int32_t *ijk, int64_t n, double *out, int32_t *trace) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const rt_sphere *sph;
    const RootRec *roots;
    WaveLDS *w = wave_setup(sc, lds_raw, &sph, &roots);
    const int64_t q = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool live = q < n;
    Cnt cnt = {};
    const int i = live ? ijk[3 * q] : 0, j = live ? ijk[3 * q + 1] : 0, k = live ? ijk[3 * q + 2] : 0;
    int32_t *tr = (trace && live) ? trace + q * 4 * RT_TRACE_LEN : nullptr;
    if (tr)
        for (int e = 0; e < RT_TRACE_LEN; e++) { tr[4 * e] = tr[4 * e + 1] = -1; tr[4 * e + 2] = 0; tr[4 * e + 3] = RT_TRACE_UNUSED; }
    V3 a = trace_pixel<false, true>(sc, sc, w, sph, roots, ra, live, i, j, k, 1, cnt, tr);
    if (live) {
        out[3 * q] = a.x;
        out[3 * q + 1] = a.y;
        out[3 * q + 2] = a.z;
    }
}

/* Scene::intersect probe */
__global__ void __launch_bounds__(RT_BLOCK) rt_intersect_kernel(DScene sc, const double *rays, int64_t n, int prune, int32_t *obj_id,
                                                                int32_t *tri_id, double *t, double *PN, double *albedo) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const rt_sphere *sph;
    const RootRec *roots;
    WaveLDS *w = wave_setup(sc, lds_raw, &sph, &roots);
    const int64_t q = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool live = q < n;
    V3 C = mk(0, 0, 0), u = mk(0, 0, 1);
    if (live) { C = ld3(rays + 6 * q); u = ld3(rays + 6 * q + 3); }
    HitRec h;
    Cnt cnt = {};
    HitRec h2;
    const RayQ qa = RayQ{live, C, u, 1E10, false}, qb = RayQ{false, C, u, 1E10, false};
    scene_closest_wave<false, true>(sc, sc, w, sph, roots, qa, qb, prune != 0, h, h2, cnt);
    const bool ok = h.obj != INT32_MAX;
    if (!live) return;
    obj_id[q] = ok ? h.obj : -1;
    tri_id[q] = (ok && h.mesh >= 0 && h.prim >= 0) ? sc.tris[h.prim].orig_id : -1;
    t[q] = h.t;
    Surf s;
    s.P = s.N = s.albedo = mk(0, 0, 0);
    s.flags = 0;
    if (ok) hit_surface(sc, sph, C, u, h, s);
    PN[6 * q] = s.P.x; PN[6 * q + 1] = s.P.y; PN[6 * q + 2] = s.P.z;
    PN[6 * q + 3] = s.N.x; PN[6 * q + 4] = s.N.y; PN[6 * q + 5] = s.N.z;
    albedo[3 * q] = s.albedo.x; albedo[3 * q + 1] = s.albedo.y; albedo[3 * q + 2] = s.albedo.z;
}

/* self-test of the DPP wave scans used by the triangle step */
__global__ void __launch_bounds__(RT_BLOCK) rt_scan_test_kernel(const int *in, int *out_add, int *out_max) {
    const int v = in[threadIdx.x];
    out_add[threadIdx.x] = wave_scan_add(v);
    out_max[threadIdx.x] = wave_scan_max(v);
}

/* raytracer.cpp:1488-1501: /nbrays, pow 0.45, min(255, .), truncation to u8, row flip */
/* Adds the sums of a launch's work units to the accumulator, unit by unit in sample order (what a sequence of calls of
 * one unit each would have added): one thread per channel of a pixel of the rectangle. */
__global__ void rt_units_add_kernel(double *accum, const double *unit_sums, int n_units, int W, int H, int j0, int i0, int rw, int rh) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rw * rh * 3) return;
    const int ch = (int)(t % 3);
    const int64_t px = t / 3;
    const int i = i0 + (int)(px / rw), j = j0 + (int)(px % rw);
    const int64_t at = ((int64_t)i * W + j) * 3 + ch;
    double a = accum[at];
    for (int u = 0; u < n_units; u++) a = a + unit_sums[(int64_t)u * rw * rh * 3 + t]; /* unit_sums is [unit][rh][rw][3] */
    accum[at] = a;
}

/* The camera rays of one launch, made ahead of it: samples k0 .. k0 + nk - 1 of every pixel of the render rectangle, one
 * thread each, all lanes busy - inside the render kernel the same arithmetic runs with only the few lanes of a wave that
 * start a sample in the same iteration. out is [sample][row][column][RT_CAM_RAY_DOUBLES]. */
__global__ void __launch_bounds__(256) rt_camera_rays_kernel(RenderArgs ra, int k0, int nk, double *out) {
    const int rw = ra.j1 - ra.j0, rh = ra.i1 - ra.i0;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)nk * rw * rh) return;
    const int64_t npix = (int64_t)rw * rh;
    const int s = (int)(t / npix);
    const int pix = (int)(t - (int64_t)s * npix);
    const int i = ra.i0 + pix / rw, j = ra.j0 + pix % rw;
    const uint64_t key = rng_key(ra.seed_mixed, (uint32_t)(i * ra.p.W + j), (uint32_t)(k0 + s));
    V3 rC, ru;
    camera_ray(ra.cam, ra.p, i, j, key, rC, ru);
    double *o = out + t * RT_CAM_RAY_DOUBLES;
    o[0] = rC.x; o[1] = rC.y; o[2] = ru.x; o[3] = ru.y; o[4] = ru.z;
}

/* the filter tree's triangles: out[k] = tris[pos[k]], with orig_id = pos[k], the position in the reference's order */
__global__ void rt_gather_tris_kernel(const rt_tri *tris, const int32_t *pos, uint32_t n, rt_tri *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    rt_tri t = tris[pos[k]];
    t.orig_id = pos[k];
    out[k] = t;
}

__global__ void rt_accum_add_kernel(double *dst, const double *src, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = dst[t] + src[t];
}

/* tone[k], k = 1 .. 255: the smallest double v with pow(v, 0.45) >= k under the HOST's libm (tone_thresholds below), so
 * the byte (unsigned char)std::min(255., pow(v, 0.45)) of raytracer.cpp:1499-1501 is the number of thresholds not above
 * v - the same byte as the CPU program computes, whatever the device's own pow would round to at a boundary. A NaN or
 * negative v gives pow = NaN and std::min keeps its first argument: 255. */
__global__ void rt_resolve_kernel(const double *accum, int W, int H, double spp, const double *tone, uint8_t *rgb) {
    __shared__ double th[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) th[k] = tone[k];
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)W * H) return;
    const int i = (int)(q / W), j = (int)(q - (int64_t)i * W);
    const double *px = accum + q * 3;
    uint8_t *o = rgb + ((int64_t)(H - i - 1) * W + j) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double v = px[c] / spp;
        int lo = 0, hi = 256; /* the largest k in [0, 255] with v >= th[k], th[0] = 0 */
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int mid = (lo + hi) >> 1;
            if (v >= th[mid]) lo = mid; else hi = mid;
        }
        o[c] = (v != v || v < 0) ? (uint8_t)255 : (uint8_t)lo;
    }
}

/* ---------------------------------------------------------------------------------------------
 * host side
 * ------------------------------------------------------------------------------------------- */

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; /* around the launches of the last rt_render call */
    bool timed = false;
    int last_launches = 0;
    /* rt_render_time_ms: event pairs of the calls since the last reset (created on demand, reused after a reset) */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    /* rt_debug_kernel_time_ms: event pairs around the render kernel of each launch alone (without the camera-ray kernel before it
     * and the unit sums' addition after it), for comparison with a profiler's per-kernel figure */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev_pool;
    size_t kev_used = 0;
    int launches_since_reset = 0;
    std::string err;
    /* scene */
    bool have_scene = false;
    bool light_ok = false;
    DScene sc{};
    std::vector<void *> scene_allocs;
    size_t scene_bytes = 0; /* device bytes of the uploaded scene, every derived array included */
    size_t lds_bytes = 0;
    /* own accumulator */
    double *accum = nullptr;
    int accW = 0, accH = 0;
    double *unit_sums = nullptr; /* sums of the work units of one launch: [unit of the pixel][H][W][3] */
    size_t unit_sums_bytes = 0;
    double *cam_rays = nullptr;  /* camera rays of one launch, made ahead of it (rt_camera_rays_kernel) */
    size_t cam_rays_bytes = 0;
    bool cam_rays_refused = false; /* an allocation failed: not tried again for this size or larger */
    size_t cam_rays_refused_bytes = 0;
    bool no_cam_rays = false;    /* rt_debug_camera_rays(0): the render kernels compute their camera rays themselves */
    rt_counters *d_counters = nullptr;
    bool own_failed = false;      /* the scene has triangles the reference can reach but no filter tree over them (it could not be built) */
    void *d_fast_tris = nullptr;  /* RT_F_FAST: the filter tree's triangles in single precision, made at the first such render */
    size_t fast_lds_bytes = 0;
    int fast_waves_per_cu = 0;
    bool count_own = false; /* RT_F_COUNTERS renders walk and count the production traversal instead of the reference's */
    int *d_overflow = nullptr;
    int *d_next_slot = nullptr;
    double *d_tone = nullptr;    /* 256 tone-map thresholds (tone_thresholds) */
    uint8_t *d_rgb = nullptr;    /* rt_resolve's device image, kept and grown on demand */
    size_t d_rgb_bytes = 0;
    double *d_peer = nullptr;    /* rt_accum_add: another context's accumulator, copied here before it is added */
    size_t d_peer_bytes = 0;
    double *h_bounce = nullptr;  /* rt_accum_add without peer access: pinned host buffer the sums pass through */
    size_t h_bounce_bytes = 0;
    bool force_host_bounce = false;
    unsigned char *static_scratch = nullptr; /* this device's copy of `scratch` (hipGetSymbolAddress at rt_create) */
    int n_cus = 256;
    int waves_per_cu = 8;
    bool probe_prune = true;
    int force_own_leaf = 0; /* tests: triangles per leaf of the next uploads' filter trees (0: by scene size) */
};

static thread_local std::string g_create_err; /* rt_last_error(NULL): the calling thread's last rt_create failure */

/* Thresholds of the tone map for rt_resolve_kernel: T[0] = 0 and T[k] = the smallest double v >= 0 with
 * pow(v, 0.45) >= k, found by bisection over the bit patterns (which order like the values for v >= 0) with the host's
 * pow - the function the reference calls at raytracer.cpp:1499-1501. Relies on pow(., 0.45) not decreasing across a
 * threshold, which tests/test_host.py checks around every one of them. */
static void tone_thresholds(double *T) {
    T[0] = 0.0;
    for (int k = 1; k < 256; k++) {
        uint64_t lo = 0, hi;
        const double top = 1e7; /* pow(1e7, 0.45) = 1412 */
        memcpy(&hi, &top, 8);
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            double v;
            memcpy(&v, &mid, 8);
            if (pow(v, 0.45) >= (double)k) hi = mid; else lo = mid;
        }
        memcpy(&T[k], &hi, 8);
    }
}

static int fail(rt_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail((c), RT_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static void forget_static_owner(rt_ctx *c);

static void free_scene(rt_ctx *c) {
    for (void *p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->own_failed = false;
    if (c->d_fast_tris) (void)hipFree(c->d_fast_tris);
    c->d_fast_tris = nullptr;
    c->scene_bytes = 0;
    c->have_scene = false;
}

template <typename T>
static int upload_array(rt_ctx *c, const T *host, size_t n, const T **dev_out) {
    *dev_out = nullptr;
    if (n == 0) return RT_OK;
    void *d = nullptr;
    if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return fail(c, RT_E_NOMEM, "hipMalloc of %zu bytes failed", n * sizeof(T));
    c->scene_allocs.push_back(d);
    c->scene_bytes += n * sizeof(T);
    HIPCHK(c, hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev_out = static_cast<const T *>(d);
    return RT_OK;
}

/* the pool's node stack is sized from the validated BVH depth; a push past it is dropped and flagged */
static int check_overflow(rt_ctx *c) {
    int flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, c->d_overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (flag) {
        (void)hipMemsetAsync(c->d_overflow, 0, sizeof(int), c->stream);
        return fail(c, RT_E_DEVICE, "traversal pool %s: results of the last launch are incomplete",
                    (flag & 8) ? "did not terminate" : (flag & 2) ? "produced an index outside the scene arrays" : "overflow");
    }
    return RT_OK;
}

extern "C" {

#ifndef RT_BUILD_SCHED
#define RT_BUILD_SCHED "default"
#endif
/* names the instruction scheduler the render kernels were compiled with (csrc/Makefile falls back to the default one if the
 * compiler fails with the preferred one) */
const char *rt_version(void) { return "rt_amd 0.1 (gfx950, fp64, -ffp-contract=off, scheduler " RT_BUILD_SCHED ")"; }

int rt_create(int device, rt_ctx **out) {
    if (!out) return fail(nullptr, RT_E_INVALID, "rt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, RT_E_DEVICE, "rt_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, RT_E_INVALID, "rt_create: device %d out of range [0,%d)", device, n);
    rt_ctx *c = new rt_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_counters, RT_COUNTER_WORDS * 8)) != hipSuccess ||
        (e = hipMemset(c->d_counters, 0, RT_COUNTER_WORDS * 8)) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_overflow, sizeof(int))) != hipSuccess || (e = hipMemset(c->d_overflow, 0, sizeof(int))) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_next_slot, sizeof(int))) != hipSuccess || (e = hipMalloc((void **)&c->d_tone, 256 * sizeof(double))) != hipSuccess) {
        fail(nullptr, RT_E_DEVICE, "rt_create: %s", hipGetErrorString(e));
        delete c;
        return RT_E_DEVICE;
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    }
    {
        double T[256];
        tone_thresholds(T);
        if ((e = hipMemcpy(c->d_tone, T, sizeof T, hipMemcpyHostToDevice)) != hipSuccess) {
            fail(nullptr, RT_E_DEVICE, "rt_create: %s", hipGetErrorString(e));
            rt_destroy(c);
            return RT_E_DEVICE;
        }
    }
    /* every GPU holds its own copy of the static scratch array at its own address: ask with this device current */
    if ((e = hipGetSymbolAddress((void **)&c->static_scratch, HIP_SYMBOL(scratch))) != hipSuccess || c->static_scratch == nullptr) {
        fail(nullptr, RT_E_DEVICE, "rt_create: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
        rt_destroy(c);
        return RT_E_DEVICE;
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    forget_static_owner(c);
    free_scene(c);
    if (c->accum) (void)hipFree(c->accum);
    if (c->unit_sums) (void)hipFree(c->unit_sums);
    if (c->cam_rays) (void)hipFree(c->cam_rays);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_overflow) (void)hipFree(c->d_overflow);
    if (c->d_next_slot) (void)hipFree(c->d_next_slot);
    if (c->d_tone) (void)hipFree(c->d_tone);
    if (c->d_peer) (void)hipFree(c->d_peer);
    if (c->h_bounce) (void)hipHostFree(c->h_bounce);
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto &e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : c->kev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *rt_last_error(const rt_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

void *rt_stream(rt_ctx *c) { return c ? (void *)c->stream : nullptr; }

/* a pending extension record: the leaf range it splits and where its index has to be written */
struct ExtJob {
    int64_t parent; /* record index, or -1 - mesh for a mesh whose root is a leaf */
    int side;
    int32_t first, count;
    int depth;
};

/* float box of triangles [first, first + n), rounded outwards */
static void tri_range_box(const rt_tri *tris, int32_t first, int32_t n, float *box) {
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (int32_t k = first; k < first + n; k++) {
        const rt_tri &t = tris[k];
        for (int a = 0; a < 3; a++) {
            const double v[3] = {t.A[a], t.A[a] + t.e1[a], t.A[a] + t.e2[a]};
            for (int q = 0; q < 3; q++) {
                if (v[q] < lo[a]) lo[a] = v[q];
                if (v[q] > hi[a]) hi[a] = v[q];
                if (!(v[q] == v[q])) { lo[a] = -HUGE_VAL; hi[a] = HUGE_VAL; } /* a NaN vertex: never filter */
            }
        }
    }
    for (int a = 0; a < 3; a++) {
        box[a] = nextafterf((float)lo[a], -HUGE_VALF);
        box[3 + a] = nextafterf((float)hi[a], HUGE_VALF);
    }
}

#define RT_EXT_SAH_DEPTH 24 /* below this depth of extension records a range is simply halved (bounds the depth) */

static inline void grow(double *lo, double *hi, const rt_tri &t) {
    for (int a = 0; a < 3; a++) {
        const double v[3] = {t.A[a], t.A[a] + t.e1[a], t.A[a] + t.e2[a]};
        for (int q = 0; q < 3; q++) {
            if (v[q] < lo[a]) lo[a] = v[q];
            if (v[q] > hi[a]) hi[a] = v[q];
        }
    }
}
static inline double half_area(const double *lo, const double *hi) {
    const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + y * z + z * x;
}
/* Where to cut triangles [first, first + n) into two contiguous parts: the cut with the least surface-area cost
 * (area of each part's box times its triangle count). The order of the triangles is fixed, so only n - 1 cuts exist. */
static int32_t ext_split(const rt_tri *tris, int32_t first, int32_t n, std::vector<double> &right_area) {
    right_area.resize((size_t)n + 1);
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (int32_t k = n - 1; k >= 1; k--) {
        grow(lo, hi, tris[first + k]);
        right_area[(size_t)k] = half_area(lo, hi);
    }
    for (int a = 0; a < 3; a++) { lo[a] = HUGE_VAL; hi[a] = -HUGE_VAL; }
    int32_t best = n / 2;
    double best_cost = HUGE_VAL;
    for (int32_t k = 1; k < n; k++) { /* k triangles on the left */
        grow(lo, hi, tris[first + k - 1]);
        const double cost = half_area(lo, hi) * k + right_area[(size_t)k] * (n - k);
        if (cost < best_cost) { best_cost = cost; best = k; }
    }
    return best;
}

/* The tree over a long sphere list: the filter tree's builder (rt_own_bvh.hpp: binned surface-area cuts, four children per
 * record, leaves of RT_SPH_LEAF = one sphere) over the spheres' boxes; the spheres are re-ordered into its leaf order. `reach`
 * is how far from the coordinate origin a ray may start for the margin of the boxes to hold: a hundred thousand times the
 * smallest radius, less the largest centre coordinate. (Sphere::intersect accepts delta = b^2 - 4c >= 0 as computed; its
 * rounding error is below 64 * 2^-53 |C - O|^2, so an accepted ray passes the centre within R sqrt(1 + 16 * 2^-53 |C - O|^2 / R^2)
 * <= R (1 + 3e-5) when |C - O| <= sqrt(3) * 1e5 R: inside the boxes' margin of R / 1000.) Returns false for a non-finite sphere. */
static bool build_sphere_tree(const rt_sphere *sph, int n, std::vector<rt_sphere> &sorted, rtown::Built &tree, std::vector<double> &boxes_out, double *reach,
                              int *n_big_out) {
    std::vector<int> idx((size_t)n);
    double rmin = HUGE_VAL, omax = 0;
    for (int i = 0; i < n; i++) {
        idx[(size_t)i] = i;
        if (!std::isfinite(sph[i].R) || !(sph[i].R >= 0)) return false;
        for (int a = 0; a < 3; a++) {
            if (!std::isfinite(sph[i].O[a])) return false;
            if (fabs(sph[i].O[a]) > omax) omax = fabs(sph[i].O[a]);
        }
        if (sph[i].R < rmin) rmin = sph[i].R;
    }
    *reach = 1e5 * rmin - omax; /* <= 0: every ray tests every sphere */
    /* Spheres much larger than the rest (the walls of the reference's scenes are spheres of radius ~1000) would blow up
     * every box above them: they stay out of the tree, at the front of the list, and every ray tests them first - which
     * also gives the traversal a closest hit to prune with. */
    int n_big = 0;
    {
        std::vector<double> radii((size_t)n);
        for (int i = 0; i < n; i++) radii[(size_t)i] = sph[i].R;
        std::nth_element(radii.begin(), radii.begin() + n / 2, radii.end());
        const double big = 4 * radii[(size_t)(n / 2)];
        n_big = (int)(std::stable_partition(idx.begin(), idx.end(), [&](int i) { return sph[i].R > big; }) - idx.begin());
        if (n_big > RT_LDS_SPHERES_MAX || n_big == n) n_big = 0; /* no such outliers (or only such): everything goes into the tree */
    }
    *n_big_out = n_big;
    std::vector<rtown::Bounds> tb((size_t)(n - n_big));
    for (int k = n_big; k < n; k++) {
        const rt_sphere &sp = sph[idx[(size_t)k]];
        const double m = sp.R * 1.001 + 1e-300;
        rtown::Bounds &b = tb[(size_t)(k - n_big)];
        for (int a = 0; a < 3; a++) { b.lo[a] = sp.O[a] - m; b.hi[a] = sp.O[a] + m; }
        for (int a = 0; a < 3; a++) { boxes_out.push_back(b.lo[a]); }
        for (int a = 0; a < 3; a++) { boxes_out.push_back(b.hi[a]); }
    }
    tree = rtown::build_boxes(tb);
    if (!tree.ok) return false;
    sorted.resize((size_t)n);
    for (int i = 0; i < n_big; i++) sorted[(size_t)i] = sph[idx[(size_t)i]];
    for (int k = 0; k < n - n_big; k++) sorted[(size_t)(n_big + k)] = sph[idx[(size_t)(n_big + tree.order[(size_t)k])]];
    return true;
}

int rt_upload_scene(rt_ctx *c, const rt_scene_desc *s) {
    if (!c) return RT_E_INVALID;
    rtown::stage_tick(nullptr);
    rtown::PoolScope pool(rtown::host_threads()); /* the caller's, when it has one already */
    if (!s) return fail(c, RT_E_INVALID, "rt_upload_scene: scene is NULL");
    if (s->n_spheres < 0 || s->n_meshes < 0 || s->n_nodes < 0 || s->n_tris < 0 || s->n_tex < 0)
        return fail(c, RT_E_INVALID, "rt_upload_scene: negative count");
    if ((s->n_spheres && !s->spheres) || (s->n_meshes && !s->meshes) || (s->n_nodes && !s->nodes) ||
        (s->n_tris && (!s->tris || !s->attrs)) || (s->n_tex && (!s->tex || !s->texels)))
        return fail(c, RT_E_INVALID, "rt_upload_scene: NULL array with non-zero count");
    if (s->n_tris >= (1LL << RT_NODE_BITS) || s->n_nodes >= (1LL << RT_NODE_BITS)) return fail(c, RT_E_INVALID, "rt_upload_scene: at most 2^25 triangles and BVH records");
    if (s->n_meshes > 0 && s->n_tris != (int64_t)s->meshes[s->n_meshes - 1].first_tri + s->meshes[s->n_meshes - 1].n_tris)
        return fail(c, RT_E_INVALID, "rt_upload_scene: n_tris does not match the mesh ranges");
    /* validate everything the kernels index with, so a bad description is an error and not a GPU fault */
    const int n_obj = s->n_spheres + s->n_meshes;
    std::vector<char> seen(n_obj, 0);
    const rt_sphere *light = nullptr;
    for (int i = 0; i < s->n_spheres; i++) {
        int id = s->spheres[i].object_id;
        if (id < 0 || id >= n_obj || seen[id]) return fail(c, RT_E_INVALID, "sphere %d: object_id %d invalid or repeated", i, id);
        seen[id] = 1;
        if (id == 0) light = &s->spheres[i];
    }
    int max_depth = 0;
    for (int mi = 0; mi < s->n_meshes; mi++) {
        const rt_mesh &m = s->meshes[mi];
        if (m.max_depth > max_depth) max_depth = m.max_depth;
        /* ties between meshes go to the lower objectId (raytracer.cpp:938) = the lower triangle position */
        if (mi > 0 && (m.object_id <= s->meshes[mi - 1].object_id || m.first_tri != s->meshes[mi - 1].first_tri + s->meshes[mi - 1].n_tris))
            return fail(c, RT_E_INVALID, "mesh %d: meshes must be in ascending object_id with consecutive triangle ranges", mi);
        if (mi == 0 && m.first_tri != 0) return fail(c, RT_E_INVALID, "mesh 0: first_tri must be 0");
        if (m.object_id < 0 || m.object_id >= n_obj || seen[m.object_id]) return fail(c, RT_E_INVALID, "mesh %d: object_id %d invalid or repeated", mi, m.object_id);
        seen[m.object_id] = 1;
        if (m.n_tris < 0 || m.first_tri < 0 || (int64_t)m.first_tri + m.n_tris > s->n_tris) return fail(c, RT_E_INVALID, "mesh %d: triangle range outside the triangle array", mi);
        if (m.n_tex < 0 || m.first_tex < 0 || m.first_tex + m.n_tex > s->n_tex) return fail(c, RT_E_INVALID, "mesh %d: texture range outside the texture array", mi);
        if (m.max_depth < 0 || m.max_depth > RT_MAX_DEPTH) return fail(c, RT_E_INVALID, "mesh %d: BVH depth %d outside [0,%d]", mi, m.max_depth, RT_MAX_DEPTH);
        {
            /* per triangle: its mesh index and, on a textured mesh, its texture group (the reference indexes
             * textures[group] unchecked, raytracer.cpp:700-701); shared out over the threads, the lowest offender is reported */
            const int thr = m.n_tris >= (1 << 16) ? rtown::host_threads() : 1;
            std::vector<int64_t> bad_mesh((size_t)thr, -1), bad_group((size_t)thr, -1);
            rtown::parallel_parts(m.n_tris, thr, [&](int part, int64_t k0, int64_t k1) {
                for (int64_t k = m.first_tri + k0; k < m.first_tri + k1; k++) {
                    if (s->tris[k].mesh != mi && bad_mesh[(size_t)part] < 0) bad_mesh[(size_t)part] = k;
                    if (m.n_tex > 0 && (s->attrs[k].group < 0 || s->attrs[k].group >= m.n_tex) && bad_group[(size_t)part] < 0) bad_group[(size_t)part] = k;
                }
            });
            for (int64_t k : bad_mesh)
                if (k >= 0) return fail(c, RT_E_INVALID, "mesh %d: triangle %lld carries mesh index %d", mi, (long long)k, s->tris[k].mesh);
            for (int64_t k : bad_group)
                if (k >= 0) return fail(c, RT_E_INVALID, "mesh %d: triangle %lld has texture group %d but the mesh has %d textures", mi, (long long)k, s->attrs[k].group, m.n_tex);
        }
        if (m.root_count < 0) return fail(c, RT_E_INVALID, "mesh %d: negative root_count", mi);
        if (m.root_count > 0) {
            if (m.root < m.first_tri || (int64_t)m.root + m.root_count > (int64_t)m.first_tri + m.n_tris) return fail(c, RT_E_INVALID, "mesh %d: root leaf outside its triangles", mi);
        } else if (m.n_tris == 0) {
            return fail(c, RT_E_INVALID, "mesh %d: empty mesh", mi);
        } else if (m.root < 0 || m.root >= s->n_nodes) {
            return fail(c, RT_E_INVALID, "mesh %d: root node outside the node array", mi);
        }
    }
    /* largest |coordinate| of any box, per axis: scales the tolerance of the fast box test (rt_device.hip.h) */
    double bmax[3] = {0, 0, 0};
    auto widen_into = [](double *bm, const double *box) {
        for (int k = 0; k < 6; k++) {
            const double a = fabs(box[k]);
            if (!(a <= bm[k % 3])) bm[k % 3] = std::isfinite(a) && std::isfinite(bm[k % 3]) ? a : HUGE_VAL;
        }
    };
    auto widen = [&](const double *box) { widen_into(bmax, box); };
    for (int mi = 0; mi < s->n_meshes; mi++) widen(s->meshes[mi].root_box);
    {
        const int thr = s->n_nodes >= (1 << 16) ? rtown::host_threads() : 1;
        struct NodeCheck { double bm[3] = {0, 0, 0}; int64_t bad = -1; int what = 0; };
        std::vector<NodeCheck> part((size_t)thr);
        rtown::parallel_parts(s->n_nodes, thr, [&](int p, int64_t k0, int64_t k1) {
            NodeCheck &P = part[(size_t)p];
            for (int64_t k = k0; k < k1; k++) {
                const rt_bvh_node &nd = s->nodes[k];
                widen_into(P.bm, nd.boxL);
                widen_into(P.bm, nd.boxR);
                if (P.bad >= 0) continue;
                const int32_t idx[2] = {nd.L, nd.R}, cnt[2] = {nd.nL, nd.nR};
                for (int q = 0; q < 2 && P.bad < 0; q++) {
                    if (cnt[q] < 0) { P.bad = k; P.what = -1; }
                    else if (cnt[q] == 0 ? (idx[q] <= k || idx[q] >= s->n_nodes) : (idx[q] < 0 || (int64_t)idx[q] + cnt[q] > s->n_tris)) { P.bad = k; P.what = 1 + q; }
                }
            }
        });
        for (const NodeCheck &P : part) {
            if (P.bad >= 0 && P.what < 0) return fail(c, RT_E_INVALID, "node %lld: negative leaf count", (long long)P.bad);
            if (P.bad >= 0) return fail(c, RT_E_INVALID, "node %lld: child %d out of range (inner children must have a larger index)", (long long)P.bad, P.what - 1);
            for (int a = 0; a < 3; a++)
                if (!(P.bm[a] <= bmax[a])) bmax[a] = std::isfinite(P.bm[a]) && std::isfinite(bmax[a]) ? P.bm[a] : HUGE_VAL;
        }
    }
    for (int k = 0; k < s->n_tex; k++) {
        const rt_texture &tx = s->tex[k];
        if (tx.W <= 0 || tx.H <= 0 || tx.offset < 0 || tx.offset + (int64_t)tx.W * tx.H * 3 > s->texel_bytes)
            return fail(c, RT_E_INVALID, "texture %d: outside the texel pool", k);
    }
    rtown::stage_tick("rt_upload_scene: validation");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    DScene d{};
    int rc;
    int ext_depth = 0, own_depth = 0, sph_depth = 0;
    std::vector<int32_t> ext_root((size_t)s->n_meshes, -1); /* extension record that replaces a mesh's root leaf */
    std::vector<double> sphere_boxes; /* of the sphere tree: they count for the box tests' tolerance like every other box */
    if (s->n_spheres > RT_LDS_SPHERES_MAX) {
        /* Long sphere lists: a tree over the spheres' boxes in the filter tree's format; the spheres go up in its leaf order. */
        std::vector<rt_sphere> sorted;
        rtown::Built tree;
        double reach = 0;
        int n_big = 0;
        if (!build_sphere_tree(s->spheres, s->n_spheres, sorted, tree, sphere_boxes, &reach, &n_big))
            return fail(c, RT_E_INVALID, "rt_upload_scene: a sphere has a non-finite centre or radius");
        std::vector<int32_t> of_obj((size_t)n_obj, -1);
        for (size_t k = 0; k < sorted.size(); k++) of_obj[(size_t)sorted[k].object_id] = (int32_t)k;
        if ((rc = upload_array(c, sorted.data(), sorted.size(), &d.spheres))) return rc;
        if ((rc = upload_array(c, tree.wide.data(), tree.wide.size(), &d.sph_rec4))) return rc;
        if ((rc = upload_array(c, of_obj.data(), of_obj.size(), &d.sph_of_obj))) return rc;
        d.n_sph_rec4 = (uint32_t)tree.wide.size();
        d.sph_reach = reach;
        d.n_sph_big = n_big;
        sph_depth = 3 * tree.wide_depth + 2; /* stack entries the depth-first tail of the traversal can add */
        for (size_t k = 0; k + 6 <= sphere_boxes.size(); k += 6) widen(&sphere_boxes[k]);
        HIPCHK(c, hipStreamSynchronize(c->stream)); /* the staging vectors go out of scope */
    } else if ((rc = upload_array(c, s->spheres, (size_t)s->n_spheres, &d.spheres))) {
        return rc;
    }
    if ((rc = upload_array(c, s->meshes, (size_t)s->n_meshes, &d.meshes))) return rc;
    {
        /* device layouts (rt_device.hip.h): every BVH record as one float record plus two double halves */
        const int host_thr = rtown::host_threads();
        rtown::raw_vector<NodeHalf> halves((size_t)s->n_nodes * 2);
        rtown::raw_vector<NodeRec32> rec32((size_t)s->n_nodes);
        rtown::parallel_parts(s->n_nodes, s->n_nodes >= (1 << 16) ? host_thr : 1, [&](int, int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; k++) {
                const rt_bvh_node &nd = s->nodes[k];
                NodeHalf &l = halves[(size_t)k * 2], &r = halves[(size_t)k * 2 + 1];
                memset(&l, 0, sizeof l);
                memset(&r, 0, sizeof r);
                for (int q = 0; q < 6; q++) { l.box[q] = nd.boxL[q]; r.box[q] = nd.boxR[q]; }
                l.child = nd.L; l.count = nd.nL;
                r.child = nd.R; r.count = nd.nR;
                for (int a = 0; a < 6; a++) { rec32[(size_t)k].boxL[a] = (float)nd.boxL[a]; rec32[(size_t)k].boxR[a] = (float)nd.boxR[a]; }
                rec32[(size_t)k].L = nd.L; rec32[(size_t)k].R = nd.R;
                rec32[(size_t)k].nL = nd.nL; rec32[(size_t)k].nR = nd.nR;
            }
        });
        rtown::stage_tick("rt_upload_scene: spheres, record repack");
        /* Extension records: the reference's builder leaves ranges of up to hundreds of triangles unsplit (:283). Such a
         * leaf is given a tree of its own over contiguous sub-ranges of its triangles (their order, which carries the
         * tie rule, is untouched). These boxes are a filter only - a ray is kept out of a sub-range just when it
         * certainly misses its box - so they change which triangles are tested, never the result. */
        ext_depth = 0;
        {
            /* every such leaf is a job of its own: the threads take leaves from a counter, build each leaf's records into
             * an array of the leaf's own (record 0 = the leaf's root, children after their parents), and the arrays are
             * appended in the order of the leaves, so the result does not depend on the thread count */
            std::vector<ExtJob> roots;
            for (int64_t k = 0; k < s->n_nodes; k++) {
                if (rec32[(size_t)k].nL > RT_EXT_LEAF) roots.push_back(ExtJob{k, 0, rec32[(size_t)k].L, rec32[(size_t)k].nL, 1});
                if (rec32[(size_t)k].nR > RT_EXT_LEAF) roots.push_back(ExtJob{k, 1, rec32[(size_t)k].R, rec32[(size_t)k].nR, 1});
            }
            for (int mi = 0; mi < s->n_meshes; mi++)
                if (s->meshes[mi].root_count > RT_EXT_LEAF) roots.push_back(ExtJob{-1 - (int64_t)mi, 0, s->meshes[mi].root, s->meshes[mi].root_count, 1});
            struct Part { std::vector<NodeRec32> recs; int depth = 0; };
            std::vector<Part> parts(roots.size());
            std::atomic<size_t> next(0);
            auto work = [&] {
                std::vector<ExtJob> jobs;
                std::vector<double> sweep;
                for (size_t t; (t = next.fetch_add(1)) < roots.size();) {
                    Part &P = parts[t];
                    jobs.clear();
                    jobs.push_back(ExtJob{-1, 0, roots[t].first, roots[t].count, 1});
                    for (size_t q = 0; q < jobs.size(); q++) { /* breadth-first: children always get a larger index than their parent */
                        const ExtJob jb = jobs[q];
                        const int64_t id = (int64_t)P.recs.size();
                        NodeRec32 r;
                        const int32_t nl = (jb.depth <= RT_EXT_SAH_DEPTH) ? ext_split(s->tris, jb.first, jb.count, sweep) : jb.count / 2;
                        const int32_t nr = jb.count - nl;
                        tri_range_box(s->tris, jb.first, nl, r.boxL);
                        tri_range_box(s->tris, jb.first + nl, nr, r.boxR);
                        r.L = jb.first; r.nL = nl;
                        r.R = jb.first + nl; r.nR = nr;
                        P.recs.push_back(r);
                        if (jb.parent >= 0) { /* the parent's leaf entry becomes an inner child (index within this leaf's array for now) */
                            NodeRec32 &pr = P.recs[(size_t)jb.parent];
                            if (jb.side == 0) { pr.L = (int32_t)id; pr.nL = 0; } else { pr.R = (int32_t)id; pr.nR = 0; }
                        }
                        if (jb.depth > P.depth) P.depth = jb.depth;
                        if (nl > RT_EXT_LEAF) jobs.push_back(ExtJob{id, 0, jb.first, nl, jb.depth + 1});
                        if (nr > RT_EXT_LEAF) jobs.push_back(ExtJob{id, 1, jb.first + nl, nr, jb.depth + 1});
                    }
                }
            };
            {
                const int threads = roots.size() >= 64 ? rtown::host_threads() : 1;
                std::vector<std::thread> th;
                for (int p = 1; p < threads; p++) th.emplace_back(work);
                work();
                for (auto &t : th) t.join();
            }
            for (size_t t = 0; t < roots.size(); t++) {
                const int64_t off = (int64_t)rec32.size();
                if (off + (int64_t)parts[t].recs.size() >= (1LL << RT_NODE_BITS)) return fail(c, RT_E_INVALID, "rt_upload_scene: more than 2^25 BVH records with the extension records");
                for (NodeRec32 r : parts[t].recs) {
                    if (r.nL == 0) r.L += (int32_t)off;
                    if (r.nR == 0) r.R += (int32_t)off;
                    rec32.push_back(r);
                }
                if (roots[t].parent >= 0) {
                    NodeRec32 &pr = rec32[(size_t)roots[t].parent];
                    if (roots[t].side == 0) { pr.L = (int32_t)off; pr.nL = 0; } else { pr.R = (int32_t)off; pr.nR = 0; }
                } else {
                    ext_root[(size_t)(-1 - roots[t].parent)] = (int32_t)off;
                }
                if (parts[t].depth > ext_depth) ext_depth = parts[t].depth;
            }
        }
        /* what the triangle step relies on (rt_device.hip.h: RT_LEAF_QUAD): no leaf of the traversal records is longer than RT_EXT_LEAF */
        for (const NodeRec32 &r : rec32)
            if (r.nL > RT_EXT_LEAF || r.nR > RT_EXT_LEAF) return fail(c, RT_E_INVALID, "rt_upload_scene: internal error, a leaf of %d triangles survived the extension records", r.nL > r.nR ? r.nL : r.nR);
        rtown::stage_tick("rt_upload_scene: extension records");
        /* The filter tree over the triangles of all meshes, and what confirms its winners: the reference leaf of every
         * triangle. A leaf whose box has no thickness along an axis is never entered by the reference (tMax > tMin cannot
         * hold, raytracer.cpp:184), so its triangles are left out of the filter tree: they cannot be part of any answer. */
        if ((rc = upload_array(c, s->tris, (size_t)s->n_tris, &d.tris))) return rc;
        if (s->n_tris > 0) {
            struct LeafRef { const double *box; int32_t first, count; };
            std::vector<LeafRef> leaves;
            for (int64_t k = 0; k < s->n_nodes; k++) {
                if (s->nodes[k].nL > 0) leaves.push_back(LeafRef{s->nodes[k].boxL, s->nodes[k].L, s->nodes[k].nL});
                if (s->nodes[k].nR > 0) leaves.push_back(LeafRef{s->nodes[k].boxR, s->nodes[k].R, s->nodes[k].nR});
            }
            for (int mi = 0; mi < s->n_meshes; mi++)
                if (s->meshes[mi].root_count > 0) leaves.push_back(LeafRef{s->meshes[mi].root_box, s->meshes[mi].root, s->meshes[mi].root_count});
            const int tri_thr = s->n_tris >= (1 << 16) ? host_thr : 1;
            rtown::raw_vector<int32_t> tri_leaf((size_t)s->n_tris);
            rtown::parallel_parts(s->n_tris, tri_thr, [&](int, int64_t k0, int64_t k1) { std::fill(tri_leaf.begin() + k0, tri_leaf.begin() + k1, -1); });
            std::vector<float> leaf_box(leaves.size() * 6);
            std::vector<char> leaf_flat(leaves.size());
            std::vector<int64_t> part_sum((size_t)tri_thr, 0);
            rtown::parallel_parts((int64_t)leaves.size(), tri_thr, [&](int p, int64_t l0, int64_t l1) {
                for (int64_t id = l0; id < l1; id++) {
                    const LeafRef &lf = leaves[(size_t)id];
                    for (int a = 0; a < 6; a++) leaf_box[(size_t)id * 6 + a] = (float)lf.box[a];
                    leaf_flat[(size_t)id] = lf.box[0] == lf.box[3] || lf.box[1] == lf.box[4] || lf.box[2] == lf.box[5];
                    for (int32_t k = lf.first; k < lf.first + lf.count; k++) tri_leaf[(size_t)k] = (int32_t)id;
                    part_sum[(size_t)p] += lf.count;
                }
            });
            /* positions of the triangles that some leaf with a box of three non-zero extents holds (a triangle outside
             * every leaf is never reached by the reference either), in ascending order */
            rtown::raw_vector<int32_t> keep;
            {
                std::vector<int64_t> cnt((size_t)tri_thr + 1, 0), held((size_t)tri_thr, 0);
                auto kept = [&](int64_t k) { return tri_leaf[(size_t)k] >= 0 && !leaf_flat[(size_t)tri_leaf[(size_t)k]]; };
                rtown::parallel_parts(s->n_tris, tri_thr, [&](int p, int64_t k0, int64_t k1) {
                    int64_t n = 0, h = 0;
                    for (int64_t k = k0; k < k1; k++) { n += kept(k); h += tri_leaf[(size_t)k] >= 0; }
                    cnt[(size_t)p + 1] = n;
                    held[(size_t)p] = h;
                });
                int64_t in_leaves = 0, in_ranges = 0;
                for (int p = 0; p < tri_thr; p++) { cnt[(size_t)p + 1] += cnt[(size_t)p]; in_leaves += held[(size_t)p]; in_ranges += part_sum[(size_t)p]; }
                /* two leaves naming one triangle: the reference would test it under either box, which one leaf per triangle cannot express */
                if (in_leaves != in_ranges) return fail(c, RT_E_INVALID, "rt_upload_scene: BVH leaves overlap (%lld triangles in ranges that hold %lld)", (long long)in_leaves, (long long)in_ranges);
                keep.resize((size_t)cnt[(size_t)tri_thr]);
                rtown::parallel_parts(s->n_tris, tri_thr, [&](int p, int64_t k0, int64_t k1) {
                    int32_t *out = keep.data() + cnt[(size_t)p];
                    for (int64_t k = k0; k < k1; k++)
                        if (kept(k)) *out++ = (int32_t)k;
                });
            }
            rtown::stage_tick("rt_upload_scene: leaf table, triangle subset");
            c->own_failed = false;
            if (!keep.empty()) {
                c->own_failed = true; /* until the tree is on the device */
                /* Leaves of two triangles up to RT_OWN_PAIR_MAX triangles, of RT_OWN_LEAF beyond: a triangle step then runs two
                 * lanes per leaf item instead of four (fuller steps: +1.5 % on the 0.7 M-triangle scene), for 40 % more
                 * records - which is what costs on the 5 M-triangle scene (-1 %, set-up +10 %). */
                const int own_leaf = c->force_own_leaf ? c->force_own_leaf : ((int64_t)keep.size() <= RT_OWN_PAIR_MAX && RT_OWN_LEAF > 2) ? 2 : RT_OWN_LEAF;
                rtown::Built own = rtown::build(s->tris, keep.data(), (int64_t)keep.size(), own_leaf);
                d.own_leaf_shift = own_leaf <= 2 ? 1 : 2;
                rtown::stage_tick("rt_upload_scene: filter tree build");
                if (own.ok && (int64_t)own.nodes.size() < (1LL << RT_NODE_BITS)) {
                    /* the filter tree's triangle array is gathered on the device from the triangles already there */
                    rtown::raw_vector<int32_t> pos(keep.size());
                    rtown::parallel_parts((int64_t)keep.size(), keep.size() >= (1u << 16) ? host_thr : 1, [&](int, int64_t k0, int64_t k1) {
                        for (int64_t k = k0; k < k1; k++) pos[(size_t)k] = keep[(size_t)own.order[(size_t)k]];
                    });
                    void *d_pos = nullptr, *d_own = nullptr;
                    if (hipMalloc(&d_own, pos.size() * sizeof(rt_tri)) != hipSuccess) return fail(c, RT_E_NOMEM, "hipMalloc of %zu bytes failed", pos.size() * sizeof(rt_tri));
                    c->scene_allocs.push_back(d_own);
                    c->scene_bytes += pos.size() * sizeof(rt_tri);
                    if (hipMalloc(&d_pos, pos.size() * sizeof(int32_t)) != hipSuccess) return fail(c, RT_E_NOMEM, "hipMalloc of %zu bytes failed", pos.size() * sizeof(int32_t));
                    hipError_t e = hipMemcpyAsync(d_pos, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
                    if (e == hipSuccess) {
                        const uint32_t n = (uint32_t)pos.size();
                        rt_gather_tris_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(d.tris, (const int32_t *)d_pos, n, (rt_tri *)d_own);
                        e = hipGetLastError();
                    }
                    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                    (void)hipFree(d_pos);
                    HIPCHK(c, e);
                    d.own_tris = (const rt_tri *)d_own;
                    if ((rc = upload_array(c, own.wide.data(), own.wide.size(), &d.own_rec4))) return rc;
#if !RT_TRI_FIN
                    if ((rc = upload_array(c, tri_leaf.data(), tri_leaf.size(), &d.tri_leaf))) return rc;
                    if ((rc = upload_array(c, leaf_box.data(), leaf_box.size(), &d.leaf_box32))) return rc;
#else
                    {
                        rtown::raw_vector<TriFin> fin((size_t)s->n_tris);
                        rtown::parallel_parts(s->n_tris, tri_thr, [&](int, int64_t k0, int64_t k1) {
                            for (int64_t k = k0; k < k1; k++) {
                                TriFin &f = fin[(size_t)k];
                                const int32_t lf = tri_leaf[(size_t)k];
                                for (int a = 0; a < 6; a++) f.box[a] = lf >= 0 ? leaf_box[(size_t)lf * 6 + a] : 0.f;
                                f.object_id = s->meshes[s->tris[k].mesh].object_id;
                                f.group = s->attrs[k].group;
                            }
                        });
                        if ((rc = upload_array(c, fin.data(), fin.size(), &d.tri_fin))) return rc;
                        HIPCHK(c, hipStreamSynchronize(c->stream));
                    }
#endif
                    d.n_own_nodes = (uint32_t)own.wide.size();
                    d.n_own_tris = (uint32_t)pos.size();
                    c->own_failed = false;
                    own_depth = 3 * own.wide_depth + 2; /* stack entries the depth-first tail of a wide traversal can add */
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                }
            }
        }
        rtown::stage_tick("rt_upload_scene: filter tree reorder + copies");
        if ((rc = upload_array(c, rec32.data(), rec32.size(), &d.rec32))) return rc;
        d.n_nodes_total = (uint32_t)rec32.size();
        if ((rc = upload_array(c, halves.data(), halves.size(), &d.halves))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream)); /* the staging vectors go out of scope */
    }
    if ((rc = upload_array(c, s->attrs, (size_t)s->n_tris, &d.attrs))) return rc;
    if ((rc = upload_array(c, s->tex, (size_t)s->n_tex, &d.tex))) return rc;
    if ((rc = upload_array(c, s->texels, (size_t)s->texel_bytes, &d.texels))) return rc;
    {
        std::vector<RootRec> rr((size_t)s->n_meshes);
        for (int mi = 0; mi < s->n_meshes; mi++) {
            memset(&rr[mi], 0, sizeof(RootRec));
            for (int k = 0; k < 6; k++) rr[mi].box[k] = (float)s->meshes[mi].root_box[k];
            rr[mi].root = s->meshes[mi].root;
            rr[mi].root_count = s->meshes[mi].root_count;
            if (ext_root[(size_t)mi] >= 0) { rr[mi].root = ext_root[(size_t)mi]; rr[mi].root_count = 0; }
        }
        if ((rc = upload_array(c, rr.data(), rr.size(), &d.roots))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream)); /* rr goes out of scope */
    }
    double lut[256];
    if (s->srgb_lut) memcpy(lut, s->srgb_lut, sizeof lut);
    else for (int i = 0; i < 256; i++) lut[i] = pow(i / 255., 2.2); /* raytracer.cpp:713-715 */
    if ((rc = upload_array(c, lut, 256, &d.lut))) return rc;
    d.n_spheres = s->n_spheres;
    d.n_meshes = s->n_meshes;
    d.I = s->I;
    for (int k = 0; k < 3; k++) d.L[k] = s->L[k];
    c->light_ok = light != nullptr;
    if (light) {
        for (int k = 0; k < 3; k++) d.lightO[k] = light->O[k];
        d.lightR = light->R;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* host arrays may be released after return */
    const int depth_need = std::max(std::max(own_depth, sph_depth), max_depth + ext_depth) + 2;
    const size_t lds_fixed = sizeof(WaveLDS) + 16 + ((s->n_spheres <= RT_LDS_SPHERES_MAX) ? (size_t)s->n_spheres * sizeof(rt_sphere) : 0);
    {
        /* One more resident wave per CU is worth more than a tall stack (measured: +11 % for the eighth wave, -2 % for
         * 320 instead of 384 entries): the stack is shortened, within limits, to keep the wave's LDS inside an eighth
         * of the CU's 160 KiB. */
        const long fit = ((long)(160 * 1024 / 8) - (long)lds_fixed) / 8 - depth_need;
        d.node_soft = fit >= RT_NODE_SOFT ? RT_NODE_SOFT : (fit >= RT_NODE_SOFT_MIN ? (int)fit : RT_NODE_SOFT);
    }
    d.node_cap = d.node_soft + depth_need;
    d.overflow = c->d_overflow;
    for (int k = 0; k < 3; k++) d.bmax[k] = bmax[k];
    d.n_tris_total = (uint32_t)s->n_tris;
    d.n_ref_nodes = (uint32_t)s->n_nodes;
    c->sc = d;
    c->lds_bytes = lds_fixed + (size_t)d.node_cap * 8;
    if (c->lds_bytes > 64 * 1024) return fail(c, RT_E_INVALID, "rt_upload_scene: per-wave LDS of %zu bytes exceeds 64 KiB", c->lds_bytes);
    /* persistent waves of the render kernel that stay resident on one CU with this scene's LDS use */
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_render_kernel<false>, RT_BLOCK, c->lds_bytes) != hipSuccess || per_cu < 1) per_cu = 4;
        const int by_regs = 4 * RT_WAVES_PER_SIMD;
        c->waves_per_cu = per_cu < by_regs ? per_cu : by_regs;
    }
    c->fast_lds_bytes = rt_fast_lds_bytes(d.node_cap, s->n_spheres <= RT_LDS_SPHERES_MAX ? s->n_spheres : 0);
    c->fast_waves_per_cu = rt_fast_waves_per_cu(c->fast_lds_bytes);
    c->have_scene = true;
    rtown::stage_tick("rt_upload_scene: remaining copies, occupancy");
    return RT_OK;
}

static int check_render_args(rt_ctx *c, const rt_camera *cam, const rt_params *p, RenderArgs *ra) {
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene uploaded");
    if (!cam || !p) return fail(c, RT_E_INVALID, "camera or params is NULL");
    if (p->W <= 0 || p->H <= 0 || (int64_t)p->W * p->H > 0x7fffffffLL) return fail(c, RT_E_INVALID, "bad image size %dx%d", p->W, p->H);
    if (p->spp < 0 || p->spp_offset < 0) return fail(c, RT_E_INVALID, "negative spp or spp_offset");
    if ((int64_t)p->spp_offset + p->spp > 0x7fffffffLL) return fail(c, RT_E_INVALID, "spp_offset + spp exceeds the 31-bit sample index");
    if (p->max_rebond < 0 || p->max_rebond > 1000) return fail(c, RT_E_INVALID, "max_rebond %d out of range", p->max_rebond);
    if (!(p->flags & RT_F_POINT_LIGHT) && !c->light_ok)
        return fail(c, RT_E_INVALID, "objects[0] is not a sphere: the area-light estimator needs the light sphere there (raytracer.cpp:970, :1062)");
    int j0 = p->rect[0], i0 = p->rect[1], j1 = p->rect[2], i1 = p->rect[3];
    if (!j0 && !i0 && !j1 && !i1) { j1 = p->W; i1 = p->H; }
    if (j0 < 0 || i0 < 0 || j1 > p->W || i1 > p->H || j0 >= j1 || i0 >= i1) return fail(c, RT_E_INVALID, "bad rect [%d,%d,%d,%d]", j0, i0, j1, i1);
    ra->cam = *cam;
    ra->p = *p;
    ra->seed_mixed = 0;
    ra->cam_rays = nullptr;
    ra->cam_k0 = 0;
    ra->cam_pad = 0;
    {
        uint64_t z = p->seed + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        ra->seed_mixed = z ^ (z >> 31);
    }
    ra->j0 = j0; ra->i0 = i0; ra->j1 = j1; ra->i1 = i1;
    return RT_OK;
}

/* samples per kernel launch: long renders are cut into several launches of bounded duration */
#ifndef RT_SPP_PER_LAUNCH
#define RT_SPP_PER_LAUNCH 64
#endif
static const int kSppPerLaunch = RT_SPP_PER_LAUNCH;
#ifndef RT_SPP_PER_UNIT
#define RT_SPP_PER_UNIT 8
#endif
static const int kSppPerUnit = RT_SPP_PER_UNIT; /* samples of one pixel that a lane takes at a time (fewer on small frames) */

/* Samples of one pixel that a lane takes at a time. The unit depends on the rectangle's size only - 8 samples from
 * 262 144 pixels up (two pixels per resident lane of an MI355X: 256 CUs x 8 waves x 64 lanes), 4 / 2 / 1 on
 * rectangles of at least 131 072 / 65 536 / fewer pixels, where whole pixels would leave lanes without work -, not on
 * the device or the scene, so a render gives the same sums on every GPU and whichever way its samples are spread over
 * calls that start on unit boundaries. */
#define RT_UNIT_LANES 131072
static int unit_for(int n_slots) {
    int unit = kSppPerUnit;
    while (unit > 1 && (int64_t)n_slots * 8 < (int64_t)RT_UNIT_LANES * 2 * unit) unit >>= 1;
    return unit;
}

/* rt_render and rt_render_static: the same launches. use_static: the tile counter and the unit sums live in this
 * device's copy of the static scratch array and nothing is allocated. */
/* rt_render_static: the context that last used each GPU's static scratch array (overlap check in render_impl) */
static std::mutex g_static_mutex;
static std::map<int, rt_ctx *> g_static_owner;

static void forget_static_owner(rt_ctx *c) {
    std::lock_guard<std::mutex> lk(g_static_mutex);
    auto it = g_static_owner.find(c->device);
    if (it != g_static_owner.end() && it->second == c) g_static_owner.erase(it);
}

/* the scene as a counted launch gets it: with the choice of what is walked and counted (rt_debug_count_own) */
static DScene counted_scene(const rt_ctx *c) {
    DScene d = c->sc;
    d.count_own = c->count_own ? 1 : 0;
    return d;
}

static const size_t kRtCamRayMaxBytes = (size_t)4 << 30; /* a 2048 x 2048 launch of 64 samples would take 10.7 GB for +0.8 % */

static int render_impl(rt_ctx *c, const rt_camera *cam, const rt_params *p, double *accum, bool use_static) {
    if (!c) return RT_E_INVALID;
    RenderArgs ra;
    int rc = check_render_args(c, cam, p, &ra);
    if (rc) return rc;
    if (!accum) {
        if (!c->accum || c->accW != p->W || c->accH != p->H) return fail(c, RT_E_STATE, "own accumulator not allocated for %dx%d: call rt_accum_clear first", p->W, p->H);
        accum = c->accum;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const bool fast = (p->flags & RT_F_FAST) != 0;
    if (fast) {
        /* the single-precision kernel: the committed estimator only, short sphere lists, the filter tree */
        if (p->flags != RT_F_FAST) return fail(c, RT_E_INVALID, "RT_F_FAST cannot be combined with other flags (the single-precision kernel has the committed estimator only)");
        if (use_static) return fail(c, RT_E_INVALID, "RT_F_FAST is not available through rt_render_static (its triangle copy is allocated on first use)");
        if (c->sc.n_spheres > RT_LDS_SPHERES_MAX) return fail(c, RT_E_INVALID, "RT_F_FAST supports sphere lists of at most %d", RT_LDS_SPHERES_MAX);
        /* no filter tree because no triangle can be reached (every leaf box is flat: the reference never enters one) is fine -
         * the meshes are invisible; no filter tree because it could not be built is not */
        if (c->sc.n_meshes > 0 && c->own_failed) return fail(c, RT_E_STATE, "RT_F_FAST needs the filter tree, which could not be built for this scene");
        if (c->sc.n_meshes > 0 && c->sc.own_rec4 != nullptr && !c->d_fast_tris) {
            const size_t nb = (size_t)c->sc.n_own_tris * rt_fast_tri_bytes();
            if (hipMalloc(&c->d_fast_tris, nb ? nb : 1) != hipSuccess) { (void)hipGetLastError(); c->d_fast_tris = nullptr; return fail(c, RT_E_NOMEM, "RT_F_FAST: hipMalloc of %zu bytes failed", nb); }
            HIPCHK(c, rt_fast_convert_tris(c->sc.own_tris, c->sc.n_own_tris, c->d_fast_tris, c->stream));
        }
    }
    const int rw = ra.j1 - ra.j0, rh = ra.i1 - ra.i0;
    const int64_t tiles = (int64_t)((rw + 7) / 8) * ((rh + 7) / 8);
    const int64_t resident = (int64_t)c->n_cus * (fast ? c->fast_waves_per_cu : c->waves_per_cu);
    /* the slot counter is a 32-bit int that every resident wave may still bump by 64 once after the slots have run out */
    const int64_t slot_limit = 0x7fffffffLL - 64 * 64 * resident;
    if (tiles * RT_WAVE > slot_limit) return fail(c, RT_E_INVALID, "too many tiles");
    const int n_slots = (int)(tiles * RT_WAVE);
    const int unit = unit_for(n_slots);
    const size_t unit_bytes = (size_t)rw * rh * 3 * sizeof(double); /* sums of one unit of every pixel of the rectangle */
    int *next_slot = c->d_next_slot;
    double *unit_sums = c->unit_sums;
    int spp_per_launch = kSppPerLaunch;
    if (use_static) {
        /* The unit sums of a launch must fit the static array: a launch takes as many whole units as fit (a 1024 x 1024
         * frame: two units of 25 MB, 16 samples per launch instead of 64). The units and the order in which they are added
         * are those of rt_render, so the bits are the same. A rectangle so large that not even two of its units fit is
         * refused when the call asks for more than one unit. */
        const uint64_t units_fit = unit_bytes ? RT_STATIC_UNIT_BYTES / unit_bytes : 0;
        if (p->spp > unit && units_fit < 2)
            return fail(c, RT_E_INVALID,
                        "rt_render_static: one work unit of %d x %d pixels takes %llu bytes, the static array holds %llu: not even two units fit. "
                        "Render a smaller rect or at most %d samples per call (or use rt_render)",
                        rw, rh, (unsigned long long)unit_bytes, (unsigned long long)RT_STATIC_UNIT_BYTES, unit);
        if (p->spp > unit && units_fit * (uint64_t)unit < (uint64_t)kSppPerLaunch) spp_per_launch = (int)units_fit * unit;
        /* One array per GPU: a second context on this GPU whose static render is still running would have its tile counter
         * and unit sums overwritten. Whoever used the array last must have finished. */
        {
            std::lock_guard<std::mutex> lk(g_static_mutex);
            rt_ctx *&owner = g_static_owner[c->device];
            if (owner && owner != c) {
                const hipError_t q = hipStreamQuery(owner->stream);
                if (q == hipErrorNotReady)
                    return fail(c, RT_E_STATE, "rt_render_static: another context's rt_render_static is still running on device %d; the static "
                                               "scratch array is shared by every context of a GPU (rt_sync the other context first)", c->device);
                if (q != hipSuccess) (void)hipGetLastError();
            }
            owner = c;
        }
        next_slot = reinterpret_cast<int *>(c->static_scratch);
        unit_sums = reinterpret_cast<double *>(c->static_scratch + RT_STATIC_HEADER);
    }
    c->last_launches = 0;
    c->timed = false;
    hipEvent_t span0 = c->ev0, span1 = c->ev1;
    if (!use_static && c->ev_used < 4096) { /* a pair of this call's own, so that a caller can add up many calls without waiting between them */
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) c->ev_pool.emplace_back(a, b);
            else { if (a) (void)hipEventDestroy(a); (void)hipGetLastError(); }
        }
        if (c->ev_used < c->ev_pool.size()) { span0 = c->ev_pool[c->ev_used].first; span1 = c->ev_pool[c->ev_used].second; c->ev_used++; }
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (span0 != c->ev0) HIPCHK(c, hipEventRecord(span0, c->stream));
    /* persistent waves: as many as stay resident, each pulling 64-pixel tiles */
    for (int k = 0; k < p->spp; k += spp_per_launch) {
        const int nk = (p->spp - k < spp_per_launch) ? (p->spp - k) : spp_per_launch;
        /* Work units: a lane takes `unit` samples of a pixel at a time, so a launch ends within one short unit per lane
         * instead of one whole pixel. */
        int chunk = nk;
        const int n_units = (nk + unit - 1) / unit;
        if (n_units > 1 && (int64_t)n_slots * n_units > slot_limit) {
            /* frames of tens of megapixels: whole pixels as units */
        } else if (n_units > 1 && use_static) {
            chunk = unit; /* fits: checked above */
        } else if (n_units > 1) {
            const size_t need = (size_t)n_units * unit_bytes;
            if (c->unit_sums_bytes < need) { /* kept for the following launches and calls */
                HIPCHK(c, hipStreamSynchronize(c->stream)); /* earlier launches may still use the old buffer */
                if (c->unit_sums) (void)hipFree(c->unit_sums);
                c->unit_sums = nullptr;
                c->unit_sums_bytes = 0;
                if (hipMalloc((void **)&c->unit_sums, need) == hipSuccess) c->unit_sums_bytes = need;
                else (void)hipGetLastError(); /* no room: whole pixels as units, slower at the end of the launch */
            }
            unit_sums = c->unit_sums;
            if (c->unit_sums_bytes >= need) chunk = unit;
        }
        /* The camera rays of the launch ahead of it, for the committed estimator (no mode flag): kRtCamRayMaxBytes bounds the
         * buffer (a 1024 x 1024 launch of 64 samples: 2.7 GB); larger launches, and contexts that cannot have the memory,
         * compute the rays in the render kernel as the other modes do. The bits are the same either way. */
        bool pregen = false;
        if (!fast && !use_static && !c->no_cam_rays && p->flags == 0) {
            const size_t need = (size_t)nk * rw * rh * RT_CAM_RAY_DOUBLES * sizeof(double);
            if (need <= kRtCamRayMaxBytes && c->cam_rays_bytes < need && !(c->cam_rays_refused && need >= c->cam_rays_refused_bytes)) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (c->cam_rays) (void)hipFree(c->cam_rays);
                c->cam_rays = nullptr;
                c->cam_rays_bytes = 0;
                if (hipMalloc((void **)&c->cam_rays, need) == hipSuccess) c->cam_rays_bytes = need;
                else { (void)hipGetLastError(); c->cam_rays_refused = true; c->cam_rays_refused_bytes = need; }
            }
            pregen = need <= kRtCamRayMaxBytes && c->cam_rays_bytes >= need;
        }
        ra.cam_rays = pregen ? c->cam_rays : nullptr;
        ra.cam_k0 = p->spp_offset + k;
        if (pregen) {
            const int64_t nthr = (int64_t)nk * rw * rh;
            hipLaunchKernelGGL(rt_camera_rays_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, c->stream, ra, p->spp_offset + k, nk, c->cam_rays);
            HIPCHK(c, hipGetLastError());
        }
        const int64_t tiles_all = tiles * (chunk < nk ? n_units : 1); /* 64-slot requests that the launch hands out */
        const unsigned pgrid = (unsigned)(tiles_all < resident ? tiles_all : resident);
        HIPCHK(c, hipMemsetAsync(next_slot, 0, sizeof(int), c->stream));
        hipEvent_t k1 = nullptr;
        if (!use_static && c->kev_used < 4096) {
            if (c->kev_used == c->kev_pool.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) c->kev_pool.emplace_back(a, b);
                else { if (a) (void)hipEventDestroy(a); (void)hipGetLastError(); }
            }
            if (c->kev_used < c->kev_pool.size()) {
                HIPCHK(c, hipEventRecord(c->kev_pool[c->kev_used].first, c->stream));
                k1 = c->kev_pool[c->kev_used].second;
                c->kev_used++;
            }
        }
        if (fast)
            HIPCHK(c, rt_fast_launch(pgrid, c->fast_lds_bytes, c->stream, c->sc, c->d_fast_tris, ra, p->spp_offset + k, nk, chunk, accum, unit_sums, next_slot, n_slots));
        else if (c->sc.sph_rec4 != nullptr && (p->flags & RT_F_COUNTERS))
            hipLaunchKernelGGL(rt_render_spheres_kernel<true>, dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, counted_scene(c), ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        else if (c->sc.sph_rec4 != nullptr && pregen)
            hipLaunchKernelGGL((rt_render_spheres_kernel<false, true>), dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        else if (c->sc.sph_rec4 != nullptr)
            hipLaunchKernelGGL(rt_render_spheres_kernel<false>, dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        else if (pregen)
            hipLaunchKernelGGL((rt_render_kernel<false, true, true>), dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        else if (p->flags == 0)
            hipLaunchKernelGGL((rt_render_kernel<false, true>), dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        else if (p->flags & RT_F_COUNTERS)
            hipLaunchKernelGGL(rt_render_kernel<true>, dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, counted_scene(c), ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        else
            hipLaunchKernelGGL(rt_render_kernel<false>, dim3(pgrid), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, ra, p->spp_offset + k, nk, chunk, accum, unit_sums, c->d_counters, next_slot, n_slots);
        HIPCHK(c, hipGetLastError());
        if (k1) HIPCHK(c, hipEventRecord(k1, c->stream));
        if (chunk < nk) {
            const int64_t nthr = (int64_t)rw * rh * 3;
            hipLaunchKernelGGL(rt_units_add_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, c->stream, accum, unit_sums, n_units, p->W, p->H,
                               ra.j0, ra.i0, rw, rh);
            HIPCHK(c, hipGetLastError());
        }
        c->last_launches++;
    }
    if (span1 != c->ev1) HIPCHK(c, hipEventRecord(span1, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    if (span1 != c->ev1) c->launches_since_reset += c->last_launches;
    return RT_OK;
}

int rt_render(rt_ctx *c, const rt_camera *cam, const rt_params *p, double *accum) { return render_impl(c, cam, p, accum, false); }

int rt_render_static(rt_ctx *c, const rt_camera *cam, const rt_params *p, double *accum) { return render_impl(c, cam, p, accum, true); }

int rt_sync(rt_ctx *c) {
    if (!c) return RT_E_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_overflow(c);
}

int rt_last_render_ms(rt_ctx *c, double *ms, int32_t *n_launches) {
    if (!c || !ms) return RT_E_INVALID;
    if (!c->timed) return fail(c, RT_E_STATE, "no render recorded");
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = f;
    if (n_launches) *n_launches = c->last_launches;
    return RT_OK;
}

int rt_render_time_ms(rt_ctx *c, double *ms_total, int32_t *n_launches, int32_t reset) {
    if (!c) return RT_E_INVALID;
    double sum = 0;
    for (size_t k = 0; k < c->ev_used; k++) {
        HIPCHK(c, hipEventSynchronize(c->ev_pool[k].second));
        float f = 0;
        HIPCHK(c, hipEventElapsedTime(&f, c->ev_pool[k].first, c->ev_pool[k].second));
        sum += f;
    }
    if (ms_total) *ms_total = sum;
    if (n_launches) *n_launches = c->launches_since_reset;
    if (reset) { c->ev_used = 0; c->launches_since_reset = 0; c->kev_used = 0; }
    return RT_OK;
}

/* the render kernels alone of the launches since the last reset of rt_render_time_ms (which resets this too) */
int rt_debug_kernel_time_ms(rt_ctx *c, double *ms_total, int32_t *n_launches) {
    if (!c || !ms_total) return RT_E_INVALID;
    double sum = 0;
    for (size_t k = 0; k < c->kev_used; k++) {
        HIPCHK(c, hipEventSynchronize(c->kev_pool[k].second));
        float f = 0;
        HIPCHK(c, hipEventElapsedTime(&f, c->kev_pool[k].first, c->kev_pool[k].second));
        sum += f;
    }
    *ms_total = sum;
    if (n_launches) *n_launches = (int32_t)c->kev_used;
    return RT_OK;
}

int rt_accum_clear(rt_ctx *c, int32_t W, int32_t H) {
    if (!c) return RT_E_INVALID;
    if (W <= 0 || H <= 0) return fail(c, RT_E_INVALID, "bad accumulator size");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->accum || c->accW != W || c->accH != H) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->accum) (void)hipFree(c->accum);
        c->accum = nullptr;
        if (hipMalloc((void **)&c->accum, (size_t)W * H * 3 * sizeof(double)) != hipSuccess) return fail(c, RT_E_NOMEM, "accumulator allocation failed");
        c->accW = W;
        c->accH = H;
    }
    HIPCHK(c, hipMemsetAsync(c->accum, 0, (size_t)W * H * 3 * sizeof(double), c->stream));
    return RT_OK;
}

double *rt_accum_device_ptr(rt_ctx *c) { return c ? c->accum : nullptr; }

int rt_accum_download(rt_ctx *c, double *host_out, int64_t n) {
    if (!c || !host_out) return RT_E_INVALID;
    if (!c->accum || n != (int64_t)c->accW * c->accH * 3) return fail(c, RT_E_INVALID, "rt_accum_download: size mismatch");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(host_out, c->accum, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_overflow(c);
}

int rt_accum_upload(rt_ctx *c, const double *host_in, int64_t n) {
    if (!c || !host_in) return RT_E_INVALID;
    if (!c->accum || n != (int64_t)c->accW * c->accH * 3) return fail(c, RT_E_INVALID, "rt_accum_upload: size mismatch");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->accum, host_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* restores the calling thread's current device when a function that switches it returns, on every path */
namespace {
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
} // namespace

int rt_accum_add(rt_ctx *dst, rt_ctx *src) {
    if (!dst || !src) return RT_E_INVALID;
    if (dst == src) return fail(dst, RT_E_INVALID, "rt_accum_add: a context cannot be added to itself");
    if (!dst->accum || !src->accum || dst->accW != src->accW || dst->accH != src->accH)
        return fail(dst, RT_E_STATE, "rt_accum_add: both contexts need their own accumulator of the same size (rt_accum_clear)");
    const int64_t n = (int64_t)dst->accW * dst->accH * 3;
    const size_t nb = (size_t)n * sizeof(double);
    DeviceGuard guard; /* the caller's current device is the same after the call, whatever happens */
    /* the source's renders must have finished before its accumulator is read from another stream */
    HIPCHK(dst, hipSetDevice(src->device));
    HIPCHK(dst, hipStreamSynchronize(src->stream));
    HIPCHK(dst, hipSetDevice(dst->device));
    if (dst->d_peer_bytes < nb) {
        HIPCHK(dst, hipStreamSynchronize(dst->stream));
        if (dst->d_peer) (void)hipFree(dst->d_peer);
        dst->d_peer = nullptr;
        dst->d_peer_bytes = 0;
        if (hipMalloc((void **)&dst->d_peer, nb) != hipSuccess) { (void)hipGetLastError(); return fail(dst, RT_E_NOMEM, "rt_accum_add: staging buffer allocation failed"); }
        dst->d_peer_bytes = nb;
    }
    /* How the source's sums get into dst's staging buffer: a device-to-device copy when the contexts share a GPU; over xGMI
     * (hipMemcpyPeerAsync) when they sit on two GPUs that can reach each other; otherwise - the runtime refuses peer access
     * (another root complex, a restricted container), or the peer copy itself fails - through a pinned host buffer. */
    bool direct = true;
    if (dst->force_host_bounce) direct = false;
    else if (src->device != dst->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, src->device) != hipSuccess) { can = 0; (void)hipGetLastError(); }
        direct = can != 0;
    }
    if (direct) {
        const hipError_t e = hipMemcpyPeerAsync(dst->d_peer, dst->device, src->accum, src->device, nb, dst->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); direct = false; } /* nothing was queued: take the other way */
    }
    if (!direct) {
        if (dst->h_bounce_bytes < nb) {
            if (dst->h_bounce) (void)hipHostFree(dst->h_bounce);
            dst->h_bounce = nullptr;
            dst->h_bounce_bytes = 0;
            if (hipHostMalloc((void **)&dst->h_bounce, nb, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(dst, RT_E_NOMEM, "rt_accum_add: pinned host buffer allocation failed"); }
            dst->h_bounce_bytes = nb;
        }
        HIPCHK(dst, hipSetDevice(src->device));
        HIPCHK(dst, hipMemcpyAsync(dst->h_bounce, src->accum, nb, hipMemcpyDeviceToHost, src->stream));
        HIPCHK(dst, hipStreamSynchronize(src->stream));
        HIPCHK(dst, hipSetDevice(dst->device));
        HIPCHK(dst, hipMemcpyAsync(dst->d_peer, dst->h_bounce, nb, hipMemcpyHostToDevice, dst->stream));
    }
    hipLaunchKernelGGL(rt_accum_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dst->stream, dst->accum, dst->d_peer, n);
    HIPCHK(dst, hipGetLastError());
    HIPCHK(dst, hipStreamSynchronize(dst->stream)); /* the caller may clear or destroy src as soon as this returns */
    return RT_OK;
}

/* not part of the public header: resident waves per CU and LDS bytes per wave of the single-precision kernel with the current scene */
int rt_debug_fast_occupancy(rt_ctx *c, int *waves_per_cu, int *lds_bytes) {
    if (!c || !waves_per_cu || !lds_bytes || !c->have_scene) return RT_E_INVALID;
    *waves_per_cu = c->fast_waves_per_cu;
    *lds_bytes = (int)c->fast_lds_bytes;
    return RT_OK;
}

/* not part of the public header: hipMemGetInfo on the context's device (tests: a call that must not allocate leaves the
 * free bytes unchanged) */
int rt_debug_mem_info(rt_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!c || !free_bytes || !total_bytes) return RT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    *free_bytes = (uint64_t)f;
    *total_bytes = (uint64_t)t;
    return RT_OK;
}

/* not part of the public header: rt_accum_add into this context goes through the pinned host buffer even where a direct
 * copy is possible (tests: the fallback path on a machine with one GPU) */
int rt_debug_force_host_bounce(rt_ctx *c, int on) {
    if (!c) return RT_E_INVALID;
    c->force_host_bounce = on != 0;
    return RT_OK;
}

int rt_resolve(rt_ctx *c, const double *accum, int32_t W, int32_t H, double spp_total, uint8_t *rgb8_host) {
    if (!c || !rgb8_host) return RT_E_INVALID;
    if (W <= 0 || H <= 0) return fail(c, RT_E_INVALID, "bad size");
    if (!accum) {
        if (!c->accum || c->accW != W || c->accH != H) return fail(c, RT_E_STATE, "own accumulator not allocated for %dx%d", W, H);
        accum = c->accum;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = (size_t)W * H * 3;
    if (c->d_rgb_bytes < nb) { /* kept for the following calls */
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_rgb) (void)hipFree(c->d_rgb);
        c->d_rgb = nullptr;
        c->d_rgb_bytes = 0;
        if (hipMalloc((void **)&c->d_rgb, nb) != hipSuccess) return fail(c, RT_E_NOMEM, "resolve buffer allocation failed");
        c->d_rgb_bytes = nb;
    }
    uint8_t *d = c->d_rgb;
    const int64_t n = (int64_t)W * H;
    hipLaunchKernelGGL(rt_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, accum, W, H, spp_total, c->d_tone, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgb8_host, d, nb, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, RT_E_DEVICE, "rt_resolve: %s", hipGetErrorString(e));
    return check_overflow(c);
}

int rt_intersect_rays(rt_ctx *c, const double *rays, int64_t n, int32_t *obj_id, int32_t *tri_id, double *t, double *PN, double *albedo) {
    if (!c) return RT_E_INVALID;
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene uploaded");
    if (n < 0 || (n && !rays)) return fail(c, RT_E_INVALID, "bad ray array");
    if (n == 0) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    /* one device block: rays | obj | tri | t | PN | albedo */
    const size_t b_rays = (size_t)n * 6 * 8, b_i = (size_t)n * 4, b_t = (size_t)n * 8, b_pn = (size_t)n * 6 * 8, b_al = (size_t)n * 3 * 8;
    unsigned char *d = nullptr;
    if (hipMalloc((void **)&d, b_rays + b_t + b_pn + b_al + 2 * b_i) != hipSuccess) return fail(c, RT_E_NOMEM, "probe buffer allocation failed");
    double *d_rays = (double *)d, *d_t = (double *)(d + b_rays), *d_pn = (double *)(d + b_rays + b_t), *d_al = (double *)(d + b_rays + b_t + b_pn);
    int32_t *d_obj = (int32_t *)(d + b_rays + b_t + b_pn + b_al), *d_tri = d_obj + n;
    hipError_t e = hipMemcpyAsync(d_rays, rays, b_rays, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_intersect_kernel, dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, d_rays, n, c->probe_prune ? 1 : 0, d_obj, d_tri, d_t, d_pn, d_al);
        e = hipGetLastError();
    }
    if (e == hipSuccess && obj_id) e = hipMemcpyAsync(obj_id, d_obj, b_i, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && tri_id) e = hipMemcpyAsync(tri_id, d_tri, b_i, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && t) e = hipMemcpyAsync(t, d_t, b_t, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && PN) e = hipMemcpyAsync(PN, d_pn, b_pn, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && albedo) e = hipMemcpyAsync(albedo, d_al, b_al, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, RT_E_DEVICE, "rt_intersect_rays: %s", hipGetErrorString(e));
    return check_overflow(c);
}

int rt_sample_radiance(rt_ctx *c, const rt_camera *cam, const rt_params *p, const int32_t *ijk, int64_t n, double *out) {
    return rt_sample_trace(c, cam, p, ijk, n, out, nullptr);
}

int rt_sample_trace(rt_ctx *c, const rt_camera *cam, const rt_params *p, const int32_t *ijk, int64_t n, double *out, int32_t *trace) {
    if (!c) return RT_E_INVALID;
    RenderArgs ra;
    int rc = check_render_args(c, cam, p, &ra);
    if (rc) return rc;
    if (n < 0 || (n && (!ijk || !out))) return fail(c, RT_E_INVALID, "bad sample list");
    if (n == 0) return RT_OK;
    for (int64_t q = 0; q < n; q++)
        if (ijk[3 * q] < 0 || ijk[3 * q] >= p->H || ijk[3 * q + 1] < 0 || ijk[3 * q + 1] >= p->W || ijk[3 * q + 2] < 0)
            return fail(c, RT_E_INVALID, "sample %lld outside the image", (long long)q);
    HIPCHK(c, hipSetDevice(c->device));
    unsigned char *d = nullptr;
    const size_t b_in = (size_t)n * 3 * 4, b_out = (size_t)n * 3 * 8, b_tr = trace ? (size_t)n * RT_TRACE_LEN * 4 * 4 : 0;
    if (hipMalloc(&d, b_out + b_in + b_tr) != hipSuccess) return fail(c, RT_E_NOMEM, "probe buffer allocation failed");
    double *d_out = (double *)d;
    int32_t *d_in = (int32_t *)(d + b_out);
    int32_t *d_tr = trace ? (int32_t *)(d + b_out + b_in) : nullptr;
    hipError_t e = hipMemcpyAsync(d_in, ijk, b_in, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_sample_kernel, dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), c->lds_bytes, c->stream, c->sc, ra, d_in, n, d_out, d_tr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, b_out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && trace) e = hipMemcpyAsync(trace, d_tr, b_tr, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, RT_E_DEVICE, "rt_sample_trace: %s", hipGetErrorString(e));
    return check_overflow(c);
}

int rt_get_counters(rt_ctx *c, rt_counters *out) {
    if (!c || !out) return RT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_counters, sizeof(rt_counters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* not part of the public header: what the renders with RT_F_COUNTERS walk and count from now on. 0 (default): the
 * reference's traversal on the reference's trees - rt_counters then equals the CPU checker's counters. 1: what a production
 * render walks - the filter tree with its single-precision box tests, then the reference's trees (with their extension
 * records) for the rays whose winner was not confirmed; rt_counters.box_tests / nodes / tris then count that work, and
 * word 22 of rt_debug_stats the double-precision box tests (undecided single-precision tests, winner confirmations). */
/* test hook: 0 makes the render kernels compute their camera rays themselves (what rt_render_static and the other modes do) */
int rt_debug_camera_rays(rt_ctx *c, int on) {
    if (!c) return RT_E_INVALID;
    c->no_cam_rays = !on;
    return RT_OK;
}

int rt_debug_count_own(rt_ctx *c, int on) {
    if (!c) return RT_E_INVALID;
    c->count_own = on != 0;
    return RT_OK;
}

/* not part of the public header: 24 diagnostic words of the counted kernel variant (cycle stamps per phase, pool occupancy,
 * 8 free slots for experiments - [6] of them: rays that went round again on the reference's trees -, exact sphere tests,
 * sphere-tree box tests, double-precision box tests, one spare), for tools/ and bench.py's report only */
int rt_debug_stats(rt_ctx *c, uint64_t *out24) {
    uint64_t *out22 = out24;
    if (!c || !out22) return RT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out22, reinterpret_cast<uint64_t *>(c->d_counters) + 8, 24 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* not part of the public header: workgroups of the render kernel that fit one CU with the current scene's LDS use (tools/ only) */
int rt_debug_occupancy(rt_ctx *c, int *blocks_per_cu, int *lds_bytes) {
    if (!c || !blocks_per_cu || !lds_bytes || !c->have_scene) return RT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, rt_render_kernel<false>, RT_BLOCK, c->lds_bytes));
    *lds_bytes = (int)c->lds_bytes;
    return RT_OK;
}

/* not part of the public header: device bytes of the uploaded scene, traversal records (reference + extension), filter-tree
 * records, LDS stack entries per wave (tools/ and bench.py's report only) */
int rt_debug_scene_sizes(rt_ctx *c, uint64_t *out4) {
    if (!c || !out4 || !c->have_scene) return RT_E_INVALID;
    out4[0] = c->scene_bytes;
    out4[1] = c->sc.n_nodes_total;
    out4[2] = c->sc.n_own_nodes;
    out4[3] = (uint64_t)c->sc.node_cap;
    return RT_OK;
}

/* not part of the public header: FNV-1a hash over the filter tree's records and its triangle array as they sit on the device
 * (tests/ only: the tree must not depend on the number of host threads that built it); 0 when the scene has no filter tree */
int rt_debug_filter_tree_hash(rt_ctx *c, uint64_t *out) {
    if (!c || !out || !c->have_scene) return RT_E_INVALID;
    *out = 0;
    if (!c->sc.own_rec4) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<unsigned char> buf((size_t)c->sc.n_own_nodes * sizeof(NodeRec4) + (size_t)c->sc.n_own_tris * sizeof(rt_tri));
    HIPCHK(c, hipMemcpyAsync(buf.data(), c->sc.own_rec4, (size_t)c->sc.n_own_nodes * sizeof(NodeRec4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf.data() + (size_t)c->sc.n_own_nodes * sizeof(NodeRec4), c->sc.own_tris, (size_t)c->sc.n_own_tris * sizeof(rt_tri), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t h = 1469598103934665603ull;
    for (unsigned char b : buf) { h ^= b; h *= 1099511628211ull; }
    *out = h;
    return RT_OK;
}

/* not part of the public header: rt_intersect_rays with / without distance pruning (tests/ only) */
int rt_debug_probe_prune(rt_ctx *c, int on) {
    if (!c) return RT_E_INVALID;
    c->probe_prune = on != 0;
    return RT_OK;
}

/* not part of the public header: the filter trees of the following uploads get leaves of 2 or 4 triangles whatever the scene's
 * size (0: back to the rule), so that tests reach both forms of the triangle step on small scenes (tests/ only) */
int rt_debug_own_leaf(rt_ctx *c, int leaf) {
    if (!c || (leaf != 0 && leaf != 2 && leaf != 4) || leaf > RT_OWN_LEAF) return RT_E_INVALID;
    c->force_own_leaf = leaf;
    return RT_OK;
}

/* not part of the public header: the 256 tone-map thresholds as the host computes them (no GPU needed; tests/ only) */
int rt_debug_tone_table(double *out256) {
    if (!out256) return RT_E_INVALID;
    tone_thresholds(out256);
    return RT_OK;
}

/* not part of the public header: runs the wave-scan self-test on 64 ints (tests/ only) */
int rt_debug_wave_scan(rt_ctx *c, const int32_t *in64, int32_t *add64, int32_t *max64) {
    if (!c || !in64 || !add64 || !max64) return RT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int *d = nullptr;
    if (hipMalloc((void **)&d, 3 * 64 * sizeof(int)) != hipSuccess) return fail(c, RT_E_NOMEM, "allocation failed");
    hipError_t e = hipMemcpyAsync(d, in64, 64 * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_scan_test_kernel, dim3(1), dim3(RT_BLOCK), 0, c->stream, d, d + 64, d + 128);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(add64, d + 64, 64 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(max64, d + 128, 64 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, RT_E_DEVICE, "rt_debug_wave_scan: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_reset_counters(rt_ctx *c) {
    if (!c) return RT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, RT_COUNTER_WORDS * 8, c->stream));
    return RT_OK;
}

} /* extern "C" */

#include "rt_bvh_build.hip.h"
