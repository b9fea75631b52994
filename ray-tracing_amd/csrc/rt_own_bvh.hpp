/*
 * rt_own_bvh.hpp - host-side builder of the filter tree.
 *
 * The reference's BVH (raytracer.cpp:244-291) decides which triangles a ray may hit; its quality is poor
 * (one tree per mesh, median cuts, leaves of up to hundreds of triangles). The traversal therefore runs on
 * a second tree built here over ALL triangles of the scene - binned surface-area cuts, leaves of at most
 * four triangles, triangles copied into this tree's own order - and uses it only to find candidate hits:
 * a ray is kept out of a box just when it certainly misses it. Whether the reference would have reached
 * the winning triangle is then checked on the reference's own leaf box (rt_device.hip.h: rays_end).
 */
#pragma once
#include <math.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_device.hip.h"

namespace rtown {

struct Bounds {
    double lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) { lo[a] = HUGE_VAL; hi[a] = -HUGE_VAL; }
    }
    void grow(const Bounds &b) {
        for (int a = 0; a < 3; a++) {
            if (b.lo[a] < lo[a]) lo[a] = b.lo[a];
            if (b.hi[a] > hi[a]) hi[a] = b.hi[a];
        }
    }
    double half_area() const {
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

struct Built {
    std::vector<rtd::NodeRec32> nodes; /* binary tree: record 0 is the root; children have larger indices */
    std::vector<rtd::NodeRec4> wide;   /* the same tree with four children per record (collapse() below) */
    int wide_depth = 0;
    std::vector<int32_t> order;        /* position in this tree's triangle array -> position in the caller's array */
    int depth = 0;
    bool ok = false;                   /* false: the scene has a non-finite coordinate or no triangle; no tree */
};

inline void collapse(Built &b);

static inline void box_to_float(const Bounds &b, float *box) { /* rounded outwards */
    for (int a = 0; a < 3; a++) {
        box[a] = nextafterf((float)b.lo[a], -HUGE_VALF);
        box[3 + a] = nextafterf((float)b.hi[a], HUGE_VALF);
    }
}

#ifndef RT_OWN_BINS
#define RT_OWN_BINS 16
#endif
static const int kBins = RT_OWN_BINS;
#ifndef RT_OWN_LEAF
#define RT_OWN_LEAF 4
#endif
static const int kLeaf = RT_OWN_LEAF; /* at most 8: the leaf count has three bits in a child code */
static const int kSahDepth = 40; /* below this depth ranges are halved by count, which bounds the depth */

/* worker threads for the host-side builds: OMP_NUM_THREADS when it is set (job schedulers size it to the process's share
 * of the machine), else the hardware's count, at most 64 */
inline int host_threads() {
    int t = 0;
    if (const char *e = getenv("OMP_NUM_THREADS")) t = atoi(e);
    if (t <= 0) t = (int)std::thread::hardware_concurrency();
    if (t <= 0) t = 1;
    return t > 64 ? 64 : t;
}

/* fn(part, begin, end) on `parts` contiguous pieces of [0, n), one thread each (the caller's thread takes piece 0) */
template <typename F>
inline void parallel_parts(int64_t n, int parts, F fn) {
    if (parts <= 1 || n < 2) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int p = 1; p < parts; p++) th.emplace_back([=] { fn(p, n * p / parts, n * (p + 1) / parts); });
    fn(0, (int64_t)0, n / parts);
    for (auto &t : th) t.join();
}

struct Builder {
    const std::vector<Bounds> &tb;
    const std::vector<double> &cen;
    std::vector<int32_t> &idx;

    struct Job {
        int64_t node; /* record that holds this range as child `side`, or -1 for the root range */
        int side;
        int64_t first, count;
        int depth;
    };

    Bounds range_bounds(int64_t first, int64_t count) const {
        Bounds b;
        b.reset();
        for (int64_t k = first; k < first + count; k++) b.grow(tb[(size_t)idx[(size_t)k]]);
        return b;
    }

    /* Cuts [first, first + count) in two: the binned surface-area cut (kBins bins per axis over the centroids' extent),
     * or - all centroids equal, or too deep - the median along the longest centroid axis. Reorders idx, returns the size
     * of the left part and both parts' bounds. threads > 1: the passes over the range are shared out (the top of the
     * tree, where one range is most of the scene). The result does not depend on the thread count. */
    int64_t cut(int64_t first, int64_t count, int depth, int threads, Bounds &bl, Bounds &br) {
        if (count < (1 << 16)) threads = 1;
        int64_t nl = count / 2;
        bool sorted = false;
        Bounds cb;
        cb.reset();
        {
            std::vector<Bounds> part((size_t)threads);
            parallel_parts(count, threads, [&](int p, int64_t b, int64_t e) {
                Bounds x;
                x.reset();
                for (int64_t k = first + b; k < first + e; k++)
                    for (int a = 0; a < 3; a++) {
                        const double c = cen[(size_t)idx[(size_t)k] * 3 + a];
                        if (c < x.lo[a]) x.lo[a] = c;
                        if (c > x.hi[a]) x.hi[a] = c;
                    }
                part[(size_t)p] = x;
            });
            for (int p = 0; p < threads; p++) cb.grow(part[(size_t)p]);
        }
        if (depth <= kSahDepth) {
            struct Bins { Bounds bb[3][kBins]; int64_t bc[3][kBins]; };
            std::vector<Bins> part((size_t)threads);
            double scale[3];
            for (int a = 0; a < 3; a++) scale[a] = (cb.hi[a] - cb.lo[a] > 0) ? kBins / (cb.hi[a] - cb.lo[a]) : 0;
            parallel_parts(count, threads, [&](int p, int64_t b, int64_t e) {
                Bins &B = part[(size_t)p];
                for (int a = 0; a < 3; a++)
                    for (int q = 0; q < kBins; q++) { B.bb[a][q].reset(); B.bc[a][q] = 0; }
                for (int64_t k = first + b; k < first + e; k++) {
                    const int32_t t = idx[(size_t)k];
                    for (int a = 0; a < 3; a++) {
                        if (!(scale[a] > 0)) continue;
                        int q = (int)((cen[(size_t)t * 3 + a] - cb.lo[a]) * scale[a]);
                        if (q >= kBins) q = kBins - 1;
                        if (q < 0) q = 0;
                        B.bb[a][q].grow(tb[(size_t)t]);
                        B.bc[a][q]++;
                    }
                }
            });
            Bins &S = part[0];
            for (int p = 1; p < threads; p++)
                for (int a = 0; a < 3; a++)
                    for (int q = 0; q < kBins; q++) { S.bb[a][q].grow(part[(size_t)p].bb[a][q]); S.bc[a][q] += part[(size_t)p].bc[a][q]; }
            double best = HUGE_VAL;
            int best_axis = -1, best_bin = 0;
            for (int a = 0; a < 3; a++) {
                if (!(scale[a] > 0)) continue;
                double ra[kBins];
                Bounds acc;
                acc.reset();
                int64_t cnt = 0;
                for (int q = kBins - 1; q >= 1; q--) { acc.grow(S.bb[a][q]); cnt += S.bc[a][q]; ra[q] = cnt ? acc.half_area() * (double)cnt : 0; }
                acc.reset();
                cnt = 0;
                for (int q = 0; q < kBins - 1; q++) {
                    acc.grow(S.bb[a][q]);
                    cnt += S.bc[a][q];
                    if (cnt == 0 || cnt == count) continue;
                    const double cost = acc.half_area() * (double)cnt + ra[q + 1];
                    if (cost < best) { best = cost; best_axis = a; best_bin = q; }
                }
            }
            if (best_axis >= 0) {
                const double sc = scale[best_axis], lo = cb.lo[best_axis];
                int32_t *beg = idx.data() + first, *end = beg + count;
                int32_t *mid = std::partition(beg, end, [&](int32_t t) {
                    int q = (int)((cen[(size_t)t * 3 + best_axis] - lo) * sc);
                    if (q >= kBins) q = kBins - 1;
                    if (q < 0) q = 0;
                    return q <= best_bin;
                });
                nl = mid - beg;
                sorted = nl > 0 && nl < count;
                if (sorted) { /* the parts' bounds are the unions of their bins */
                    bl.reset();
                    br.reset();
                    for (int q = 0; q < kBins; q++) (q <= best_bin ? bl : br).grow(S.bb[best_axis][q]);
                    return nl;
                }
            }
        }
        /* all centroids coincide, or too deep: halve the range along its longest centroid axis */
        nl = count / 2;
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
        std::nth_element(idx.begin() + first, idx.begin() + first + nl, idx.begin() + first + count,
                         [&](int32_t x, int32_t y) { return cen[(size_t)x * 3 + axis] < cen[(size_t)y * 3 + axis]; });
        bl = range_bounds(first, nl);
        br = range_bounds(first + nl, count - nl);
        return nl;
    }

    /* Breadth-first build of the ranges in `jobs` into `nodes` (a child's index is larger than its parent's). Ranges of at
     * most `defer` triangles are not cut here but handed back in `deferred` (0: none are). */
    void run(std::vector<Job> &jobs, std::vector<rtd::NodeRec32> &nodes, int &max_depth, int threads, int64_t defer, std::vector<Job> *deferred) {
        for (size_t q = 0; q < jobs.size(); q++) {
            const Job jb = jobs[q];
            if (deferred && jb.count <= defer) { deferred->push_back(jb); continue; }
            Bounds bl, br;
            const int64_t nl = cut(jb.first, jb.count, jb.depth, threads, bl, br), nr = jb.count - nl;
            const int64_t id = (int64_t)nodes.size();
            rtd::NodeRec32 r;
            box_to_float(bl, r.boxL);
            box_to_float(br, r.boxR);
            r.L = (int32_t)jb.first; r.nL = (int32_t)nl;
            r.R = (int32_t)(jb.first + nl); r.nR = (int32_t)nr;
            nodes.push_back(r);
            if (jb.node >= 0) {
                rtd::NodeRec32 &pr = nodes[(size_t)jb.node];
                if (jb.side == 0) { pr.L = (int32_t)id; pr.nL = 0; } else { pr.R = (int32_t)id; pr.nR = 0; }
            }
            if (jb.depth > max_depth) max_depth = jb.depth;
            if (nl > kLeaf) jobs.push_back(Job{id, 0, jb.first, nl, jb.depth + 1});
            if (nr > kLeaf) jobs.push_back(Job{id, 1, jb.first + nl, nr, jb.depth + 1});
        }
    }
};

inline Built build_boxes(const std::vector<Bounds> &tb);

/* the tree over triangles */
inline Built build(const rt_tri *tris, int64_t n) {
    if (n <= 0) return Built();
    const int threads = host_threads();
    std::vector<Bounds> tb((size_t)n);
    std::vector<char> bad((size_t)threads, 0);
    parallel_parts(n, n >= (1 << 16) ? threads : 1, [&](int p, int64_t b0, int64_t e0) {
        for (int64_t k = b0; k < e0; k++) {
            Bounds &b = tb[(size_t)k];
            b.reset();
            for (int a = 0; a < 3; a++) {
                const double v[3] = {tris[k].A[a], tris[k].A[a] + tris[k].e1[a], tris[k].A[a] + tris[k].e2[a]};
                for (int q = 0; q < 3; q++) {
                    if (!std::isfinite(v[q])) bad[(size_t)p] = 1;
                    if (v[q] < b.lo[a]) b.lo[a] = v[q];
                    if (v[q] > b.hi[a]) b.hi[a] = v[q];
                }
            }
        }
    });
    for (char x : bad)
        if (x) return Built();
    return build_boxes(tb);
}

/* the tree over any primitives, given their (finite) boxes: triangles, or the spheres of a long sphere list */
inline Built build_boxes(const std::vector<Bounds> &tb) {
    Built out;
    const int64_t n = (int64_t)tb.size();
    if (n <= 0) return out;
    const int threads = host_threads();
    std::vector<double> cen((size_t)n * 3);
    std::vector<int32_t> idx((size_t)n);
    parallel_parts(n, n >= (1 << 16) ? threads : 1, [&](int, int64_t b0, int64_t e0) {
        for (int64_t k = b0; k < e0; k++) {
            for (int a = 0; a < 3; a++) cen[(size_t)k * 3 + a] = 0.5 * (tb[(size_t)k].lo[a] + tb[(size_t)k].hi[a]);
            idx[(size_t)k] = (int32_t)k;
        }
    });
    Builder B{tb, cen, idx};
    if (n <= kLeaf) { /* a tree needs one record: both children are leaves (the right one may be empty) */
        rtd::NodeRec32 r;
        const int64_t nl = (n + 1) / 2;
        box_to_float(B.range_bounds(0, nl), r.boxL);
        r.L = 0; r.nL = (int32_t)nl;
        if (n - nl > 0) { box_to_float(B.range_bounds(nl, n - nl), r.boxR); r.R = (int32_t)nl; r.nR = (int32_t)(n - nl); }
        else { for (int a = 0; a < 3; a++) { r.boxR[a] = HUGE_VALF; r.boxR[3 + a] = -HUGE_VALF; } r.R = 0; r.nR = 1; } /* inverted box: certainly missed */
        out.nodes.push_back(r);
        out.order = idx;
        out.depth = 1;
        out.ok = true;
        collapse(out);
        return out;
    }
    /* The top of the tree with all threads inside each cut; ranges below n / (8 threads) triangles are whole subtrees that
     * the threads then build one each, into arrays of their own, appended in the order of the ranges (so the tree is
     * the same whatever the thread count and timing). */
    std::vector<Builder::Job> jobs, sub;
    jobs.push_back(Builder::Job{-1, 0, 0, n, 1});
    const int64_t defer = (threads > 1 && n >= (1 << 16)) ? n / (8 * (int64_t)threads) : 0;
    B.run(jobs, out.nodes, out.depth, threads, defer, defer > 0 ? &sub : nullptr);
    if (!sub.empty()) {
        struct Part { std::vector<rtd::NodeRec32> nodes; int depth = 0; };
        std::vector<Part> parts(sub.size());
        std::atomic<size_t> next(0);
        auto work = [&] {
            for (size_t t; (t = next.fetch_add(1)) < sub.size();) {
                std::vector<Builder::Job> local;
                local.push_back(Builder::Job{-1, 0, sub[t].first, sub[t].count, sub[t].depth});
                B.run(local, parts[t].nodes, parts[t].depth, 1, 0, nullptr);
            }
        };
        std::vector<std::thread> th;
        for (int p = 1; p < threads; p++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (size_t t = 0; t < sub.size(); t++) {
            const int64_t off = (int64_t)out.nodes.size();
            for (rtd::NodeRec32 r : parts[t].nodes) {
                if (r.nL == 0) r.L += (int32_t)off;
                if (r.nR == 0) r.R += (int32_t)off;
                out.nodes.push_back(r);
            }
            rtd::NodeRec32 &pr = out.nodes[(size_t)sub[t].node]; /* a deferred range is never the root: n > defer */
            if (sub[t].side == 0) { pr.L = (int32_t)off; pr.nL = 0; } else { pr.R = (int32_t)off; pr.nR = 0; }
            if (parts[t].depth > out.depth) out.depth = parts[t].depth;
        }
    }
    out.order = idx;
    out.ok = true;
    collapse(out);
    return out;
}

/* Four children per record: a record starts with the two children of a binary node and, while it has room,
 * replaces the inner child with the largest box by that child's own two children. Halves the number of levels,
 * and with it the number of dependent steps of a traversal. */
inline void collapse(Built &b) {
    struct Cand { float box[6]; int32_t ref, count; };
    struct Job { int64_t bin; int64_t parent; int slot; int depth; };
    auto area = [](const float *x) {
        const double dx = (double)x[3] - x[0], dy = (double)x[4] - x[1], dz = (double)x[5] - x[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<Job> jobs;
    jobs.push_back(Job{0, -1, 0, 1});
    for (size_t q = 0; q < jobs.size(); q++) {
        const Job jb = jobs[q];
        Cand c[4];
        int n = 2;
        {
            const rtd::NodeRec32 &r = b.nodes[(size_t)jb.bin];
            for (int a = 0; a < 6; a++) { c[0].box[a] = r.boxL[a]; c[1].box[a] = r.boxR[a]; }
            c[0].ref = r.L; c[0].count = r.nL;
            c[1].ref = r.R; c[1].count = r.nR;
        }
        while (n < 4) {
            int pick = -1;
            double best = -1;
            for (int k = 0; k < n; k++)
                if (c[k].count == 0 && area(c[k].box) > best) { best = area(c[k].box); pick = k; }
            if (pick < 0) break;
            const rtd::NodeRec32 &r = b.nodes[(size_t)c[pick].ref];
            Cand l, rr;
            for (int a = 0; a < 6; a++) { l.box[a] = r.boxL[a]; rr.box[a] = r.boxR[a]; }
            l.ref = r.L; l.count = r.nL;
            rr.ref = r.R; rr.count = r.nR;
            c[pick] = l;
            c[n++] = rr;
        }
        const int64_t id = (int64_t)b.wide.size();
        rtd::NodeRec4 w;
        for (int k = 0; k < 4; k++) {
            if (k < n) {
                for (int a = 0; a < 6; a++) w.box[k][a] = c[k].box[a];
                if (c[k].count > 0 && !(c[k].box[0] <= c[k].box[3])) w.child[k] = RT_CHILD_EMPTY; /* the inverted box of an empty leaf */
                else if (c[k].count > 0) w.child[k] = 0x80000000u | ((uint32_t)(c[k].count - 1) << 28) | (uint32_t)c[k].ref;
                else w.child[k] = 0; /* filled in when the child's record gets its index */
            } else {
                for (int a = 0; a < 3; a++) { w.box[k][a] = HUGE_VALF; w.box[k][3 + a] = -HUGE_VALF; }
                w.child[k] = RT_CHILD_EMPTY;
            }
            w.pad[k] = 0;
        }
        b.wide.push_back(w);
        if (jb.parent >= 0) b.wide[(size_t)jb.parent].child[jb.slot] = (uint32_t)id;
        if (jb.depth > b.wide_depth) b.wide_depth = jb.depth;
        for (int k = 0; k < n; k++)
            if (c[k].count == 0) jobs.push_back(Job{c[k].ref, id, k, jb.depth + 1});
    }
}

} // namespace rtown
