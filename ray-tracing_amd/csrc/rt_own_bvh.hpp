/*
 * rt_own_bvh.hpp - host-side builder of the filter tree.
 *
 * The reference's BVH (raytracer.cpp:244-291) decides which triangles a ray may hit; its quality is poor
 * (one tree per mesh, median cuts, leaves of up to hundreds of triangles). The traversal therefore runs on
 * a second tree built here over ALL triangles of the scene - binned surface-area cuts, leaves of at most
 * two or four triangles (by scene size), triangles copied into this tree's own order - and uses it only to find candidate hits:
 * a ray is kept out of a box just when it certainly misses it. Whether the reference would have reached
 * the winning triangle is then checked on the reference's own leaf box (rt_device.hip.h: rays_end).
 */
#pragma once
#include <math.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "rt_amd.h"
#include "rt_device.hip.h"
#include "rt_threads.hpp"

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
    raw_vector<int32_t> order;         /* position in this tree's triangle array -> position in the caller's array */
    int depth = 0;
    bool ok = false;                   /* false: the scene has a non-finite (or beyond 1e37) coordinate or no triangle; no tree */
};

inline void collapse(Built &b);

/* A primitive's box in single precision, rounded outwards. Conversion and rounding are monotonic, so the union of such
 * boxes equals the outward-rounded union of the double boxes: the records hold the same boxes as if the build had run
 * in double precision, and the build moves half the bytes. */
struct FBox {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) { lo[a] = HUGE_VALF; hi[a] = -HUGE_VALF; }
    }
    void grow(const FBox &b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = b.lo[a] < lo[a] ? b.lo[a] : lo[a];
            hi[a] = b.hi[a] > hi[a] ? b.hi[a] : hi[a];
        }
    }
    double half_area() const {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
    float centre2(int a) const { return lo[a] + hi[a]; } /* twice the centre: only compared and binned */
    void store(float *box) const {
        for (int a = 0; a < 3; a++) { box[a] = lo[a]; box[3 + a] = hi[a]; }
    }
};

static inline FBox box_to_float(const Bounds &b) {
    FBox f;
    for (int a = 0; a < 3; a++) {
        f.lo[a] = nextafterf((float)b.lo[a], -HUGE_VALF);
        f.hi[a] = nextafterf((float)b.hi[a], HUGE_VALF);
    }
    return f;
}

#ifndef RT_OWN_BINS
#define RT_OWN_BINS 16
#endif
static const int kBins = RT_OWN_BINS;
static const int kLeaf = RT_OWN_LEAF;    /* triangles per leaf; at most 8: the leaf count has three bits in a child code */
static const int kSphLeaf = RT_SPH_LEAF; /* spheres per leaf of a sphere tree */
static const int kSahDepth = 40; /* below this depth ranges are halved by count, which bounds the depth */
static const double kCoordMax = 1e37; /* larger coordinates (or non-finite ones) get no tree: twice a centre must fit a float */
#ifndef RT_OWN_SUBTREE
#define RT_OWN_SUBTREE (1 << 16)
#endif
static const int64_t kSubtree = RT_OWN_SUBTREE; /* ranges up to this size are built by one thread, on a copy of their boxes */

/* Builds the binary tree over the primitives idx[0 .. n) of `fb`; `tmp` is scratch of the same length as idx. A leaf of
 * a record names positions pos0 + k of idx. */
struct Builder {
    const FBox *fb;
    int32_t *idx, *tmp;
    int64_t pos0;
    int leaf; /* ranges of at most this many primitives are leaves */

    struct Job {
        int64_t node; /* record that holds this range as child `side`, or -1 for the root range */
        int side;
        int64_t first, count;
        int depth;
    };

    FBox range_bounds(int64_t first, int64_t count) const {
        FBox b;
        b.reset();
        for (int64_t k = first; k < first + count; k++) b.grow(fb[(size_t)idx[(size_t)k]]);
        return b;
    }

    /* Cuts [first, first + count) in two: the binned surface-area cut (kBins bins per axis over the centres' extent),
     * or - all centres equal, or too deep - the median along the longest centre axis. Reorders idx (keeping the order
     * inside each part, so a range's indices stay ascending and its boxes are read front to back), returns the size of
     * the left part and both parts' bounds. threads > 1: the passes over the range are shared out (the top of the tree,
     * where one range is most of the scene). The result does not depend on the thread count. */
    int64_t cut(int64_t first, int64_t count, int depth, int threads, FBox &bl, FBox &br) {
        if (count < kSubtree) threads = 1;
        int64_t nl = count / 2;
        FBox cb; /* of the doubled centres */
        cb.reset();
        {
            std::vector<FBox> part((size_t)threads);
            parallel_parts(count, threads, [&](int p, int64_t b, int64_t e) {
                FBox x;
                x.reset();
                for (int64_t k = first + b; k < first + e; k++) {
                    const FBox &t = fb[(size_t)idx[(size_t)k]];
                    for (int a = 0; a < 3; a++) {
                        const float c = t.centre2(a);
                        x.lo[a] = c < x.lo[a] ? c : x.lo[a];
                        x.hi[a] = c > x.hi[a] ? c : x.hi[a];
                    }
                }
                part[(size_t)p] = x;
            });
            for (int p = 0; p < threads; p++) cb.grow(part[(size_t)p]);
        }
        if (depth <= kSahDepth) {
            struct Bins { FBox bb[3][kBins]; int64_t bc[3][kBins]; };
            std::vector<Bins> part((size_t)threads);
            float scale[3];
            for (int a = 0; a < 3; a++) {
                const float ext = cb.hi[a] - cb.lo[a];
                scale[a] = (ext > 0 && std::isfinite(ext)) ? (float)kBins / ext : 0;
            }
            auto bin_of = [&](const FBox &t, int a) {
                int q = (int)((t.centre2(a) - cb.lo[a]) * scale[a]);
                if (q >= kBins) q = kBins - 1;
                if (q < 0) q = 0;
                return q;
            };
            parallel_parts(count, threads, [&](int p, int64_t b, int64_t e) {
                Bins &B = part[(size_t)p];
                for (int a = 0; a < 3; a++)
                    for (int q = 0; q < kBins; q++) { B.bb[a][q].reset(); B.bc[a][q] = 0; }
                for (int64_t k = first + b; k < first + e; k++) {
                    const FBox &t = fb[(size_t)idx[(size_t)k]];
                    for (int a = 0; a < 3; a++) {
                        if (!(scale[a] > 0)) continue;
                        const int q = bin_of(t, a);
                        B.bb[a][q].grow(t);
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
                FBox acc;
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
                nl = 0;
                for (int q = 0; q <= best_bin; q++) nl += S.bc[best_axis][q];
                /* stable: lefts to tmp[first ..), rights to tmp[first + nl ..), then back */
                std::vector<int64_t> lefts((size_t)threads + 1, 0);
                if (threads > 1) {
                    parallel_parts(count, threads, [&](int p, int64_t b, int64_t e) {
                        int64_t c = 0;
                        for (int64_t k = first + b; k < first + e; k++) c += bin_of(fb[(size_t)idx[(size_t)k]], best_axis) <= best_bin;
                        lefts[(size_t)p + 1] = c;
                    });
                    for (int p = 0; p < threads; p++) lefts[(size_t)p + 1] += lefts[(size_t)p];
                }
                parallel_parts(count, threads, [&](int p, int64_t b, int64_t e) {
                    int32_t *l = tmp + first + lefts[(size_t)p], *r = tmp + first + nl + (b - lefts[(size_t)p]);
                    for (int64_t k = first + b; k < first + e; k++) {
                        const int32_t t = idx[(size_t)k];
                        if (bin_of(fb[(size_t)t], best_axis) <= best_bin) *l++ = t; else *r++ = t;
                    }
                });
                parallel_parts(count, threads, [&](int, int64_t b, int64_t e) {
                    std::copy(tmp + first + b, tmp + first + e, idx + first + b);
                });
                bl.reset();
                br.reset();
                for (int q = 0; q < kBins; q++) (q <= best_bin ? bl : br).grow(S.bb[best_axis][q]); /* the parts' bounds are the unions of their bins */
                return nl;
            }
        }
        /* all centres coincide, or too deep: halve the range along its longest centre axis */
        nl = count / 2;
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
        std::nth_element(idx + first, idx + first + nl, idx + first + count, [&](int32_t x, int32_t y) {
            const float cx = fb[(size_t)x].centre2(axis), cy = fb[(size_t)y].centre2(axis);
            return cx < cy || (cx == cy && x < y);
        });
        bl = range_bounds(first, nl);
        br = range_bounds(first + nl, count - nl);
        return nl;
    }

    /* Breadth-first build of the ranges in `jobs` into `nodes` (a child's index is larger than its parent's). Ranges of at
     * most `defer` primitives are not cut here but handed back in `deferred` (0: none are). */
    void run(std::vector<Job> &jobs, std::vector<rtd::NodeRec32> &nodes, int &max_depth, int threads, int64_t defer, std::vector<Job> *deferred) {
        for (size_t q = 0; q < jobs.size(); q++) {
            const Job jb = jobs[q];
            if (deferred && jb.count <= defer) { deferred->push_back(jb); continue; }
            FBox bl, br;
            const int64_t nl = cut(jb.first, jb.count, jb.depth, threads, bl, br), nr = jb.count - nl;
            const int64_t id = (int64_t)nodes.size();
            rtd::NodeRec32 r;
            bl.store(r.boxL);
            br.store(r.boxR);
            r.L = (int32_t)(pos0 + jb.first); r.nL = (int32_t)nl;
            r.R = (int32_t)(pos0 + jb.first + nl); r.nR = (int32_t)nr;
            nodes.push_back(r);
            if (jb.node >= 0) {
                rtd::NodeRec32 &pr = nodes[(size_t)jb.node];
                if (jb.side == 0) { pr.L = (int32_t)id; pr.nL = 0; } else { pr.R = (int32_t)id; pr.nR = 0; }
            }
            if (jb.depth > max_depth) max_depth = jb.depth;
            if (nl > leaf) jobs.push_back(Job{id, 0, jb.first, nl, jb.depth + 1});
            if (nr > leaf) jobs.push_back(Job{id, 1, jb.first + nl, nr, jb.depth + 1});
        }
    }
};

inline Built build_boxes(const raw_vector<FBox> &fb, int leaf);

/* the tree over the triangles tris[keep[0 .. n)] (keep == nullptr: tris[0 .. n)); `order` counts within keep */
inline Built build(const rt_tri *tris, const int32_t *keep, int64_t n, int leaf = kLeaf) {
    if (n <= 0) return Built();
    const int threads = host_threads();
    PoolScope pool(n >= kSubtree ? threads : 1);
    raw_vector<FBox> fb((size_t)n);
    std::vector<char> bad((size_t)threads, 0);
    parallel_parts(n, n >= kSubtree ? threads : 1, [&](int p, int64_t b0, int64_t e0) {
        for (int64_t k = b0; k < e0; k++) {
            const rt_tri &t = tris[keep ? keep[k] : k];
            Bounds b;
            b.reset();
            for (int a = 0; a < 3; a++) {
                const double v[3] = {t.A[a], t.A[a] + t.e1[a], t.A[a] + t.e2[a]};
                for (int q = 0; q < 3; q++) {
                    if (!(fabs(v[q]) < kCoordMax)) bad[(size_t)p] = 1;
                    if (v[q] < b.lo[a]) b.lo[a] = v[q];
                    if (v[q] > b.hi[a]) b.hi[a] = v[q];
                }
            }
            fb[(size_t)k] = box_to_float(b);
        }
    });
    for (char x : bad)
        if (x) return Built();
    return build_boxes(fb, leaf);
}

/* the tree over any primitives, given their (finite) boxes: the spheres of a long sphere list */
inline Built build_boxes(const std::vector<Bounds> &tb) {
    raw_vector<FBox> fb(tb.size());
    for (size_t k = 0; k < tb.size(); k++) {
        for (int a = 0; a < 3; a++)
            if (!(fabs(tb[k].lo[a]) < kCoordMax) || !(fabs(tb[k].hi[a]) < kCoordMax)) return Built();
        fb[k] = box_to_float(tb[k]);
    }
    return build_boxes(fb, kSphLeaf);
}

inline Built build_boxes(const raw_vector<FBox> &fb, int leaf) {
    Built out;
    const int64_t n = (int64_t)fb.size();
    if (n <= 0) return out;
    const int threads = host_threads();
    PoolScope pool(n >= kSubtree ? threads : 1);
    raw_vector<int32_t> idx((size_t)n), tmp((size_t)n);
    parallel_parts(n, n >= kSubtree ? threads : 1, [&](int, int64_t b0, int64_t e0) {
        for (int64_t k = b0; k < e0; k++) idx[(size_t)k] = (int32_t)k;
    });
    Builder B{fb.data(), idx.data(), tmp.data(), 0, leaf};
    if (n <= leaf) { /* a tree needs one record: both children are leaves (the right one may be empty) */
        rtd::NodeRec32 r;
        const int64_t nl = (n + 1) / 2;
        B.range_bounds(0, nl).store(r.boxL);
        r.L = 0; r.nL = (int32_t)nl;
        if (n - nl > 0) { B.range_bounds(nl, n - nl).store(r.boxR); r.R = (int32_t)nl; r.nR = (int32_t)(n - nl); }
        else { for (int a = 0; a < 3; a++) { r.boxR[a] = HUGE_VALF; r.boxR[3 + a] = -HUGE_VALF; } r.R = 0; r.nR = 1; } /* inverted box: certainly missed */
        out.nodes.push_back(r);
        out.order = idx;
        out.depth = 1;
        out.ok = true;
        collapse(out);
        return out;
    }
    /* The top of the tree with all threads inside each cut; ranges of at most kSubtree primitives are whole subtrees
     * that the threads then build one at a time (largest first), each on a copy of its range's boxes small enough to
     * stay in the core's cache, into arrays of their own. These are appended in the order of the ranges, so the tree is
     * the same whatever the thread count and timing. */
    std::vector<Builder::Job> jobs, sub;
    jobs.push_back(Builder::Job{-1, 0, 0, n, 1});
    const int64_t defer = n > kSubtree ? kSubtree : 0;
    stage_tick("  tree build: boxes, indices");
    B.run(jobs, out.nodes, out.depth, threads, defer, defer > 0 ? &sub : nullptr);
    stage_tick("  tree build: top");
    if (!sub.empty()) {
        struct Part { std::vector<rtd::NodeRec32> nodes; int depth = 0; };
        std::vector<Part> parts(sub.size());
        std::vector<size_t> by_size(sub.size());
        for (size_t t = 0; t < sub.size(); t++) by_size[t] = t;
        std::sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return sub[a].count > sub[b].count || (sub[a].count == sub[b].count && a < b); });
        parallel_tasks(sub.size(), threads, [&](size_t q) {
            const size_t t = by_size[q];
            const int64_t first = sub[t].first, count = sub[t].count;
            raw_vector<FBox> lb((size_t)count);
            raw_vector<int32_t> li((size_t)count), lt((size_t)count);
            for (int64_t k = 0; k < count; k++) { lb[(size_t)k] = fb[(size_t)idx[(size_t)(first + k)]]; li[(size_t)k] = (int32_t)k; }
            Builder L{lb.data(), li.data(), lt.data(), first, leaf};
            std::vector<Builder::Job> local;
            local.push_back(Builder::Job{-1, 0, 0, count, sub[t].depth});
            parts[t].nodes.reserve((size_t)count / 2);
            L.run(local, parts[t].nodes, parts[t].depth, 1, 0, nullptr);
            for (int64_t k = 0; k < count; k++) lt[(size_t)k] = idx[(size_t)(first + li[(size_t)k])];
            std::copy(lt.begin(), lt.end(), idx.begin() + first);
        });
        stage_tick("  tree build: subtrees");
        std::vector<int64_t> off(sub.size() + 1);
        off[0] = (int64_t)out.nodes.size();
        for (size_t t = 0; t < sub.size(); t++) off[t + 1] = off[t] + (int64_t)parts[t].nodes.size();
        out.nodes.resize((size_t)off[sub.size()]);
        for (size_t t = 0; t < sub.size(); t++) {
            rtd::NodeRec32 &pr = out.nodes[(size_t)sub[t].node]; /* a deferred range is never the root: n > defer */
            if (sub[t].side == 0) { pr.L = (int32_t)off[t]; pr.nL = 0; } else { pr.R = (int32_t)off[t]; pr.nR = 0; }
            if (parts[t].depth > out.depth) out.depth = parts[t].depth;
        }
        parallel_tasks(sub.size(), threads, [&](size_t t) {
            rtd::NodeRec32 *dst = out.nodes.data() + off[t];
            for (rtd::NodeRec32 r : parts[t].nodes) {
                if (r.nL == 0) r.L += (int32_t)off[t];
                if (r.nR == 0) r.R += (int32_t)off[t];
                *dst++ = r;
            }
        });
    }
    out.order.swap(idx);
    out.ok = true;
    stage_tick("  tree build: append");
    collapse(out);
    stage_tick("  tree build: collapse");
    return out;
}

/* Four children per record: a record starts with the two children of a binary node and, while it has room,
 * replaces the inner child with the largest box by that child's own two children. Halves the number of levels,
 * and with it the number of dependent steps of a traversal. */
struct Collapser {
    const std::vector<rtd::NodeRec32> &nodes;
    struct Job { int64_t bin; int64_t parent; int slot; int depth; };

    /* Breadth-first from `jobs` into `wide`. With `frontier`: stops as soon as kFrontier jobs are pending and hands
     * them back (the same place whatever the thread count). */
    void run(std::vector<Job> &jobs, std::vector<rtd::NodeRec4> &wide, int &max_depth, std::vector<Job> *frontier) const {
        struct Cand { float box[6]; int32_t ref, count; };
        auto area = [](const float *x) {
            const double dx = (double)x[3] - x[0], dy = (double)x[4] - x[1], dz = (double)x[5] - x[2];
            return dx * dy + dy * dz + dz * dx;
        };
        for (size_t q = 0; q < jobs.size(); q++) {
            if (frontier && jobs.size() - q >= kFrontier) {
                frontier->assign(jobs.begin() + (ptrdiff_t)q, jobs.end());
                return;
            }
            const Job jb = jobs[q];
            Cand c[4];
            int n = 2;
            {
                const rtd::NodeRec32 &r = nodes[(size_t)jb.bin];
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
                const rtd::NodeRec32 &r = nodes[(size_t)c[pick].ref];
                Cand l, rr;
                for (int a = 0; a < 6; a++) { l.box[a] = r.boxL[a]; rr.box[a] = r.boxR[a]; }
                l.ref = r.L; l.count = r.nL;
                rr.ref = r.R; rr.count = r.nR;
                c[pick] = l;
                c[n++] = rr;
            }
            const int64_t id = (int64_t)wide.size();
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
            wide.push_back(w);
            if (jb.parent >= 0) wide[(size_t)jb.parent].child[jb.slot] = (uint32_t)id;
            if (jb.depth > max_depth) max_depth = jb.depth;
            for (int k = 0; k < n; k++)
                if (c[k].count == 0) jobs.push_back(Job{c[k].ref, id, k, jb.depth + 1});
        }
    }
    static const size_t kFrontier = 512;
};

/* The records are laid out breadth-first down to the level that first holds 512 of them; the subtree under each of those
 * follows as one block, breadth-first inside (built by the threads, one subtree at a time). */
inline void collapse(Built &b) {
    const int threads = b.nodes.size() >= (size_t)kSubtree ? host_threads() : 1;
    Collapser C{b.nodes};
    std::vector<Collapser::Job> jobs, frontier;
    jobs.push_back(Collapser::Job{0, -1, 0, 1});
    b.wide.reserve(b.nodes.size() / 2 + 16);
    C.run(jobs, b.wide, b.wide_depth, &frontier);
    if (frontier.empty()) return;
    struct Part { std::vector<rtd::NodeRec4> wide; int depth = 0; };
    std::vector<Part> parts(frontier.size());
    parallel_tasks(frontier.size(), threads, [&](size_t t) {
        std::vector<Collapser::Job> local;
        local.push_back(Collapser::Job{frontier[t].bin, -1, 0, frontier[t].depth});
        C.run(local, parts[t].wide, parts[t].depth, nullptr);
    });
    std::vector<int64_t> off(frontier.size() + 1);
    off[0] = (int64_t)b.wide.size();
    for (size_t t = 0; t < frontier.size(); t++) off[t + 1] = off[t] + (int64_t)parts[t].wide.size();
    b.wide.resize((size_t)off[frontier.size()]);
    for (size_t t = 0; t < frontier.size(); t++) {
        b.wide[(size_t)frontier[t].parent].child[frontier[t].slot] = (uint32_t)off[t]; /* the root is never handed back: it is processed first */
        if (parts[t].depth > b.wide_depth) b.wide_depth = parts[t].depth;
    }
    parallel_tasks(frontier.size(), threads, [&](size_t t) {
        rtd::NodeRec4 *dst = b.wide.data() + off[t];
        for (rtd::NodeRec4 w : parts[t].wide) {
            for (int k = 0; k < 4; k++)
                if (!(w.child[k] & 0x80000000u)) w.child[k] += (uint32_t)off[t];
            *dst++ = w;
        }
    });
}

} // namespace rtown
