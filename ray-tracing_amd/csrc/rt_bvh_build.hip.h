/*
 * rt_bvh_build.hip.h - TriangleMesh::buildBVH (raytracer.cpp:220-291) on the GPU, with the reference's result:
 * the same boxes, the same tree and the same order of the triangles (which the traversal's tie rule depends on).
 * Included by rt_capi.hip (it uses the context and the error helpers defined there).
 *
 * The reference builds depth first; here all nodes of one depth are processed together ("level"), which gives the
 * same tree because a node's work depends only on its own range [debut, fin). Per level, for every open node:
 *   1. box of the range (:220-242): minima / maxima are order-independent except for the sign of a zero: std::min and
 *      std::max keep the first of two equal values, so a zero minimum or maximum has the sign of the first zero the
 *      loop meets along that axis, which a second pass finds (smallest position, then vertex slot);
 *   2. the axis and the middle of the box (:250-267);
 *   3. the partition loop (:268-278) "for i: if centre(i) < middle: swap(a[i], a[pivot]); pivot++". Its result is
 *      not a stable partition: with S = "centre < middle" and U = the rest,
 *        - the S elements end up in front in their original order;
 *        - the U elements form a queue between pivot and i: a U element is appended at its back, and every S element
 *          met while the queue is not empty moves the queue's front element to its back (that is what the swap does).
 *      Written as one append-only sequence V - a U element appends itself, a rotating S element appends a copy of
 *      V[number of rotations so far] - the queue at the end is the last |U| entries of V. Which slot of V a position
 *      writes, and which earlier slot a copy refers to, follow from prefix sums of S over the positions; the copies
 *      are resolved by pointer jumping (log2 of the largest range rounds).
 *   4. the split rule (:283): the nodes that go on get their two children appended to the node array (places from a
 *      prefix sum over the level), and these form the next level.
 * Everything of a level stays on the device; the host reads three numbers per level (nodes of the next level, the
 * largest of their ranges, an error flag) and the finished tree at the end, which it renumbers into the order in which
 * the reference's recursion creates the nodes.
 */
#pragma once
#include <rocprim/rocprim.hpp>

#include <memory>

namespace rtbuild {

/* order-preserving map double -> uint64 (no NaNs here: the input is validated); -0.0 is taken as +0.0 */
__device__ __forceinline__ unsigned long long key_of(double d) {
    if (d == 0.0) d = 0.0;
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
struct LevelNode { /* one open node of the current level, as the kernels see it */
    int32_t debut, fin;
    int32_t dim;      /* axis of the cut */
    int32_t lead;     /* position of the first U element, or -1 */
    int32_t base_s;   /* prefix sum of S at debut */
    int32_t n_s;      /* S elements in the range */
    double milieu;
};

/* node of a position: the last open node whose debut is <= pos, if pos is inside it; else -1 (a finished leaf) */
__device__ __forceinline__ int node_at(const LevelNode *nd, int n_nodes, int pos) {
    int lo = 0, hi = n_nodes - 1, at = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (nd[mid].debut <= pos) { at = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return (at >= 0 && pos < nd[at].fin) ? at : -1;
}

__global__ void k_node_of(const LevelNode *nd, int n_nodes, int n, int32_t *node_of) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < n) node_of[pos] = node_at(nd, n_nodes, pos);
}

/* minima and maxima of the vertices of the triangles of every open node */
__global__ void k_box(const double *verts, const int32_t *tri_vtx, const int32_t *perm, const int32_t *node_of, int n, unsigned long long *keys) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const int node = pos < n ? node_of[pos] : -1;
    unsigned long long lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = ~0ull; hi[a] = 0ull; }
    if (node >= 0) {
        const int t = perm[pos];
        for (int q = 0; q < 3; q++) {
            const double *v = verts + (size_t)tri_vtx[3 * (size_t)t + q] * 3;
            for (int a = 0; a < 3; a++) {
                const unsigned long long k = key_of(v[a]);
                if (k < lo[a]) lo[a] = k;
                if (k > hi[a]) hi[a] = k;
            }
        }
    }
    /* The positions of a node are contiguous, so the lanes of a wave fall into runs of equal node: a segmented
     * reduction over the wave (combine with lane + 1, 2, 4 ... while it is in the same node) leaves every run's result
     * in its first lane, and only those lanes touch memory. */
    const int lane = (int)(threadIdx.x & 63);
    for (int off = 1; off < 64; off <<= 1) {
        const int other = __shfl_down(node, off, 64);
        const bool same = (lane + off < 64) && other == node;
        for (int a = 0; a < 3; a++) {
            const unsigned long long l2 = __shfl_down(lo[a], off, 64), h2 = __shfl_down(hi[a], off, 64);
            if (same && l2 < lo[a]) lo[a] = l2;
            if (same && h2 > hi[a]) hi[a] = h2;
        }
    }
    const int before = __shfl_up(node, 1, 64);
    if (node >= 0 && (lane == 0 || before != node))
        for (int a = 0; a < 3; a++) {
            atomicMin(keys + (size_t)node * 6 + a, lo[a]);
            atomicMax(keys + (size_t)node * 6 + 3 + a, hi[a]);
        }
}

/* S = "the triangle's centre along the node's axis lies below the middle" (:271-272); 0 outside the open nodes */
__global__ void k_flags(const double *verts, const int32_t *tri_vtx, const int32_t *perm, const int32_t *node_of, const LevelNode *nd, int n, int32_t *flag_s) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int node = node_of[pos];
    int s = 0;
    if (node >= 0) {
        const int t = perm[pos], dim = nd[node].dim;
        const double a = verts[(size_t)tri_vtx[3 * (size_t)t] * 3 + dim], b = verts[(size_t)tri_vtx[3 * (size_t)t + 1] * 3 + dim],
                     c = verts[(size_t)tri_vtx[3 * (size_t)t + 2] * 3 + dim];
        const double centre = (a + b + c) / 3;
        s = centre < nd[node].milieu ? 1 : 0;
    }
    flag_s[pos] = s;
}

/* per node: the prefix sum at debut, the number of S elements, and the first U position (the one U position before
 * which everything in the range is S) */
__global__ void k_node_counts(const int32_t *flag_s, const int32_t *scan_s, const int32_t *node_of, LevelNode *nd, int n) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int node = node_of[pos];
    if (node < 0) return;
    const int debut = nd[node].debut, fin = nd[node].fin;
    if (pos == debut) nd[node].base_s = scan_s[debut];
    if (pos == fin - 1) nd[node].n_s = scan_s[pos] + flag_s[pos] - scan_s[debut];
    if (!flag_s[pos] && scan_s[pos] - scan_s[debut] == pos - debut) nd[node].lead = pos;
}

/* the front part of the new order, and the append-only sequence V of the queue (stored from the node's debut on):
 * a value >= 0 refers to another slot of V, a value < 0 is the triangle -1 - value */
__global__ void k_build_v(const int32_t *perm, const int32_t *flag_s, const int32_t *scan_s, const int32_t *node_of, const LevelNode *nd, int n,
                          int32_t *perm2, int32_t *V) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int node = node_of[pos];
    if (node < 0) return;
    const LevelNode d = nd[node];
    const int t = perm[pos];
    const int ns = scan_s[pos] - d.base_s, nu = (pos - d.debut) - ns;
    if (flag_s[pos]) {
        perm2[d.debut + ns] = t;
        if (d.lead >= 0 && pos > d.lead) { /* the queue is not empty: its front goes to its back */
            const int r = ns - (d.lead - d.debut);
            V[d.debut + nu + r] = d.debut + r;
        }
    } else {
        const int r = ns - (d.lead - d.debut); /* pos >= lead here, so r >= 0 */
        V[d.debut + nu + r] = -1 - t;
    }
}

__global__ void k_jump(const int32_t *node_of, const LevelNode *nd, int n, int32_t *V) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int node = node_of[pos];
    if (node < 0) return;
    const LevelNode d = nd[node];
    if (d.lead < 0) return;
    const int used = ((d.fin - d.debut) - d.n_s) + (d.n_s - (d.lead - d.debut)); /* U elements + rotations */
    if (pos - d.debut >= used) return;
    const int v = V[pos];
    if (v >= 0) V[pos] = V[v]; /* a race with the owner of slot v only yields an older or a newer link of the same chain */
}

/* the back part of the new order: the queue as it stands at the end of the loop */
__global__ void k_back(const int32_t *node_of, const LevelNode *nd, int n, const int32_t *V, int32_t *perm2, int32_t *unresolved) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int node = node_of[pos];
    if (node < 0) return;
    const LevelNode d = nd[node];
    const int pivot = d.debut + d.n_s;
    if (pos < pivot) return;
    const int rot = d.n_s - (d.lead - d.debut);
    const int v = V[d.debut + rot + (pos - pivot)];
    if (v >= 0) { atomicOr(unresolved, 1); return; }
    perm2[pos] = -1 - v;
}


__device__ __forceinline__ double dvalue_of(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

struct LevelInfo { /* what the host reads after a level */
    int32_t m_next, largest_next, unresolved, pad;
};

/* the open nodes of this level as the kernels want them, their boxes at the start values of :223-224 */
__global__ void k_level_init(const int32_t *open, int m, const rt_build_node *T, LevelNode *nd, unsigned long long *keys, unsigned long long *zfirst) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const rt_build_node &t = T[open[q]];
    LevelNode d;
    d.debut = t.debut; d.fin = t.fin; d.dim = 0; d.lead = -1; d.base_s = 0; d.n_s = 0; d.milieu = 0;
    nd[q] = d;
    for (int a = 0; a < 3; a++) {
        keys[(size_t)q * 6 + a] = key_of(1E9);
        keys[(size_t)q * 6 + 3 + a] = key_of(-1E9);
        zfirst[(size_t)q * 3 + a] = ~0ull;
    }
}

/* for the axes along which a node's minimum or maximum is zero: the first zero coordinate in the loop's order
 * (position, then vertex slot) and its sign bit */
__global__ void k_zero_first(const double *verts, const int32_t *tri_vtx, const int32_t *perm, const int32_t *node_of, int n, const unsigned long long *keys,
                             unsigned long long *zfirst) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int node = node_of[pos];
    if (node < 0) return;
    const unsigned long long kz = key_of(0.0);
    for (int a = 0; a < 3; a++) {
        if (keys[(size_t)node * 6 + a] != kz && keys[(size_t)node * 6 + 3 + a] != kz) continue;
        const int t = perm[pos];
        for (int slot = 0; slot < 3; slot++) {
            const double x = verts[(size_t)tri_vtx[3 * (size_t)t + slot] * 3 + a];
            if (x == 0.0) {
                const unsigned long long sign = (unsigned long long)__double_as_longlong(x) >> 63;
                atomicMin(zfirst + (size_t)node * 3 + a, ((((unsigned long long)pos << 2) | (unsigned long long)slot) << 1) | sign);
                break; /* later slots of this triangle come later in the loop */
            }
        }
    }
}

/* box, axis and middle of every open node (:246-267) */
__global__ void k_axis(const int32_t *open, int m, const unsigned long long *keys, const unsigned long long *zfirst, rt_build_node *T, LevelNode *nd) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    double box[6];
    for (int a = 0; a < 6; a++) {
        box[a] = dvalue_of(keys[(size_t)q * 6 + a]);
        if (box[a] == 0.0) {
            const unsigned long long z = zfirst[(size_t)q * 3 + (a % 3)];
            box[a] = (z != ~0ull && (z & 1ull)) ? -0.0 : 0.0;
        }
    }
    rt_build_node &t = T[open[q]];
    for (int a = 0; a < 6; a++) t.box[a] = box[a];
    const double dx = box[3] - box[0], dy = box[4] - box[1], dz = box[5] - box[2];
    int dim;
    if (dx >= dy && dx >= dz) dim = 0;
    else if (dy >= dx && dy >= dz) dim = 1;
    else dim = 2;
    nd[q].dim = dim;
    nd[q].milieu = (box[dim] + box[3 + dim]) * 0.5;
}

/* :283 - which open nodes are split */
__global__ void k_split_flags(const LevelNode *nd, int m, int32_t *flag) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const LevelNode d = nd[q];
    const int pivot = d.debut + d.n_s;
    flag[q] = (pivot == d.debut || pivot == d.fin || d.fin - d.debut < 5) ? 0 : 1;
}

/* the children of the nodes that are split: appended to the node array in the level's order, two per node */
__global__ void k_split_apply(const LevelNode *nd, int m, const int32_t *flag, const int32_t *rank, const int32_t *open, int tree_count, rt_build_node *T,
                              int32_t *open_next, LevelInfo *info) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    if (q == m - 1) info->m_next = 2 * (rank[q] + flag[q]);
    if (!flag[q]) return;
    const LevelNode d = nd[q];
    const int pivot = d.debut + d.n_s, id = open[q], l = tree_count + 2 * rank[q];
    rt_build_node a, b;
    for (int k = 0; k < 6; k++) a.box[k] = b.box[k] = 0;
    a.debut = d.debut; a.fin = pivot; b.debut = pivot; b.fin = d.fin;
    a.fg = a.fd = b.fg = b.fd = -1;
    a.depth = b.depth = T[id].depth + 1;
    a.pad = b.pad = 0;
    T[l] = a;
    T[l + 1] = b;
    T[id].fg = l;
    T[id].fd = l + 1;
    open_next[2 * rank[q]] = l;
    open_next[2 * rank[q] + 1] = l + 1;
    const int big = (pivot - d.debut) > (d.fin - pivot) ? (pivot - d.debut) : (d.fin - pivot);
    atomicMax(&info->largest_next, big);
}

__global__ void k_iota(int32_t *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

struct DeviceArrays { /* device allocations of one build, released when it ends */
    std::vector<void *> p;
    ~DeviceArrays() { for (void *q : p) (void)hipFree(q); }
    template <typename T> T *get(size_t count) {
        void *q = nullptr;
        if (hipMalloc(&q, count * sizeof(T) + 16) != hipSuccess) return nullptr;
        p.push_back(q);
        return (T *)q;
    }
};

} // namespace rtbuild

extern "C" int rt_build_bvh(rt_ctx *c, const double *verts, int64_t nv, const int32_t *tri_vtx, int64_t nt, int32_t *order, rt_build_node *nodes,
                            int64_t node_cap, int64_t *n_nodes_out) {
    using namespace rtbuild;
    if (!c) return RT_E_INVALID;
    if (!verts || !tri_vtx || !order || !nodes || !n_nodes_out || nv <= 0 || nt <= 0 || nt >= (1LL << 30)) return fail(c, RT_E_INVALID, "rt_build_bvh: bad arguments");
    for (int64_t i = 0; i < nv * 3; i++)
        if (!std::isfinite(verts[i])) return fail(c, RT_E_INVALID, "rt_build_bvh: vertex %lld is not finite", (long long)(i / 3));
    for (int64_t i = 0; i < nt * 3; i++)
        if (tri_vtx[i] < 0 || tri_vtx[i] >= nv) return fail(c, RT_E_INVALID, "rt_build_bvh: triangle %lld refers to a vertex that does not exist", (long long)(i / 3));
    HIPCHK(c, hipSetDevice(c->device));
    const int n = (int)nt;
    DeviceArrays dev;
    const int64_t cap_dev = 2 * nt + 1;
    double *d_verts = dev.get<double>((size_t)nv * 3);
    int32_t *d_tri = dev.get<int32_t>((size_t)nt * 3), *d_perm = dev.get<int32_t>(n), *d_perm2 = dev.get<int32_t>(n), *d_node_of = dev.get<int32_t>(n),
            *d_flag = dev.get<int32_t>(n), *d_scan = dev.get<int32_t>(n), *d_V = dev.get<int32_t>(n), *d_open = dev.get<int32_t>(n), *d_open2 = dev.get<int32_t>(n),
            *d_sflag = dev.get<int32_t>(n), *d_srank = dev.get<int32_t>(n);
    LevelNode *d_nd = dev.get<LevelNode>(n);
    LevelInfo *d_info = dev.get<LevelInfo>(1);
    unsigned long long *d_keys = dev.get<unsigned long long>((size_t)n * 6), *d_zfirst = dev.get<unsigned long long>((size_t)n * 3);
    rt_build_node *d_T = dev.get<rt_build_node>((size_t)cap_dev);
    size_t tmp_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, tmp_bytes, d_flag, d_scan, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream);
    void *d_tmp = dev.get<char>(tmp_bytes);
    if (!d_verts || !d_tri || !d_perm || !d_perm2 || !d_node_of || !d_flag || !d_scan || !d_V || !d_open || !d_open2 || !d_sflag || !d_srank || !d_nd || !d_info ||
        !d_keys || !d_zfirst || !d_T || !d_tmp)
        return fail(c, RT_E_NOMEM, "rt_build_bvh: device allocation failed");
    const int B = 256, G = (n + B - 1) / B;
    HIPCHK(c, hipMemcpyAsync(d_verts, verts, (size_t)nv * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_tri, tri_vtx, (size_t)nt * 12, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_iota, dim3(G), dim3(B), 0, c->stream, d_perm, n);
    {
        rt_build_node root;
        memset(&root, 0, sizeof root);
        root.debut = 0; root.fin = n; root.fg = root.fd = -1; root.depth = 0;
        const int32_t zero = 0;
        HIPCHK(c, hipMemcpyAsync(d_T, &root, sizeof root, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_open, &zero, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream)); /* root and zero are locals */
    }
    int m = 1, largest = n, tree_count = 1;
    while (m > 0) {
        const int Gm = (m + B - 1) / B;
        HIPCHK(c, hipMemsetAsync(d_info, 0, sizeof(LevelInfo), c->stream));
        hipLaunchKernelGGL(k_level_init, dim3(Gm), dim3(B), 0, c->stream, d_open, m, d_T, d_nd, d_keys, d_zfirst);
        hipLaunchKernelGGL(k_node_of, dim3(G), dim3(B), 0, c->stream, d_nd, m, n, d_node_of);
        hipLaunchKernelGGL(k_box, dim3(G), dim3(B), 0, c->stream, d_verts, d_tri, d_perm, d_node_of, n, d_keys);
        hipLaunchKernelGGL(k_zero_first, dim3(G), dim3(B), 0, c->stream, d_verts, d_tri, d_perm, d_node_of, n, d_keys, d_zfirst);
        hipLaunchKernelGGL(k_axis, dim3(Gm), dim3(B), 0, c->stream, d_open, m, d_keys, d_zfirst, d_T, d_nd);
        hipLaunchKernelGGL(k_flags, dim3(G), dim3(B), 0, c->stream, d_verts, d_tri, d_perm, d_node_of, d_nd, n, d_flag);
        if (rocprim::exclusive_scan(d_tmp, tmp_bytes, d_flag, d_scan, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream) != hipSuccess)
            return fail(c, RT_E_DEVICE, "rt_build_bvh: prefix sum failed");
        hipLaunchKernelGGL(k_node_counts, dim3(G), dim3(B), 0, c->stream, d_flag, d_scan, d_node_of, d_nd, n);
        HIPCHK(c, hipMemcpyAsync(d_perm2, d_perm, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_build_v, dim3(G), dim3(B), 0, c->stream, d_perm, d_flag, d_scan, d_node_of, d_nd, n, d_perm2, d_V);
        int rounds = 1;
        while ((1 << rounds) < largest + 1) rounds++;
        for (int r = 0; r < rounds + 1; r++) hipLaunchKernelGGL(k_jump, dim3(G), dim3(B), 0, c->stream, d_node_of, d_nd, n, d_V);
        hipLaunchKernelGGL(k_back, dim3(G), dim3(B), 0, c->stream, d_node_of, d_nd, n, d_V, d_perm2, &d_info->unresolved);
        hipLaunchKernelGGL(k_split_flags, dim3(Gm), dim3(B), 0, c->stream, d_nd, m, d_sflag);
        if (rocprim::exclusive_scan(d_tmp, tmp_bytes, d_sflag, d_srank, 0, (size_t)m, rocprim::plus<int32_t>(), c->stream) != hipSuccess)
            return fail(c, RT_E_DEVICE, "rt_build_bvh: prefix sum failed");
        hipLaunchKernelGGL(k_split_apply, dim3(Gm), dim3(B), 0, c->stream, d_nd, m, d_sflag, d_srank, d_open, tree_count, d_T, d_open2, d_info);
        HIPCHK(c, hipGetLastError());
        LevelInfo info;
        HIPCHK(c, hipMemcpyAsync(&info, d_info, sizeof info, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (info.unresolved) return fail(c, RT_E_DEVICE, "rt_build_bvh: internal error (unresolved queue entry)");
        std::swap(d_perm, d_perm2);
        std::swap(d_open, d_open2);
        tree_count += info.m_next;
        m = info.m_next;
        largest = info.largest_next;
    }
    if ((int64_t)tree_count > node_cap) return fail(c, RT_E_INVALID, "rt_build_bvh: %d nodes do not fit the caller's %lld", tree_count, (long long)node_cap);
    HIPCHK(c, hipMemcpy(order, d_perm, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::unique_ptr<rt_build_node[]> tree(new rt_build_node[(size_t)tree_count]); /* level order; renumbered below */
    HIPCHK(c, hipMemcpy(tree.get(), d_T, (size_t)tree_count * sizeof(rt_build_node), hipMemcpyDeviceToHost));
    /* the numbering of the reference's recursion: a node's two children get the next two numbers when it is split, and the
     * left subtree is completed before the right one */
    {
        std::vector<int> newid((size_t)tree_count, -1), stack;
        int count = 1;
        newid[0] = 0;
        stack.push_back(0);
        while (!stack.empty()) {
            const int id = stack.back();
            stack.pop_back();
            if (tree[(size_t)id].fg < 0) continue;
            newid[(size_t)tree[(size_t)id].fg] = count++;
            newid[(size_t)tree[(size_t)id].fd] = count++;
            stack.push_back(tree[(size_t)id].fd);
            stack.push_back(tree[(size_t)id].fg);
        }
        for (int id = 0; id < tree_count; id++) {
            const rt_build_node &t = tree[(size_t)id];
            rt_build_node &o = nodes[(size_t)newid[(size_t)id]];
            o = t;
            o.fg = t.fg >= 0 ? newid[(size_t)t.fg] : -1;
            o.fd = t.fd >= 0 ? newid[(size_t)t.fd] : -1;
        }
    }
    *n_nodes_out = (int64_t)tree_count;
    return RT_OK;
}
