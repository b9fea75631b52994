/*
 * rt_bvh_build.hip.h - TriangleMesh::buildBVH (raytracer.cpp:220-291) on the GPU, with the reference's result:
 * the same boxes, the same tree and the same order of the triangles (which the traversal's tie rule depends on).
 * Included by rt_capi.hip (it uses the context and the error helpers defined there).
 *
 * The reference builds depth first; here all nodes of one depth are processed together ("level"), which gives the
 * same tree because a node's work depends only on its own range [debut, fin). Per level, for every open node:
 *   1. box of the range (:220-242): minima / maxima are order-independent except for the sign of a zero, so a node
 *      with a zero box coordinate gets its box recomputed on the host in the reference's order;
 *   2. the axis and the middle of the box (:250-267), on the host (a few numbers per node);
 *   3. the partition loop (:268-278) "for i: if centre(i) < middle: swap(a[i], a[pivot]); pivot++". Its result is
 *      not a stable partition: with S = "centre < middle" and U = the rest,
 *        - the S elements end up in front in their original order;
 *        - the U elements form a queue between pivot and i: a U element is appended at its back, and every S element
 *          met while the queue is not empty moves the queue's front element to its back (that is what the swap does).
 *      Written as one append-only sequence V - a U element appends itself, a rotating S element appends a copy of
 *      V[number of rotations so far] - the queue at the end is the last |U| entries of V. Which slot of V a position
 *      writes, and which earlier slot a copy refers to, follow from prefix sums of S over the positions; the copies
 *      are resolved by pointer jumping (log2 of the largest range rounds).
 *   4. the split rule (:283), on the host: the level's nodes that go on form the next level.
 */
#pragma once
#include <rocprim/rocprim.hpp>

namespace rtbuild {

/* order-preserving map double -> uint64 (no NaNs here: the input is validated); -0.0 is taken as +0.0 */
__device__ __forceinline__ unsigned long long key_of(double d) {
    if (d == 0.0) d = 0.0;
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static inline double value_of(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double d;
    memcpy(&d, &b, 8);
    return d;
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

__global__ void k_box_init(unsigned long long *keys, int n_nodes) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_nodes * 6) keys[q] = key_of((q % 6) < 3 ? 1E9 : -1E9); /* :223-224 */
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
    /* one atomic per wave and value where the whole wave sits in one node (the upper levels), else one per lane */
    const int first = __builtin_amdgcn_readfirstlane(node);
    const bool uniform = __ballot(node != first) == 0ull;
    if (uniform) {
        if (first < 0) return;
        for (int a = 0; a < 3; a++) {
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long l2 = __shfl_xor(lo[a], off, 64), h2 = __shfl_xor(hi[a], off, 64);
                if (l2 < lo[a]) lo[a] = l2;
                if (h2 > hi[a]) hi[a] = h2;
            }
        }
        if ((threadIdx.x & 63) == 0)
            for (int a = 0; a < 3; a++) {
                atomicMin(keys + (size_t)first * 6 + a, lo[a]);
                atomicMax(keys + (size_t)first * 6 + 3 + a, hi[a]);
            }
    } else if (node >= 0) {
        for (int a = 0; a < 3; a++) {
            atomicMin(keys + (size_t)node * 6 + a, lo[a]);
            atomicMax(keys + (size_t)node * 6 + 3 + a, hi[a]);
        }
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

struct TreeNode { /* host side: a node of the result, in the numbering of the reference's recursion */
    double box[6];
    int debut, fin, fg, fd, depth;
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
    double *d_verts = dev.get<double>((size_t)nv * 3);
    int32_t *d_tri = dev.get<int32_t>((size_t)nt * 3), *d_perm = dev.get<int32_t>(n), *d_perm2 = dev.get<int32_t>(n), *d_node_of = dev.get<int32_t>(n),
            *d_flag = dev.get<int32_t>(n), *d_scan = dev.get<int32_t>(n), *d_V = dev.get<int32_t>(n), *d_unres = dev.get<int32_t>(1);
    LevelNode *d_nd = dev.get<LevelNode>(n);
    unsigned long long *d_keys = dev.get<unsigned long long>((size_t)n * 6);
    size_t tmp_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, tmp_bytes, d_flag, d_scan, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream);
    void *d_tmp = dev.get<char>(tmp_bytes);
    if (!d_verts || !d_tri || !d_perm || !d_perm2 || !d_node_of || !d_flag || !d_scan || !d_V || !d_unres || !d_nd || !d_keys || !d_tmp)
        return fail(c, RT_E_NOMEM, "rt_build_bvh: device allocation failed");
    HIPCHK(c, hipMemcpyAsync(d_verts, verts, (size_t)nv * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_tri, tri_vtx, (size_t)nt * 12, hipMemcpyHostToDevice, c->stream));
    {
        std::vector<int32_t> ident((size_t)n);
        for (int i = 0; i < n; i++) ident[(size_t)i] = i;
        HIPCHK(c, hipMemcpyAsync(d_perm, ident.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    std::vector<TreeNode> tree; /* level order first; renumbered at the end */
    struct Open { int tree_id; };
    std::vector<Open> open;
    std::vector<LevelNode> lvl;
    tree.push_back(TreeNode{{0, 0, 0, 0, 0, 0}, 0, n, -1, -1, 0});
    open.push_back(Open{0});
    std::vector<unsigned long long> keys;
    std::vector<int32_t> h_perm;
    const int B = 256, G = (n + B - 1) / B;
    while (!open.empty()) {
        const int m = (int)open.size();
        lvl.assign((size_t)m, LevelNode{});
        int largest = 0;
        for (int q = 0; q < m; q++) {
            const TreeNode &t = tree[(size_t)open[(size_t)q].tree_id];
            lvl[(size_t)q].debut = t.debut; lvl[(size_t)q].fin = t.fin; lvl[(size_t)q].lead = -1;
            if (t.fin - t.debut > largest) largest = t.fin - t.debut;
        }
        HIPCHK(c, hipMemcpyAsync(d_nd, lvl.data(), (size_t)m * sizeof(LevelNode), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_node_of, dim3(G), dim3(B), 0, c->stream, d_nd, m, n, d_node_of);
        hipLaunchKernelGGL(k_box_init, dim3((m * 6 + B - 1) / B), dim3(B), 0, c->stream, d_keys, m);
        hipLaunchKernelGGL(k_box, dim3(G), dim3(B), 0, c->stream, d_verts, d_tri, d_perm, d_node_of, n, d_keys);
        keys.resize((size_t)m * 6);
        HIPCHK(c, hipMemcpyAsync(keys.data(), d_keys, (size_t)m * 48, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool need_perm = false;
        for (int q = 0; q < m; q++) {
            TreeNode &t = tree[(size_t)open[(size_t)q].tree_id];
            bool zero = false;
            for (int a = 0; a < 6; a++) { t.box[a] = value_of(keys[(size_t)q * 6 + a]); zero = zero || t.box[a] == 0.0; }
            if (zero) need_perm = true;
        }
        if (need_perm) { /* the sign of a zero minimum or maximum depends on the order of the comparisons: redo those boxes as the reference does */
            h_perm.resize((size_t)n);
            HIPCHK(c, hipMemcpy(h_perm.data(), d_perm, (size_t)n * 4, hipMemcpyDeviceToHost));
            for (int q = 0; q < m; q++) {
                TreeNode &t = tree[(size_t)open[(size_t)q].tree_id];
                bool zero = false;
                for (int a = 0; a < 6; a++) zero = zero || t.box[a] == 0.0;
                if (!zero) continue;
                for (int a = 0; a < 3; a++) { t.box[a] = 1E9; t.box[3 + a] = -1E9; }
                for (int i = t.debut; i < t.fin; i++)
                    for (int a = 0; a < 3; a++)
                        for (int v = 0; v < 3; v++) {
                            const double x = verts[(size_t)tri_vtx[3 * (size_t)h_perm[(size_t)i] + v] * 3 + a];
                            if (x < t.box[a]) t.box[a] = x;         /* std::min(a, b): b if b < a */
                            if (t.box[3 + a] < x) t.box[3 + a] = x; /* std::max(a, b): b if a < b */
                        }
            }
        }
        for (int q = 0; q < m; q++) { /* :250-267 */
            const TreeNode &t = tree[(size_t)open[(size_t)q].tree_id];
            const double dx = t.box[3] - t.box[0], dy = t.box[4] - t.box[1], dz = t.box[5] - t.box[2];
            int dim;
            if (dx >= dy && dx >= dz) dim = 0;
            else if (dy >= dx && dy >= dz) dim = 1;
            else dim = 2;
            lvl[(size_t)q].dim = dim;
            lvl[(size_t)q].milieu = (t.box[dim] + t.box[3 + dim]) * 0.5;
        }
        HIPCHK(c, hipMemcpyAsync(d_nd, lvl.data(), (size_t)m * sizeof(LevelNode), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_flags, dim3(G), dim3(B), 0, c->stream, d_verts, d_tri, d_perm, d_node_of, d_nd, n, d_flag);
        if (rocprim::exclusive_scan(d_tmp, tmp_bytes, d_flag, d_scan, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream) != hipSuccess)
            return fail(c, RT_E_DEVICE, "rt_build_bvh: prefix sum failed");
        hipLaunchKernelGGL(k_node_counts, dim3(G), dim3(B), 0, c->stream, d_flag, d_scan, d_node_of, d_nd, n);
        HIPCHK(c, hipMemcpyAsync(d_perm2, d_perm, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_build_v, dim3(G), dim3(B), 0, c->stream, d_perm, d_flag, d_scan, d_node_of, d_nd, n, d_perm2, d_V);
        int rounds = 1;
        while ((1 << rounds) < largest + 1) rounds++;
        for (int r = 0; r < rounds + 1; r++) hipLaunchKernelGGL(k_jump, dim3(G), dim3(B), 0, c->stream, d_node_of, d_nd, n, d_V);
        HIPCHK(c, hipMemsetAsync(d_unres, 0, 4, c->stream));
        hipLaunchKernelGGL(k_back, dim3(G), dim3(B), 0, c->stream, d_node_of, d_nd, n, d_V, d_perm2, d_unres);
        HIPCHK(c, hipGetLastError());
        int unres = 0;
        HIPCHK(c, hipMemcpyAsync(lvl.data(), d_nd, (size_t)m * sizeof(LevelNode), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&unres, d_unres, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (unres) return fail(c, RT_E_DEVICE, "rt_build_bvh: internal error (unresolved queue entry)");
        std::swap(d_perm, d_perm2);
        std::vector<Open> next;
        for (int q = 0; q < m; q++) { /* :283 */
            const int id = open[(size_t)q].tree_id;
            const int debut = tree[(size_t)id].debut, fin = tree[(size_t)id].fin, pivot = debut + lvl[(size_t)q].n_s;
            if (pivot == debut || pivot == fin || fin - debut < 5) continue;
            const int l = (int)tree.size();
            tree.push_back(TreeNode{{0, 0, 0, 0, 0, 0}, debut, pivot, -1, -1, tree[(size_t)id].depth + 1});
            tree.push_back(TreeNode{{0, 0, 0, 0, 0, 0}, pivot, fin, -1, -1, tree[(size_t)id].depth + 1});
            tree[(size_t)id].fg = l;
            tree[(size_t)id].fd = l + 1;
            next.push_back(Open{l});
            next.push_back(Open{l + 1});
        }
        open.swap(next);
    }
    if ((int64_t)tree.size() > node_cap) return fail(c, RT_E_INVALID, "rt_build_bvh: %zu nodes do not fit the caller's %lld", tree.size(), (long long)node_cap);
    HIPCHK(c, hipMemcpy(order, d_perm, (size_t)n * 4, hipMemcpyDeviceToHost));
    /* the numbering of the reference's recursion: a node's two children get the next two numbers when it is split, and the
     * left subtree is completed before the right one */
    {
        std::vector<int> newid(tree.size(), -1), stack;
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
        for (size_t id = 0; id < tree.size(); id++) {
            const TreeNode &t = tree[id];
            rt_build_node &o = nodes[(size_t)newid[id]];
            for (int a = 0; a < 6; a++) o.box[a] = t.box[a];
            o.debut = t.debut; o.fin = t.fin; o.depth = t.depth; o.pad = 0;
            o.fg = t.fg >= 0 ? newid[(size_t)t.fg] : -1;
            o.fd = t.fd >= 0 ? newid[(size_t)t.fd] : -1;
        }
    }
    *n_nodes_out = (int64_t)tree.size();
    return RT_OK;
}
