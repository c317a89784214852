// Device-side build of the 8-wide quantised BVH (bvh8.h) for large scenes.
//
// The reference leaves the build to Embree inside rtcCommitScene (src/raytracer.cpp:172-175), on the
// CPU.  The host builder of this library (bvh_build.cpp, binned SAH) gives the better tree and is the
// default; for tens of millions of triangles it takes tens of seconds, and this builder takes about one:
//   1. triangle boxes and centroid bounds                       (k_prim, k_bounds)
//   2. 63-bit Morton codes of the centroids, radix sort         (k_morton, rocPRIM)
//   3. binary tree over the sorted order: parallel locally-ordered clustering (Meister and Bittner 2018) - every
//      cluster looks for the neighbour within a window of the current order whose union with it has the smallest
//      surface area, mutual choices merge, the survivors are compacted, until one cluster is left
//      (k_ploc_nearest, k_ploc_merge, k_ploc_compact); WAPT_DEVICE_BVH=lbvh selects the older, faster and
//      lower-quality binary radix tree over the codes (Karras 2012: k_hierarchy, boxes bottom-up in k_refit)
//   4. collapse to 8-wide nodes level by level, with the same child-opening rule, slot assignment and
//      conservative quantisation as the host builder; leaves are subtrees of at most 3 triangles   (k_collapse)
// The result lives in device memory in the layout the traversal kernels read.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bvh8.h"
#include "bvh_device.h"

namespace {

#define DCHK(call)                                                                         \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            *err = std::string(#call) + ": " + hipGetErrorString(_e);                      \
            return false;                                                                  \
        }                                                                                  \
    } while (0)

constexpr uint32_t LEAF = 0x80000000u;  // reference to sorted triangle k: LEAF | k; else inner node index

struct Box6 {
    float lo[3], hi[3];
};

__device__ __forceinline__ void box_reset(Box6& b) {
    for (int a = 0; a < 3; a++) {
        b.lo[a] = FLT_MAX;
        b.hi[a] = -FLT_MAX;
    }
}
__device__ __forceinline__ void box_grow(Box6& b, const Box6& o) {
    for (int a = 0; a < 3; a++) {
        b.lo[a] = fminf(b.lo[a], o.lo[a]);
        b.hi[a] = fmaxf(b.hi[a], o.hi[a]);
    }
}
__device__ __forceinline__ float box_area(const Box6& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

// Triangle boxes; per block the bounds of the centroids and a flag for non-finite input.
__global__ void k_prim(const float* __restrict__ verts, uint32_t n, Box6* __restrict__ tbox, Box6* __restrict__ block_bounds,
                       uint32_t* __restrict__ bad) {
    __shared__ Box6 s_b[256];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Box6 cb;
    box_reset(cb);
    if (t < n) {
        const float* v = verts + 9 * (size_t)t;
        Box6 b;
        box_reset(b);
        bool fin = true;
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) {
                float x = v[3 * k + a];
                fin = fin && isfinite(x);
                b.lo[a] = fminf(b.lo[a], x);
                b.hi[a] = fmaxf(b.hi[a], x);
            }
        if (!fin) atomicOr(bad, 1u);
        tbox[t] = b;
        for (int a = 0; a < 3; a++) cb.lo[a] = cb.hi[a] = 0.5f * (b.lo[a] + b.hi[a]);
    }
    s_b[threadIdx.x] = cb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) box_grow(s_b[threadIdx.x], s_b[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_bounds[blockIdx.x] = s_b[0];
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(const Box6* __restrict__ tbox, uint32_t n, Box6 cb, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Box6 b = tbox[t];
    uint64_t q[3];
    for (int a = 0; a < 3; a++) {
        float ext = cb.hi[a] - cb.lo[a];
        float c = 0.5f * (b.lo[a] + b.hi[a]);
        double u = ext > 0.f ? ((double)c - cb.lo[a]) / (double)ext : 0.0;
        u = fmin(fmax(u, 0.0), 1.0);
        q[a] = (uint64_t)fmin(u * 2097152.0, 2097151.0);
    }
    keys[t] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[t] = t;
}

// Length of the common prefix of sorted keys i and j (equal keys are told apart by their position).
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, long long n, long long i, long long j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clzll((unsigned long long)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Binary radix tree (Karras 2012): inner node i in [0, n-1) covers a contiguous range of the sorted order.
__global__ void k_hierarchy(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                            uint32_t* __restrict__ first, uint32_t* __restrict__ count, uint32_t* __restrict__ parent_inner,
                            uint32_t* __restrict__ parent_leaf) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = n;
    if (i >= N - 1) return;
    int d = (common_prefix(keys, N, i, i + 1) - common_prefix(keys, N, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = common_prefix(keys, N, i, i - d);
    long long lmax = 2;
    while (common_prefix(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    long long l = 0;
    for (long long t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(keys, N, i, i + (l + t) * d) > dmin) l += t;
    long long j = i + l * d;
    int dnode = common_prefix(keys, N, i, j);
    long long s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (common_prefix(keys, N, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    long long gamma = i + s * d + (d < 0 ? -1 : 0);
    long long lo = i < j ? i : j, hi = i < j ? j : i;
    uint32_t L = (lo == gamma) ? (LEAF | (uint32_t)gamma) : (uint32_t)gamma;
    uint32_t R = (hi == gamma + 1) ? (LEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = L;
    right[i] = R;
    first[i] = (uint32_t)lo;
    count[i] = (uint32_t)(hi - lo + 1);
    if (L & LEAF) parent_leaf[L & ~LEAF] = (uint32_t)i;
    else parent_inner[L] = (uint32_t)i;
    if (R & LEAF) parent_leaf[R & ~LEAF] = (uint32_t)i;
    else parent_inner[R] = (uint32_t)i;
    if (i == 0) parent_inner[0] = 0xffffffffu;
}

// Boxes of the inner nodes: every leaf walks up; the second thread to arrive at a node computes its
// box from the two children (the first one stops).
__global__ void k_refit(const Box6* __restrict__ tbox, const uint32_t* __restrict__ sorted, uint32_t n,
                        const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                        const uint32_t* __restrict__ parent_inner, const uint32_t* __restrict__ parent_leaf, Box6* ibox,
                        uint32_t* flags) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t node = parent_leaf[k];
    while (node != 0xffffffffu) {
        __threadfence();
        if (atomicAdd(&flags[node], 1u) == 0u) return;
        __threadfence();
        Box6 b;
        box_reset(b);
        uint32_t c[2] = {left[node], right[node]};
        for (int s = 0; s < 2; s++) {
            if (c[s] & LEAF) {
                box_grow(b, tbox[sorted[c[s] & ~LEAF]]);
            } else {
                // written by another thread of this launch, made visible by its fence + the flag atomic
                Box6 o = ibox[c[s]];
                box_grow(b, o);
            }
        }
        ibox[node] = b;
        node = parent_inner[node];
    }
}

struct Tree2 {
    const Box6* tbox;
    const uint32_t* sorted;
    const uint32_t *left, *right, *first, *count;
    const Box6* ibox;
};

__device__ __forceinline__ uint32_t ref_count(const Tree2& T, uint32_t r) { return (r & LEAF) ? 1u : T.count[r]; }
__device__ __forceinline__ uint32_t ref_first(const Tree2& T, uint32_t r) { return (r & LEAF) ? (r & ~LEAF) : T.first[r]; }
__device__ __forceinline__ Box6 ref_box(const Tree2& T, uint32_t r) { return (r & LEAF) ? T.tbox[T.sorted[r & ~LEAF]] : T.ibox[r]; }

// The (at most three) triangles under a small subtree, as indices into the sorted order.
__device__ __forceinline__ uint32_t ref_gather(const Tree2& T, uint32_t r, uint32_t out[3]) {
    if (r & LEAF) {
        out[0] = r & ~LEAF;
        return 1u;
    }
    uint32_t n = 0;
    const uint32_t c[2] = {T.left[r], T.right[r]};
    for (int s = 0; s < 2; s++) {
        if (c[s] & LEAF) {
            if (n < 3u) out[n] = c[s] & ~LEAF;
            n++;
        } else {  // (a subtree of three triangles has at most one inner child, whose children are leaves)
            const uint32_t g[2] = {T.left[c[s]], T.right[c[s]]};
            for (int k = 0; k < 2; k++) {
                if (n < 3u) out[n] = g[k] & ~LEAF;
                n++;
            }
        }
    }
    return n;
}

// ---- parallel locally-ordered clustering ----------------------------------------------------------------------

// nn[i]: the cluster within `radius` positions of i whose union with cluster i has the smallest surface area
// (ties go to the partner by position i ^ 1, else to the lower position).
// radius 0: neighbours are paired by position (2k with 2k+1) - the safety net for a round that merged next to nothing.
__global__ void k_ploc_nearest(const Box6* __restrict__ cbox, uint32_t m, int radius, uint32_t* __restrict__ nn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (radius == 0) {
        const uint32_t j = i ^ 1u;
        nn[i] = j < m ? j : i;
        return;
    }
    const Box6 b = cbox[i];
    float best = FLT_MAX;
    uint32_t bj = i;
    const long long lo = max((long long)i - radius, 0ll), hi = min((long long)i + radius, (long long)m - 1);
    for (long long j = lo; j <= hi; j++) {
        if (j == (long long)i) continue;
        Box6 u = cbox[j];
        box_grow(u, b);
        const float a = box_area(u);
        // (equal areas: the partner by position wins, so that a run of indistinguishable clusters - coincident
        //  triangles - pairs up 2k with 2k+1 instead of all choosing the same neighbour)
        if (a < best || (a == best && (uint32_t)j == (i ^ 1u))) {
            best = a;
            bj = (uint32_t)j;
        }
    }
    nn[i] = bj;
}

// Mutual nearest neighbours become one cluster (a new inner node) at the lower of the two positions.
__global__ void k_ploc_merge(const uint32_t* __restrict__ C, const Box6* __restrict__ cbox, const uint32_t* __restrict__ nn, uint32_t m,
                             uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ count, Box6* __restrict__ ibox,
                             uint32_t* __restrict__ node_counter, uint32_t* __restrict__ outC, Box6* __restrict__ outBox,
                             uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    if (mutual && i > j) {
        keep[i] = 0u;
        return;
    }
    keep[i] = 1u;
    if (!mutual) {
        outC[i] = C[i];
        outBox[i] = cbox[i];
        return;
    }
    const uint32_t a = C[i], b = C[j];
    const uint32_t idx = atomicAdd(node_counter, 1u);
    left[idx] = a;
    right[idx] = b;
    count[idx] = ((a & LEAF) ? 1u : count[a]) + ((b & LEAF) ? 1u : count[b]);
    Box6 u = cbox[i];
    box_grow(u, cbox[j]);
    ibox[idx] = u;
    outC[i] = idx;
    outBox[i] = u;
}

__global__ void k_ploc_compact(const uint32_t* __restrict__ inC, const Box6* __restrict__ inBox, const uint32_t* __restrict__ keep,
                               const uint32_t* __restrict__ pos, uint32_t m, uint32_t* __restrict__ C, Box6* __restrict__ cbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !keep[i]) return;
    C[pos[i]] = inC[i];
    cbox[pos[i]] = inBox[i];
}

__global__ void k_ploc_init(const Box6* __restrict__ tbox, const uint32_t* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ C,
                            Box6* __restrict__ cbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    C[i] = LEAF | i;
    cbox[i] = tbox[sorted[i]];
}

struct CollapseOut {
    Bvh8Node* nodes;
    Bvh8Tri* tris;
    uint32_t* node_counter;  // next free wide node
    uint32_t* tri_counter;   // next free triangle slot
    uint32_t node_cap, tri_cap;
    double* sah;
};

// One wide node per thread: refs[i] is the binary subtree that becomes wide node level_start + i.
__global__ void k_collapse(Tree2 T, const uint32_t* __restrict__ refs, uint32_t n_refs, uint32_t level_start,
                           uint32_t* __restrict__ next_refs, CollapseOut out, const float* __restrict__ verts,
                           const int32_t* __restrict__ obj, const int32_t* __restrict__ prim, const uint8_t* __restrict__ opaque,
                           float root_area, uint32_t* __restrict__ fail, uint32_t leaf_max) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_refs) return;
    const uint32_t me = refs[i];
    uint32_t ch[8];
    Box6 cbx[8];
    int nch = 0;
    if (ref_count(T, me) <= leaf_max) {
        ch[nch++] = me;  // the whole tree is one leaf
    } else {
        ch[nch++] = T.left[me];
        ch[nch++] = T.right[me];
        for (int k = 0; k < 2; k++) cbx[k] = ref_box(T, ch[k]);
        while (nch < 8) {
            int bi = -1;
            float ba = -1.f;
            for (int k = 0; k < nch; k++) {
                if (ref_count(T, ch[k]) > leaf_max) {
                    float a = box_area(cbx[k]);
                    if (a > ba) {
                        ba = a;
                        bi = k;
                    }
                }
            }
            if (bi < 0) break;
            uint32_t c = ch[bi];
            ch[bi] = T.left[c];
            cbx[bi] = ref_box(T, ch[bi]);
            ch[nch] = T.right[c];
            cbx[nch] = ref_box(T, ch[nch]);
            nch++;
        }
    }
    if (nch == 1) cbx[0] = ref_box(T, ch[0]);
    Box6 nb;
    box_reset(nb);
    for (int k = 0; k < nch; k++) box_grow(nb, cbx[k]);
    double sah = (double)box_area(nb) / root_area;
    float nc[3];
    for (int a = 0; a < 3; a++) nc[a] = 0.5f * (nb.lo[a] + nb.hi[a]);

    // slot assignment: greedy on the child-centre x octant-direction table
    int child_in_slot[8];
    bool placed[8];
    for (int s = 0; s < 8; s++) child_in_slot[s] = -1;
    for (int k = 0; k < 8; k++) placed[k] = false;
    for (int round = 0; round < nch; round++) {
        float best = -FLT_MAX;
        int bi = -1, bs = -1;
        for (int k = 0; k < nch; k++) {
            if (placed[k]) continue;
            float d0 = 0.5f * (cbx[k].lo[0] + cbx[k].hi[0]) - nc[0];
            float d1 = 0.5f * (cbx[k].lo[1] + cbx[k].hi[1]) - nc[1];
            float d2 = 0.5f * (cbx[k].lo[2] + cbx[k].hi[2]) - nc[2];
            for (int s = 0; s < 8; s++) {
                if (child_in_slot[s] >= 0) continue;
                float c = d0 * ((s & 4) ? 1.f : -1.f) + d1 * ((s & 2) ? 1.f : -1.f) + d2 * ((s & 1) ? 1.f : -1.f);
                if (c > best) {
                    best = c;
                    bi = k;
                    bs = s;
                }
            }
        }
        if (bi < 0) {  // (no comparison succeeded: take the first child and slot that are left)
            for (int k = nch - 1; k >= 0; k--)
                if (!placed[k]) bi = k;
            for (int s = 7; s >= 0; s--)
                if (child_in_slot[s] < 0) bs = s;
        }
        placed[bi] = true;
        child_in_slot[bs] = bi;
    }

    Bvh8Node node;
    memset(&node, 0, sizeof(node));
    float cell[3];
    for (int a = 0; a < 3; a++) {
        node.p[a] = nb.lo[a];
        float ext = nb.hi[a] - nb.lo[a];
        int e = -125;
        if (ext > 0.f) {
            e = (int)ceil(log2((double)ext / 255.0));
            while (ldexp(255.0, e) < (double)ext) e++;  // log2 rounding: 255 cells must cover the extent
        }
        e = max(e, -125);
        e = min(e, 126);
        node.e[a] = (uint8_t)(e + 127);
        cell[a] = ldexpf(1.0f, e);
    }
    uint32_t n_inner = 0, n_tri = 0;
    for (int k = 0; k < nch; k++) {
        uint32_t c = ref_count(T, ch[k]);
        if (c > leaf_max) n_inner++;
        else n_tri += c;
    }
    uint32_t base = n_inner ? atomicAdd(out.node_counter, n_inner) : 0u;
    uint32_t tbase = n_tri ? atomicAdd(out.tri_counter, n_tri) : 0u;
    if ((n_inner && (unsigned long long)base + n_inner > out.node_cap) || (n_tri && (unsigned long long)tbase + n_tri > out.tri_cap)) {
        atomicOr(fail, 1u);
        return;
    }
    node.child_base = base;
    node.tri_base = tbase;
    uint8_t* qlo[3] = {node.qlo_x, node.qlo_y, node.qlo_z};
    uint8_t* qhi[3] = {node.qhi_x, node.qhi_y, node.qhi_z};
    uint32_t tri_off = 0, inner_k = 0;
    for (int s = 0; s < 8; s++) {
        int k = child_in_slot[s];
        if (k < 0) {
            node.meta[s] = 0;
            for (int a = 0; a < 3; a++) {
                qlo[a][s] = 255;
                qhi[a][s] = 0;
            }
            continue;
        }
        const Box6& cb = cbx[k];
        for (int a = 0; a < 3; a++) {
            double lo = ((double)cb.lo[a] - (double)node.p[a]) / (double)cell[a];
            double hi = ((double)cb.hi[a] - (double)node.p[a]) / (double)cell[a];
            int ql = (int)floor(lo), qh = (int)ceil(hi);
            ql = min(max(ql, 0), 255);
            qh = min(max(qh, 0), 255);
            // keep the box conservative under the float evaluation p + q*cell of the traversal
            while (ql > 0 && node.p[a] + (float)ql * cell[a] > cb.lo[a]) ql--;
            while (qh < 255 && node.p[a] + (float)qh * cell[a] < cb.hi[a]) qh++;
            qlo[a][s] = (uint8_t)ql;
            qhi[a][s] = (uint8_t)qh;
        }
        uint32_t c = ref_count(T, ch[k]);
        if (c > leaf_max) {
            node.imask |= (uint8_t)(1u << s);
            node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
            next_refs[base - (level_start + n_refs) + inner_k] = ch[k];
            inner_k++;
        } else {
            const uint8_t unary[4] = {0, 1, 3, 7};
            node.meta[s] = (uint8_t)((unary[c] << 5) | tri_off);
            uint32_t which[3] = {0u, 0u, 0u};
            ref_gather(T, ch[k], which);
            for (uint32_t j = 0; j < c; j++) {
                uint32_t t = T.sorted[which[j]];
                const float* v = verts + 9 * (size_t)t;
                Bvh8Tri tr;
                for (int a = 0; a < 3; a++) {
                    tr.v0[a] = v[a];
                    tr.v1[a] = v[3 + a];
                    tr.v2[a] = v[6 + a];
                }
                tr.obj = obj[t];
                tr.prim = prim ? prim[t] : (int32_t)t;
                tr.pad = (uint32_t)opaque[tr.obj] | ((uint32_t)tr.obj << 8);
                out.tris[tbase + tri_off + j] = tr;
            }
            tri_off += c;
            sah += (double)box_area(cb) / root_area * c;
        }
    }
    out.nodes[level_start + i] = node;
    atomicAdd(out.sah, sah);
}

struct Buffers {
    std::vector<void*> ptrs;
    ~Buffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    bool alloc(T** p, size_t count, std::string* err) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) {
            *err = std::string("hipMalloc: ") + hipGetErrorString(e);
            return false;
        }
        ptrs.push_back(q);
        *p = (T*)q;
        return true;
    }
};

}  // namespace

bool bvh8_build_device(const float* verts, const int32_t* obj, const int32_t* prim, uint64_t n64, const uint8_t* obj_opaque,
                       uint32_t n_objects, hipStream_t st, DeviceBvh8* out, std::string* err) {
    auto t0 = std::chrono::steady_clock::now();
    if (n64 == 0) {
        *err = "scene has no triangles";
        return false;
    }
    if (n64 >= (1ull << 31)) {
        *err = "more than 2^31 triangles";
        return false;
    }
    const uint32_t n = (uint32_t)n64;
    const uint32_t nb = (n + 255) / 256;
    Buffers B;
    float* d_verts;
    int32_t *d_obj, *d_prim = nullptr;
    uint8_t* d_opaque;
    Box6 *d_tbox, *d_bb, *d_ibox;
    uint32_t *d_bad, *d_vals_in, *d_vals, *d_left, *d_right, *d_first, *d_count, *d_pin, *d_pleaf, *d_flags, *d_refs_a, *d_refs_b,
        *d_counters;
    uint64_t *d_keys_in, *d_keys;
    double* d_sah;
    if (!B.alloc(&d_verts, (size_t)n * 9, err) || !B.alloc(&d_obj, n, err) || !B.alloc(&d_opaque, n_objects, err) ||
        !B.alloc(&d_tbox, n, err) || !B.alloc(&d_bb, nb, err) || !B.alloc(&d_ibox, n, err) || !B.alloc(&d_bad, 4, err) ||
        !B.alloc(&d_vals_in, n, err) || !B.alloc(&d_vals, n, err) || !B.alloc(&d_keys_in, n, err) || !B.alloc(&d_keys, n, err) ||
        !B.alloc(&d_left, n, err) || !B.alloc(&d_right, n, err) || !B.alloc(&d_first, n, err) || !B.alloc(&d_count, n, err) ||
        !B.alloc(&d_pin, n, err) || !B.alloc(&d_pleaf, n, err) || !B.alloc(&d_flags, n, err) || !B.alloc(&d_refs_a, n, err) ||
        !B.alloc(&d_refs_b, n, err) || !B.alloc(&d_counters, 4, err) || !B.alloc(&d_sah, 1, err))
        return false;
    if (prim && !B.alloc(&d_prim, n, err)) return false;
    DCHK(hipMemcpyAsync(d_verts, verts, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice, st));
    DCHK(hipMemcpyAsync(d_obj, obj, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (prim) DCHK(hipMemcpyAsync(d_prim, prim, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    DCHK(hipMemcpyAsync(d_opaque, obj_opaque, n_objects, hipMemcpyHostToDevice, st));
    DCHK(hipMemsetAsync(d_bad, 0, 16, st));
    DCHK(hipMemsetAsync(d_flags, 0, (size_t)n * 4, st));
    DCHK(hipMemsetAsync(d_sah, 0, 8, st));

    // 1. boxes, centroid bounds
    hipLaunchKernelGGL(k_prim, dim3(nb), dim3(256), 0, st, d_verts, n, d_tbox, d_bb, d_bad);
    std::vector<Box6> hb(nb);
    uint32_t bad = 0;
    DCHK(hipMemcpyAsync(hb.data(), d_bb, (size_t)nb * sizeof(Box6), hipMemcpyDeviceToHost, st));
    DCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    if (bad) {
        *err = "non-finite vertex coordinate";
        return false;
    }
    Box6 cb;
    for (int a = 0; a < 3; a++) {
        cb.lo[a] = FLT_MAX;
        cb.hi[a] = -FLT_MAX;
    }
    for (const Box6& b : hb)
        for (int a = 0; a < 3; a++) {
            cb.lo[a] = std::min(cb.lo[a], b.lo[a]);
            cb.hi[a] = std::max(cb.hi[a], b.hi[a]);
        }

    // 2. Morton codes, sort
    hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, st, d_tbox, n, cb, d_keys_in, d_vals_in);
    {
        size_t tmp_bytes = 0;
        DCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in, d_keys, d_vals_in, d_vals, (size_t)n, 0u, 63u, st));
        char* d_tmp;
        if (!B.alloc(&d_tmp, tmp_bytes, err)) return false;
        DCHK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys_in, d_keys, d_vals_in, d_vals, (size_t)n, 0u, 63u, st));
    }

    // 3. binary tree over the sorted order, with the boxes of its inner nodes
    uint32_t root_ref = LEAF | 0u;
    const char* kind = getenv("WAPT_DEVICE_BVH");
    const bool lbvh = kind && !strcmp(kind, "lbvh");
    if (n >= 2 && lbvh) {
        hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + 255) / 256), dim3(256), 0, st, d_keys, n, d_left, d_right, d_first, d_count,
                           d_pin, d_pleaf);
        hipLaunchKernelGGL(k_refit, dim3(nb), dim3(256), 0, st, d_tbox, d_vals, n, d_left, d_right, d_pin, d_pleaf, d_ibox, d_flags);
        root_ref = 0u;
    } else if (n >= 2) {
        int radius = 8;
        if (const char* e = getenv("WAPT_PLOC_RADIUS")) radius = std::min(64, std::max(1, atoi(e)));
        Box6 *d_cb_a, *d_cb_b;
        uint32_t *d_nn, *d_keep, *d_pos, *d_nodes_made;
        if (!B.alloc(&d_cb_a, n, err) || !B.alloc(&d_cb_b, n, err) || !B.alloc(&d_nn, n, err) || !B.alloc(&d_keep, n, err) ||
            !B.alloc(&d_pos, n, err) || !B.alloc(&d_nodes_made, 4, err))
            return false;
        size_t scan_bytes = 0;
        DCHK(rocprim::exclusive_scan(nullptr, scan_bytes, d_keep, d_pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        char* d_scan_tmp;
        if (!B.alloc(&d_scan_tmp, scan_bytes, err)) return false;
        DCHK(hipMemsetAsync(d_nodes_made, 0, 16, st));
        // the cluster lists alternate between (refs_a, cb_a) and (refs_b, cb_b)
        uint32_t *C = d_refs_a, *C2 = d_refs_b;
        Box6 *cb1 = d_cb_a, *cb2 = d_cb_b;
        hipLaunchKernelGGL(k_ploc_init, dim3(nb), dim3(256), 0, st, d_tbox, d_vals, n, C, cb1);
        uint32_t m = n;
        int rounds = 0;
        bool stalled = false;  // the last round merged next to nothing (never seen on real scenes: rounds remove 15-40 %)
        while (m > 1) {
            if (++rounds > 4096) {
                *err = "clustering did not converge";
                return false;
            }
            const dim3 g((m + 255) / 256);
            hipLaunchKernelGGL(k_ploc_nearest, g, dim3(256), 0, st, cb1, m, stalled ? 0 : radius, d_nn);
            hipLaunchKernelGGL(k_ploc_merge, g, dim3(256), 0, st, C, cb1, d_nn, m, d_left, d_right, d_count, d_ibox, d_nodes_made, C2, cb2,
                               d_keep);
            DCHK(rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_keep, d_pos, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(k_ploc_compact, g, dim3(256), 0, st, C2, cb2, d_keep, d_pos, m, C, cb1);
            uint32_t made = 0;
            DCHK(hipMemcpyAsync(&made, d_nodes_made, 4, hipMemcpyDeviceToHost, st));
            DCHK(hipStreamSynchronize(st));
            const uint32_t m_new = n - made;  // every merge removes one cluster
            if (m_new >= m) {
                *err = "clustering made no progress";
                return false;
            }
            // (shorter lists too, down to 64 clusters: a run of nested or growing boxes merges one pair per round and builds
            //  a chain as deep as the list is long.  The bar stays far below what an ordinary round removes - 15-40 % -
            //  because pairing by position costs tree quality: with "less than a sixteenth" for short lists the last
            //  rounds of every build were paired that way and dragonChrome's rays visited 40 % more nodes)
            stalled = m > 1024u ? (m - m_new) < m / 256u : (m > 64u && (m - m_new) * 64u < m);
            if (getenv("WAPT_PLOC_VERBOSE")) fprintf(stderr, "ploc round %d: %u -> %u clusters%s\n", rounds, m, m_new, stalled ? " (stalled)" : "");
            m = m_new;
        }
        DCHK(hipMemcpyAsync(&root_ref, C, 4, hipMemcpyDeviceToHost, st));
        DCHK(hipStreamSynchronize(st));
    }
    Box6 root_box;
    if (n >= 2) DCHK(hipMemcpyAsync(&root_box, d_ibox + (root_ref & ~LEAF), sizeof(Box6), hipMemcpyDeviceToHost, st));
    else DCHK(hipMemcpyAsync(&root_box, d_tbox, sizeof(Box6), hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    DCHK(hipGetLastError());
    for (int a = 0; a < 3; a++) {
        out->lo[a] = root_box.lo[a];
        out->hi[a] = root_box.hi[a];
    }
    float root_area;
    {
        float dx = root_box.hi[0] - root_box.lo[0], dy = root_box.hi[1] - root_box.lo[1], dz = root_box.hi[2] - root_box.lo[2];
        root_area = std::max(2.f * (dx * dy + dy * dz + dz * dx), 1e-30f);
    }

    // 5. collapse, one level per launch
    // (a subtree of at most leaf_max triangles becomes a leaf; WAPT_DEVICE_LEAF_MAX = 1..3 for experiments)
    uint32_t leaf_max = 3u;
    if (const char* e = getenv("WAPT_DEVICE_LEAF_MAX")) leaf_max = (uint32_t)std::min(3, std::max(1, atoi(e)));
    const uint32_t node_cap = std::max(n, 2u);  // every wide node is a distinct inner node of the binary tree
    Bvh8Node* d_nodes = nullptr;
    Bvh8Tri* d_tris = nullptr;
    DCHK(hipMalloc((void**)&d_nodes, (size_t)node_cap * sizeof(Bvh8Node)));
    hipError_t e2 = hipMalloc((void**)&d_tris, (size_t)n * sizeof(Bvh8Tri));
    if (e2 != hipSuccess) {
        (void)hipFree(d_nodes);
        *err = std::string("hipMalloc: ") + hipGetErrorString(e2);
        return false;
    }
    auto fail = [&](const std::string& m) {
        (void)hipFree(d_nodes);
        (void)hipFree(d_tris);
        *err = m;
        return false;
    };
    uint32_t h_counters[4] = {1u, 0u, 0u, 0u};  // nodes used, triangles used, failure flag
    if (hipMemcpyAsync(d_counters, h_counters, 16, hipMemcpyHostToDevice, st) != hipSuccess) return fail("copy failed");
    if (hipMemcpyAsync(d_refs_a, &root_ref, 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail("copy failed");
    Tree2 T{d_tbox, d_vals, d_left, d_right, d_first, d_count, d_ibox};
    CollapseOut co{d_nodes, d_tris, d_counters, d_counters + 1, node_cap, n, d_sah};
    uint32_t level_start = 0, level_count = 1, depth = 0;
    uint32_t *cur = d_refs_a, *nxt = d_refs_b;
    while (level_count > 0) {
        depth++;
        if (depth > 64) return fail("binary tree too deep to collapse");
        hipLaunchKernelGGL(k_collapse, dim3((level_count + 127) / 128), dim3(128), 0, st, T, cur, level_count, level_start, nxt, co,
                           d_verts, d_obj, d_prim, d_opaque, root_area, d_counters + 2, leaf_max);
        if (hipMemcpyAsync(h_counters, d_counters, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return fail(std::string("collapse: ") + hipGetErrorString(hipGetLastError()));
        if (h_counters[2]) return fail("collapse ran out of node or triangle space");
        uint32_t total = h_counters[0];
        level_start += level_count;
        level_count = total - level_start;
        std::swap(cur, nxt);
    }
    if (h_counters[1] != n) return fail("collapse lost triangles: " + std::to_string(h_counters[1]) + " of " + std::to_string(n));
    double sah = 0.0;
    (void)hipMemcpy(&sah, d_sah, 8, hipMemcpyDeviceToHost);

    // shrink the node array to what is used
    const uint32_t n_nodes = h_counters[0];
    Bvh8Node* d_exact = nullptr;
    if (hipMalloc((void**)&d_exact, (size_t)n_nodes * sizeof(Bvh8Node)) == hipSuccess) {
        (void)hipMemcpyAsync(d_exact, d_nodes, (size_t)n_nodes * sizeof(Bvh8Node), hipMemcpyDeviceToDevice, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(d_nodes);
        d_nodes = d_exact;
    }
    out->d_nodes = d_nodes;
    out->d_tris = d_tris;
    out->n_nodes = n_nodes;
    out->n_tris = n;
    out->max_depth = depth;
    out->sah_cost = (float)sah;
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}
