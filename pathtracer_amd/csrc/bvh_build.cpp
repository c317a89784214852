// Host-side builder of the 8-wide quantised BVH (format: bvh8.h).
//
//   1. binary BVH by binned SAH (16 bins per axis), leaves of at most 3
//      triangles, built with one thread per large subtree;
//   2. collapse to 8-wide nodes by repeatedly opening the child with the
//      largest surface area;
//   3. slot assignment: a child lands in the slot whose octant direction
//      best matches its offset from the node centre, which makes
//      (slot ^ inverse ray octant) a front-to-back order for the traversal;
//   4. conservative 8-bit quantisation of the child boxes on a per-node
//      power-of-two grid.
//
// The reference delegates all of this to Embree (rtcCommitScene,
// src/raytracer.cpp:172-175); nothing here is derived from it.
#include "bvh8.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = FLT_MAX;
            hi[a] = -FLT_MAX;
        }
    }
    void grow(const float* p) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Node2 {
    Box box;
    uint32_t left = 0, right = 0;   // children when count == 0
    uint32_t first = 0, count = 0;  // triangle range in the index array when a leaf
};

struct Builder2 {
    const float* verts;
    uint64_t n;
    std::vector<Box> tbox;
    std::vector<float> cent;  // 3 per triangle
    std::vector<uint32_t> idx;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> next{0};
    std::atomic<int> threads_free{0};

    static constexpr int BINS = 16;
    float C_TRAV = 0.5f;   // cost of a node visit relative to a triangle test
    int max_leaf = 3;      // the leaf record counts triangles in unary in 3 bits

    void build_range(uint32_t node, uint32_t b, uint32_t e) {
        // iterative on the right child, recursive (or threaded) on the left
        std::vector<std::thread> spawned;
        for (;;) {
            Node2& N = nodes[node];
            Box bb, cb;
            bb.reset();
            cb.reset();
            for (uint32_t i = b; i < e; i++) {
                bb.grow(tbox[idx[i]]);
                cb.grow(&cent[3 * (size_t)idx[i]]);
            }
            N.box = bb;
            uint32_t cnt = e - b;
            if (cnt == 1) {
                N.first = b;
                N.count = 1;
                break;
            }
            // binned SAH
            float best = FLT_MAX;
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.f)) continue;
                Box bins[BINS];
                uint32_t bc[BINS];
                for (int k = 0; k < BINS; k++) {
                    bins[k].reset();
                    bc[k] = 0;
                }
                float scale = BINS / ext;
                for (uint32_t i = b; i < e; i++) {
                    uint32_t t = idx[i];
                    int k = (int)((cent[3 * (size_t)t + a] - cb.lo[a]) * scale);
                    k = std::min(std::max(k, 0), BINS - 1);
                    bins[k].grow(tbox[t]);
                    bc[k]++;
                }
                float ra[BINS];
                uint32_t rc[BINS];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int k = BINS - 1; k > 0; k--) {
                    acc.grow(bins[k]);
                    c += bc[k];
                    ra[k] = acc.area();
                    rc[k] = c;
                }
                acc.reset();
                c = 0;
                for (int k = 0; k < BINS - 1; k++) {
                    acc.grow(bins[k]);
                    c += bc[k];
                    if (c == 0 || rc[k + 1] == 0) continue;
                    float cost = acc.area() * c + ra[k + 1] * rc[k + 1];
                    if (cost < best) {
                        best = cost;
                        best_axis = a;
                        best_bin = k;
                    }
                }
            }
            float area = bb.area();
            if (cnt <= (uint32_t)max_leaf) {
                float leaf_cost = area * cnt;
                float split_cost = (best_axis >= 0) ? C_TRAV * area + best : FLT_MAX;
                if (leaf_cost <= split_cost) {
                    N.first = b;
                    N.count = cnt;
                    break;
                }
            }
            uint32_t mid;
            if (best_axis >= 0) {
                float ext = cb.hi[best_axis] - cb.lo[best_axis];
                float scale = BINS / ext;
                float lo = cb.lo[best_axis];
                int a = best_axis;
                auto it = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t t) {
                    int k = (int)((cent[3 * (size_t)t + a] - lo) * scale);
                    k = std::min(std::max(k, 0), BINS - 1);
                    return k <= best_bin;
                });
                mid = (uint32_t)(it - idx.begin());
            } else {
                mid = b + cnt / 2;  // coincident centroids: split by index
            }
            if (mid == b || mid == e) mid = b + cnt / 2;
            uint32_t l = next.fetch_add(2);
            uint32_t r = l + 1;
            N.left = l;
            N.right = r;
            N.count = 0;
            // left: thread if large and a thread slot is free, else recurse
            uint32_t lcnt = mid - b;
            if (lcnt > 100000 && threads_free.fetch_sub(1) > 0) {
                spawned.emplace_back([this, l, b, mid]() {
                    build_range(l, b, mid);
                    threads_free.fetch_add(1);
                });
            } else {
                if (lcnt > 100000) threads_free.fetch_add(1);
                build_range(l, b, mid);
            }
            node = r;
            b = mid;
        }
        for (auto& t : spawned) t.join();
    }
};

inline float box_area(const float* lo, const float* hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

}  // namespace

bool bvh8_build(const float* verts, const int32_t* obj, const int32_t* prim, uint64_t n, Bvh8* out,
                const char** err) {
    static const char* e_empty = "scene has no triangles";
    static const char* e_big = "more than 2^31 triangles";
    static const char* e_nan = "non-finite vertex coordinate";
    if (n == 0) {
        *err = e_empty;
        return false;
    }
    if (n >= (1ull << 31)) {
        *err = e_big;
        return false;
    }
    Builder2 B;
    B.verts = verts;
    B.n = n;
    B.tbox.resize(n);
    B.cent.resize(3 * n);
    B.idx.resize(n);
    for (uint64_t t = 0; t < n; t++) {
        const float* v = verts + 9 * t;
        for (int k = 0; k < 9; k++)
            if (!std::isfinite(v[k])) {
                *err = e_nan;
                return false;
            }
        Box b;
        b.reset();
        b.grow(v);
        b.grow(v + 3);
        b.grow(v + 6);
        B.tbox[t] = b;
        for (int a = 0; a < 3; a++) B.cent[3 * t + a] = 0.5f * (b.lo[a] + b.hi[a]);
        B.idx[t] = (uint32_t)t;
    }
    if (const char* e = getenv("WAPT_BVH_MAX_LEAF")) B.max_leaf = std::min(3, std::max(1, atoi(e)));
    if (const char* e = getenv("WAPT_BVH_CTRAV")) B.C_TRAV = (float)atof(e);
    B.nodes.resize(2 * n + 1);
    B.next = 1;
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    B.threads_free = (int)std::min(hw, 32u) - 1;
    B.build_range(0, 0, (uint32_t)n);

    const std::vector<Node2>& N2 = B.nodes;
    out->nodes.clear();
    out->tris.clear();
    out->tris.reserve(n);
    out->nodes.reserve(n / 4 + 16);
    for (int a = 0; a < 3; a++) {
        out->lo[a] = N2[0].box.lo[a];
        out->hi[a] = N2[0].box.hi[a];
    }

    struct Work {
        uint32_t n2;     // binary node to collapse
        uint32_t out;    // index of the wide node to fill
        uint32_t depth;
    };
    std::deque<Work> q;
    out->nodes.emplace_back();
    q.push_back({0u, 0u, 1u});
    uint32_t max_depth = 0;
    double sah = 0.0;
    const float root_area = std::max(N2[0].box.area(), 1e-30f);

    while (!q.empty()) {
        Work w = q.front();
        q.pop_front();
        max_depth = std::max(max_depth, w.depth);
        // gather up to 8 children by opening the largest inner child
        uint32_t ch[8];
        int nch = 0;
        if (N2[w.n2].count > 0) {
            ch[nch++] = w.n2;  // root is a single leaf
        } else {
            ch[nch++] = N2[w.n2].left;
            ch[nch++] = N2[w.n2].right;
            while (nch < 8) {
                int bi = -1;
                float ba = -1.f;
                for (int i = 0; i < nch; i++) {
                    if (N2[ch[i]].count == 0) {
                        float a = N2[ch[i]].box.area();
                        if (a > ba) {
                            ba = a;
                            bi = i;
                        }
                    }
                }
                if (bi < 0) break;
                uint32_t c = ch[bi];
                ch[bi] = N2[c].left;
                ch[nch++] = N2[c].right;
            }
        }
        // node box = union of children
        Box nb;
        nb.reset();
        for (int i = 0; i < nch; i++) nb.grow(N2[ch[i]].box);
        sah += (double)nb.area() / root_area;
        float nc[3];
        for (int a = 0; a < 3; a++) nc[a] = 0.5f * (nb.lo[a] + nb.hi[a]);

        // slot assignment (greedy on the 8x8 cost table)
        int slot_of[8];
        int child_in_slot[8];
        for (int s = 0; s < 8; s++) child_in_slot[s] = -1;
        for (int i = 0; i < 8; i++) slot_of[i] = -1;
        float cost[8][8];
        for (int i = 0; i < nch; i++) {
            const Box& cb = N2[ch[i]].box;
            float d[3];
            for (int a = 0; a < 3; a++) d[a] = 0.5f * (cb.lo[a] + cb.hi[a]) - nc[a];
            for (int s = 0; s < 8; s++) {
                float sx = (s & 4) ? 1.f : -1.f, sy = (s & 2) ? 1.f : -1.f, sz = (s & 1) ? 1.f : -1.f;
                cost[i][s] = d[0] * sx + d[1] * sy + d[2] * sz;
            }
        }
        for (int k = 0; k < nch; k++) {
            float best = -FLT_MAX;
            int bi = -1, bs = -1;
            for (int i = 0; i < nch; i++) {
                if (slot_of[i] >= 0) continue;
                for (int s = 0; s < 8; s++) {
                    if (child_in_slot[s] >= 0) continue;
                    if (cost[i][s] > best) {
                        best = cost[i][s];
                        bi = i;
                        bs = s;
                    }
                }
            }
            slot_of[bi] = bs;
            child_in_slot[bs] = bi;
        }

        Bvh8Node node;
        std::memset(&node, 0, sizeof(node));
        // grid: origin = node lo, cell = 2^e with extent/2^e <= 255
        int ebias[3];
        float cell[3];
        for (int a = 0; a < 3; a++) {
            node.p[a] = nb.lo[a];
            float ext = nb.hi[a] - nb.lo[a];
            int e = -126 + 1;
            if (ext > 0.f) {
                e = (int)std::ceil(std::log2((double)ext / 255.0));
                // guard against log2 rounding: make sure 255 cells cover the extent
                while (std::ldexp(255.0, e) < (double)ext) e++;
            }
            e = std::max(e, -125);
            e = std::min(e, 126);
            ebias[a] = e + 127;
            node.e[a] = (uint8_t)ebias[a];
            cell[a] = std::ldexp(1.0f, e);
        }
        node.child_base = (uint32_t)out->nodes.size();
        node.tri_base = (uint32_t)out->tris.size();
        uint32_t n_inner = 0, tri_off = 0;
        uint8_t* qlo[3] = {node.qlo_x, node.qlo_y, node.qlo_z};
        uint8_t* qhi[3] = {node.qhi_x, node.qhi_y, node.qhi_z};
        for (int s = 0; s < 8; s++) {
            int i = child_in_slot[s];
            if (i < 0) {
                node.meta[s] = 0;
                for (int a = 0; a < 3; a++) {
                    qlo[a][s] = 255;
                    qhi[a][s] = 0;
                }
                continue;
            }
            const Node2& c = N2[ch[i]];
            for (int a = 0; a < 3; a++) {
                double lo = ((double)c.box.lo[a] - (double)node.p[a]) / (double)cell[a];
                double hi = ((double)c.box.hi[a] - (double)node.p[a]) / (double)cell[a];
                int ql = (int)std::floor(lo), qh = (int)std::ceil(hi);
                ql = std::min(std::max(ql, 0), 255);
                qh = std::min(std::max(qh, 0), 255);
                // keep the box conservative under float evaluation of p + q*cell
                while (ql > 0 && node.p[a] + (float)ql * cell[a] > c.box.lo[a]) ql--;
                while (qh < 255 && node.p[a] + (float)qh * cell[a] < c.box.hi[a]) qh++;
                qlo[a][s] = (uint8_t)ql;
                qhi[a][s] = (uint8_t)qh;
            }
            if (c.count == 0) {
                node.imask |= (uint8_t)(1u << s);
                node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
                n_inner++;
            } else {
                static const uint8_t unary[4] = {0, 1, 3, 7};
                node.meta[s] = (uint8_t)((unary[c.count] << 5) | tri_off);
                for (uint32_t k = 0; k < c.count; k++) {
                    uint32_t t = B.idx[c.first + k];
                    const float* v = verts + 9 * (size_t)t;
                    Bvh8Tri T;
                    for (int a = 0; a < 3; a++) {
                        T.v0[a] = v[a];
                        T.v1[a] = v[3 + a];
                        T.v2[a] = v[6 + a];
                    }
                    T.obj = obj ? obj[t] : 0;
                    T.prim = prim ? prim[t] : (int32_t)t;
                    T.pad = 0;
                    out->tris.push_back(T);
                }
                tri_off += c.count;
                sah += (double)c.box.area() / root_area * c.count;
            }
        }
        // inner children get consecutive indices in slot order
        size_t base = out->nodes.size();
        out->nodes.resize(base + n_inner);
        uint32_t k = 0;
        for (int s = 0; s < 8; s++) {
            int i = child_in_slot[s];
            if (i < 0 || N2[ch[i]].count != 0) continue;
            q.push_back({ch[i], (uint32_t)(base + k), w.depth + 1});
            k++;
        }
        out->nodes[w.out] = node;
    }
    out->max_depth = max_depth;
    out->sah_cost = (float)sah;
    (void)box_area;
    return true;
}
