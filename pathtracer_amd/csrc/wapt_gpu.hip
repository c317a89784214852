// HIP kernels of the radiance loop and the GPU half of the C ABI (include/wapt.h).
// Queue layout and task encoding: kernels.h.  Traversal: traverse.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <memory>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh8.h"
#include "bvh_device.h"
#include "kernels.h"
#include "wapt_internal.h"

// Store of a queue entry by the march kernel: non-temporal (the entries are read once, by a later kernel, long after
// they have left the caches: march kernel 82.0 -> 78.6 ms per fog frame; WAPT_VOL_NT=0: plain stores).
#ifndef WAPT_VOL_NT
#define WAPT_VOL_NT 1
#endif
#if WAPT_VOL_NT
typedef float wapt_vf4 __attribute__((ext_vector_type(4)));
#define VSTORE(ptr, val)                                                                   \
    do {                                                                                   \
        const float4 _v = (val);                                                           \
        __builtin_nontemporal_store(wapt_vf4{_v.x, _v.y, _v.z, _v.w}, (wapt_vf4*)(ptr));   \
    } while (0)
#else
#define VSTORE(ptr, val) (*(ptr) = (val))
#endif

// =============================================================== device helpers

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Block size of the shading kernels.  Measured (surface kernel, ms per 64 spp of dragonChrome 1080p): 1024 threads
// 14.9 (one block per CU: everything waits at its barriers), 768 threads at 6 waves per SIMD 12.9 (two blocks per
// CU), 512 13.5, 256 19.7 (the smaller the block, the more atomics the queue counters take).
#ifndef WAPT_SBLOCK
#define WAPT_SBLOCK 768
#endif
#ifndef WAPT_SURFACE_WAVES
#define WAPT_SURFACE_WAVES 6
#endif
// 1: the surface kernel reserves the slots of a task's child and of its first shadow walk in one step (see k_surface)
#ifndef WAPT_SURFACE_MERGE
#define WAPT_SURFACE_MERGE 1
#endif
// block size of the march kernel (113 registers, 4 waves per SIMD: one block of 1024 threads per CU).  The kernel's
// time follows the number of block-wide queue reservations, not the occupancy: fog scene, ms per 2-spp frame,
// 264 / 152 / 135 with blocks of 256 / 512 / 1024 threads.
#ifndef WAPT_VBLOCK
#define WAPT_VBLOCK 1024
#endif
// waves per SIMD of the march kernel: in the open 63 registers, inside objects 70 (64 would spill six)
#ifndef WAPT_VOLUME_WAVES
#define WAPT_VOLUME_WAVES 8
#endif
#ifndef WAPT_VOLUME_WAVES_IN
#define WAPT_VOLUME_WAVES_IN 7
#endif
// block size of the march kernel proper (no block-wide steps in it)
#ifndef WAPT_VOLBLOCK
#define WAPT_VOLBLOCK 256
#endif
#ifndef WAPT_FAST_ATTENUATION
#define WAPT_FAST_ATTENUATION 1
#endif
#ifndef WAPT_PROBE_WAVES
#define WAPT_PROBE_WAVES 7
#endif

// Queue compaction.  Lanes that have something to emit get consecutive slots: a ballot
// and a population count give each lane its rank inside the wave, the waves' totals are
// scanned by the block's first wave, and the block takes ONE atomic on the queue counter.
// (One returning atomic per wave on a single counter tops out near 90 per microsecond on this
// chip, far below what the streaming kernels need; one per block is 12x fewer.)
// Every thread of the block must call this together.
// Inclusive prefix sums within every row of 16 lanes, by moves between the lanes' registers (a lane without a source
// lane adds zero).  The block's first wave runs this between two barriers that every other wave waits at.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t row_scan_u(uint32_t x) {
    x += dpp_u<0x111>(x);  // row_shr 1
    x += dpp_u<0x112>(x);  // row_shr 2
    x += dpp_u<0x114>(x);  // row_shr 4
    x += dpp_u<0x118>(x);  // row_shr 8
    return x;
}

__device__ __forceinline__ uint32_t block_push(uint32_t* counter, bool pred) {
    __shared__ uint32_t s_cnt[16];
    __shared__ uint32_t s_off[16];
    unsigned long long m = __ballot(pred);
    uint32_t lane = lane_id();
    uint32_t wave = threadIdx.x >> 6;
    uint32_t nw = blockDim.x >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (wave == 0) {
        uint32_t c = lane < nw ? s_cnt[lane] : 0u;
        const uint32_t x = row_scan_u(c);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 15);  // (lanes from nw on hold zero)
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(counter, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < nw) s_off[lane] = base + x - c;
    }
    __syncthreads();
    return s_off[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// The same for up to three queues at once, with one pair of barriers: sixteen lanes of the first wave
// scan each queue's wave totals side by side, and the three atomics leave in one instruction.
// slot[k] is meaningful for lanes with pred[k].
__device__ __forceinline__ void block_push3(uint32_t* const counter[3], const bool pred[3], uint32_t slot[3]) {
    __shared__ uint32_t s_cnt3[3][16];
    __shared__ uint32_t s_off3[3][16];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t nw = blockDim.x >> 6;
    unsigned long long m[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        m[k] = __ballot(pred[k]);
        if (lane == 0) s_cnt3[k][wave] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t qi = lane >> 4, w = lane & 15u;
        uint32_t c = (qi < 3u && w < nw) ? s_cnt3[qi][w] : 0u;
        const uint32_t x = row_scan_u(c);
        // the row's total sits in its last lane, which takes the atomic; the answer goes back to the row
        uint32_t base = 0;
        uint32_t* cp = qi == 0u ? counter[0] : (qi == 1u ? counter[1] : counter[2]);
        if (w == 15u && qi < 3u && x) base = atomicAdd(cp, x);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)base, 15), b1 = (uint32_t)__builtin_amdgcn_readlane((int)base, 31),
                       b2 = (uint32_t)__builtin_amdgcn_readlane((int)base, 47);
        base = qi == 0u ? b0 : (qi == 1u ? b1 : b2);
        if (qi < 3u && w < nw) s_off3[qi][w] = base + x - c;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; k++) slot[k] = s_off3[k][wave] + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum of v over the 64 lanes of the wave, the same value in every lane.  Data-parallel-primitive moves between the
// lanes' registers (pairs, quads, rotations within a row of 16, then the rows' last lanes handed on) instead of six
// round trips through the LDS crossbar: the pixel sums of the shadow kernels are on the path of every loop trip
// (shadow kernel of dragonChrome 33.5 -> 32.2 ms per frame with them).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_total_f(float v) {
    v += dpp_f<0xb1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4e>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x124>(v);  // row_ror 4
    v += dpp_f<0x128>(v);  // row_ror 8: every lane holds the sum of its row of 16
    v += dpp_f<0x142>(v);  // row_bcast 15: plus the sum of the row before (rows 1 and 3 are what matters)
    v += dpp_f<0x143>(v);  // row_bcast 31: rows 2 and 3 plus what lane 31 holds; lane 63 now has the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

#ifndef WAPT_ADD_PASSES
#define WAPT_ADD_PASSES 2
#endif

__device__ __forceinline__ void add_rgb(float* sum, uint32_t pixel, V3 c) {
    float* p = sum + (size_t)pixel * 3;
    if (c.x != 0.f) unsafeAtomicAdd(p + 0, c.x);
    if (c.y != 0.f) unsafeAtomicAdd(p + 1, c.y);
    if (c.z != 0.f) unsafeAtomicAdd(p + 2, c.z);
}

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

__device__ __forceinline__ uint32_t make_info(int depth, bool emit, int raytype, int medium) {
    return (uint32_t)depth | ((uint32_t)emit << 8) | ((uint32_t)raytype << 9) | ((uint32_t)(medium + 1) << 12);
}
__device__ __forceinline__ int info_depth(uint32_t i) { return (int)(i & 0xffu); }
__device__ __forceinline__ bool info_emit(uint32_t i) { return (i >> 8) & 1u; }
__device__ __forceinline__ int info_raytype(uint32_t i) { return (int)((i >> 9) & 3u); }
__device__ __forceinline__ int info_medium(uint32_t i) { return (int)(i >> 12) - 1; }

__device__ __forceinline__ const DMaterial& medium_of(const KScene& sc, int medium) {
    return sc.mats[medium < 0 ? sc.n_objects : medium];
}

// volumeAttenuation (cpp:407-427)
__device__ __forceinline__ V3 attenuation(const DMaterial& m, float dist) {
    if (m.flags & WAPT_MAT_CLEAR) return v3(1.f, 1.f, 1.f);
    return vexp(v3(-m.st[0] * dist, -m.st[1] * dist, -m.st[2] * dist));
}

// exp(a) where the result only multiplies a contribution (never decides anything): the hardware's 2^x on a * log2(e),
// good to about 1e-6 relative for the arguments of an attenuation, in 2 instructions instead of the library's 12
__device__ __forceinline__ V3 vexp_mul(V3 a) {
#if WAPT_FAST_ATTENUATION
    return v3(__expf(a.x), __expf(a.y), __expf(a.z));
#else
    return vexp(a);
#endif
}
// 1 / x to an ulp, for factors of contributions
__device__ __forceinline__ float fast_rcp(float x) {
#if WAPT_FAST_ATTENUATION
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// volumeAttenuation where the result is a factor of a contribution only
__device__ __forceinline__ V3 attenuation_mul(const DMaterial& m, float dist) {
    if (m.flags & WAPT_MAT_CLEAR) return v3(1.f, 1.f, 1.f);
    return vexp_mul(v3(-m.st[0] * dist, -m.st[1] * dist, -m.st[2] * dist));
}

__device__ __forceinline__ bool finite_dir(V3 d) {
    float l = dot(d, d);
    return l > 0.25f && l < 4.0f;  // false for NaN
}

// Stores a radiance task of depth d+1 in its slot (a slot beyond the queue is counted and dropped).
__device__ __forceinline__ void write_radiance(const KQueues& q, int depth_next, uint32_t slot, V3 o, V3 dir, V3 beta,
                                               uint32_t pixel, uint32_t info, uint32_t key) {
    if (slot >= q.capR) {
        atomicAdd(&q.cnt[CNT_QOVERFLOW], 1u);
        return;
    }
    size_t k = (size_t)depth_next * q.capR + slot;
    q.RA[k] = make_float4(o.x, o.y, o.z, u2f(pixel));
    q.RB[k] = make_float4(dir.x, dir.y, dir.z, u2f(info));
    q.RC[k] = make_float4(beta.x, beta.y, beta.z, u2f(key));
}

// Emits a radiance task of depth d+1 (all threads of the block call this together).
__device__ __forceinline__ void push_radiance(const KQueues& q, int depth_next, bool pred, V3 o, V3 dir, V3 beta,
                                              uint32_t pixel, uint32_t info, uint32_t key) {
    pred = pred && finite_dir(dir);
    uint32_t slot = block_push(&q.cnt[CNT_R(depth_next)], pred);
    if (pred) write_radiance(q, depth_next, slot, o, dir, beta, pixel, info, key);
}

// `target`: 0, or 1 + the index of the area light whose quad `to` was sampled on when the walk arrives
// at that quad steeply enough for its last hit to be decided without a second look (see k_shadow).
// Bit 31: nothing but the quad itself can be met in the last stretch before `to` (DLight::front_clear).
#define WAPT_TARGET_CLEAR 0x80000000u
__device__ __forceinline__ uint32_t shadow_target_steep(const DLight& L, int l) {
    return ((uint32_t)l + 1u) | (L.front_clear ? WAPT_TARGET_CLEAR : 0u);
}
// `wi`: unit direction from the walk's start to its end point
__device__ __forceinline__ uint32_t shadow_target(const DLight& L, int l, V3 wi) {
    if (L.type != 0) return 0u;
    return fabsf(dot(wi, ld3(L.N))) >= 0.25f ? shadow_target_steep(L, l) : 0u;
}

// A shadow walk from `from` along the unit vector `dir` over the distance `len` (to the sampled light point).
__device__ __forceinline__ void write_shadow(const KQueues& q, uint32_t slot, V3 from, V3 dir, float len, V3 w, uint32_t pixel,
                                             uint32_t target) {
    if (slot >= q.capSA) {
        atomicAdd(&q.cnt[CNT_QOVERFLOW], 1u);
        return;
    }
    q.SA[slot] = make_float4(from.x, from.y, from.z, u2f(pixel));
    q.SB[slot] = make_float4(dir.x, dir.y, dir.z, u2f(target));
    q.SC[slot] = make_float4(w.x, w.y, w.z, len);
}

// Body of a march sample's shadow walk in its slot (origin and pixel: KQueues::VO); length 0 = no walk.
__device__ __forceinline__ void write_shadow_body(const KQueues& q, uint32_t slot, bool pred, V3 dir, float len, V3 w, uint32_t target) {
    if (slot >= q.capS) {
        atomicAdd(&q.cnt[CNT_QOVERFLOW], 1u);
        return;
    }
    if (pred) VSTORE(&q.SB[slot], make_float4(dir.x, dir.y, dir.z, u2f(target)));
    VSTORE(&q.SC[slot], make_float4(w.x, w.y, w.z, pred ? len : 0.f));
}

__device__ __forceinline__ void push_shadow(const KQueues& q, bool pred, V3 from, V3 dir, float len, V3 w, uint32_t pixel,
                                            uint32_t target) {
    pred = pred && !is_zero(w);
    uint32_t slot = block_push(&q.cnt[CNT_S], pred);
    if (pred) write_shadow(q, slot, from, dir, len, w, pixel, target);
}

// Body of a probe task in its slot (the origin and the pixel are those of its march sample, KQueues::VO); an empty
// entry (zero direction) for a probe that is not sent.
__device__ __forceinline__ void write_probe_body(const KQueues& q, uint32_t slot, bool pred, V3 dir, V3 w, uint32_t info, int medium) {
    if (slot >= q.capP) {
        atomicAdd(&q.cnt[CNT_QOVERFLOW], 1u);
        return;
    }
    if (!pred) dir = v3(0.f, 0.f, 0.f);
    VSTORE(&q.PB[slot], make_float4(dir.x, dir.y, dir.z, u2f(info)));
    if (pred) VSTORE(&q.PC[slot], make_float4(w.x, w.y, w.z, u2f((uint32_t)(medium + 1))));
}

// One task of the current step together with its closest hit.
struct HitCtx {
    V3 o, dir, beta;
    uint32_t pixel, info, key, stream;
    float t;
    uint32_t tri;
    // valid when tri != miss
    V3 P, N;       // hit point, unit geometric normal
    int obj;
    bool entering; // dot(N, D) > 0 with D = -dir (cpp:236)
};

__device__ __forceinline__ void load_task(const KQueues& q, int depth, uint32_t start, uint32_t i, HitCtx& c) {
    size_t k = (size_t)depth * q.capR + start + i;
    float4 a = q.RA[k], b = q.RB[k], cc = q.RC[k];
    c.o = v3(a.x, a.y, a.z);
    c.pixel = f2u(a.w);
    c.dir = v3(b.x, b.y, b.z);
    c.info = f2u(b.w);
    c.beta = v3(cc.x, cc.y, cc.z);
    c.key = f2u(cc.w);
    c.stream = wapt_stream(c.key, c.pixel);
}

__device__ __forceinline__ void set_hit(const KScene& sc, uint2 h, HitCtx& c) {
    c.t = u2f(h.x);
    c.tri = h.y;
    if (c.tri == 0xffffffffu) return;
    int prim;
    c.N = normalized(tri_normal(sc.tm.tris, c.tri, &c.obj, &prim));
    c.P = point_at(c.o, c.t, c.dir);  // content.cpp:98-100
    c.entering = dot(c.N, -c.dir) > 0.0f;
}

__device__ __forceinline__ void load_hit(const KScene& sc, const KQueues& q, uint32_t i, HitCtx& c) {
    uint2 h = q.hits[i];
    c.t = u2f(h.x);
    c.tri = h.y;
    if (c.tri == 0xffffffffu) return;
    int prim;
    c.N = normalized(tri_normal(sc.tm.tris, c.tri, &c.obj, &prim));
    c.P = point_at(c.o, c.t, c.dir);  // content.cpp:98-100
    c.entering = dot(c.N, -c.dir) > 0.0f;
}

// Roulette thresholds (cpp:252, 256)
__device__ __forceinline__ float p_surface(int depth) { return 1.0f / (0.125f * (float)depth + 1.0f); }
__device__ __forceinline__ float p_volume(int depth) { return 1.0f / (0.04f * (float)depth + 1.0f); }

// beta * attenuation(medium0, P0->hit) / pS: the factor on everything the surface event returns (cpp:283-286).
// `gflags`, `gst`: flags and extinction of the global medium (the medium of most tasks), held by the caller.
__device__ __forceinline__ V3 surface_weight(const DMaterial* __restrict__ mats, const HitCtx& c, uint32_t gflags, V3 gst) {
    const int med = info_medium(c.info);
    uint32_t f0 = gflags;
    if (med >= 0) f0 = mats[med].flags;
    V3 w = c.beta;
    if (!(f0 & WAPT_MAT_CLEAR)) {
        V3 st = gst;
        if (med >= 0) st = ld3(mats[med].st);
        const float dist = mag(c.P - c.o);
        w = w * vexp_mul(v3(-st.x * dist, -st.y * dist, -st.z * dist));
    }
#if WAPT_FAST_FACTORS
    return (0.125f * (float)info_depth(c.info) + 1.0f) * w;  // 1 / pS without its two divisions (a factor only)
#else
    return (1.0f / p_surface(info_depth(c.info))) * w;
#endif
}

#ifdef WAPT_DEBUG_PIXEL
#define DBG(pix, ...) do { if ((pix) == q.debug_pixel) printf(__VA_ARGS__); } while (0)
#else
#define DBG(pix, ...) do { } while (0)
#endif

#define GRID_LOOP(var, n)                                                                       \
    for (uint32_t var = blockIdx.x * blockDim.x + threadIdx.x, _n = (((n) + blockDim.x - 1u) / blockDim.x) * blockDim.x; var < _n; \
         var += gridDim.x * blockDim.x)

// =============================================================== kernels

__global__ void k_reset(uint32_t* cnt, int r_slot, uint32_t r_value) {
    uint32_t t = threadIdx.x;
    if (t == 0) cnt[CNT_QV] = cnt[CNT_QM] = cnt[CNT_QMI] = cnt[CNT_S] = cnt[CNT_P] = 0u;
    cnt[CNT_FEED + t * WAPT_FEED_STRIDE] = 0u;  // 96 threads, 96 feed counters (three traversal kernels per step)
    if (r_slot >= 0 && t == 0) cnt[r_slot] = r_value;
}

// Which triangles of the tree belong to each light object (DLight::ntri, tri0, tri1; see k_probe).
__global__ void k_light_tris(const float4* tris, uint32_t n_tris, int n_lights, DLight* lights) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    int obj = (int)__float_as_uint(tris[(size_t)i * 3 + 2].y);
    if (obj >= n_lights) return;
    int k = atomicAdd(&lights[obj].ntri, 1);
    if (k == 0) lights[obj].tri0 = (int)i;
    if (k == 1) lights[obj].tri1 = (int)i;
}

// DLight::front_clear for the area lights (set to 1 by the host beforehand): cleared if a triangle of another object
// comes within `reach` of the light's quad without staying on or behind the quad's plane (see wapt_scene_create).
// One thread per triangle of the tree.
__global__ void k_front_clear(const float4* tris, uint32_t n_tris, int n_lights, DLight* lights, double reach) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const float4 a = tris[(size_t)i * 3], b = tris[(size_t)i * 3 + 1], c = tris[(size_t)i * 3 + 2];
    const int obj = (int)__float_as_uint(c.y);
    const float v[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x};
    for (int l = 0; l < n_lights; l++) {
        const DLight& L = lights[l];
        if (L.type != 0 || l == obj || !L.front_clear) continue;
        bool overlap = true;
        for (int ax = 0; ax < 3 && overlap; ax++) {
            const double q[4] = {L.P[ax], (double)L.P[ax] + L.U[ax], (double)L.P[ax] + L.V[ax], (double)L.P[ax] + L.U[ax] + L.V[ax]};
            const double lo = fmin(fmin(q[0], q[1]), fmin(q[2], q[3])) - reach, hi = fmax(fmax(q[0], q[1]), fmax(q[2], q[3])) + reach;
            const double tlo = fmin(fmin((double)v[ax], (double)v[3 + ax]), (double)v[6 + ax]);
            const double thi = fmax(fmax((double)v[ax], (double)v[3 + ax]), (double)v[6 + ax]);
            overlap = tlo <= hi && thi >= lo;
        }
        if (!overlap) continue;
        for (int k = 0; k < 3; k++) {
            const double sd = ((double)v[3 * k] - L.P[0]) * L.N[0] + ((double)v[3 * k + 1] - L.P[1]) * L.N[1] +
                              ((double)v[3 * k + 2] - L.P[2]) * L.N[2];
            if (!(sd <= 2.5e-4)) atomicExch(&lights[l].front_clear, 0);
        }
    }
}

// Between two slices of a step's march samples: the feeds of the shadow and probe kernels start over, and
// (shadows_too) the shadow queue, once its kernel has run.
__global__ void k_reset_sp(uint32_t* cnt, int shadows_too) {
    uint32_t t = threadIdx.x;
    if (t == 0 && shadows_too) cnt[CNT_S] = 0u;
    cnt[CNT_FEED + (32 + t) * WAPT_FEED_STRIDE] = 0u;  // 64 threads: feed counters of k_shadow and k_probe
}

// One iteration of renderTile's sample loop + generateRayForPixel (cpp:122-132, fundamentals.cpp:306-327): the camera
// task number `path` of a call - its pixel, the key of its sample and its ray.  A pure function of (path, parameters):
// the kernels of depth 0 can call it instead of reading the task from the queue.  False for a slot outside the image
// or the rendered region.
__device__ __forceinline__ bool camera_task(const KScene& sc, const KParams& p, unsigned long long path, V3* o, V3* d,
                                            uint32_t* pixel_out, uint32_t* key_out) {
    uint32_t s_local, r;
    if (p.pixel_major) {
        r = (uint32_t)(path / p.sample_count);
        s_local = (uint32_t)(path % p.sample_count);
    } else {
        s_local = (uint32_t)(path / p.slots_per_sample);
        r = (uint32_t)(path % p.slots_per_sample);
    }
    uint32_t tile = r >> 6, within = r & 63u;
    int px = (int)((tile % p.tilesX) * 8u + (within & 7u));
    int py = (int)((tile / p.tilesX) * 8u + (within >> 3));
    const bool valid = px < sc.cam.resI && py < sc.cam.resJ && px >= p.rl && px < p.rr && py >= p.rb && py < p.rt;
    uint32_t pixel = (uint32_t)(py * sc.cam.resI + px);
    uint32_t key = wapt_key_root(p.seed, (uint32_t)p.sample_begin + s_local);
    uint32_t stream = wapt_stream(key, pixel);
    camera_ray(sc.cam, px, py, wapt_rand(stream, WAPT_DIM_CAM0), wapt_rand(stream, WAPT_DIM_CAM1), o, d);
    *pixel_out = pixel;
    *key_out = key;
    return valid;
}

#define WAPT_CAMERA_INFO make_info(0, true, RAY_CAMERA, -1)

// The same for slot x (< 2^31) of a chunk of camera tasks that starts on a whole sample: what the depth-0 kernels call
// instead of reading the task.  No division instructions (see KParams::div_*); every slot is a pixel sample here
// (whole image, dimensions multiples of 8).
__device__ __forceinline__ void camera_task_at(const KScene& sc, const KParams& p, uint32_t x, V3* o, V3* d, uint32_t* pixel_out,
                                               uint32_t* key_out) {
    // (div_slots_* divide by slots_per_sample or, pixel by pixel, by sample_count)
    const uint32_t qd = udiv_by(x, p.div_slots_m, p.div_slots_k);
    const uint32_t qs = p.pixel_major ? x - qd * p.sample_count : p.cam_s0 + qd;       // sample of the call
    const uint32_t r = p.pixel_major ? p.cam_s0 + qd : x - qd * p.slots_per_sample;  // image slot
    const uint32_t tile = r >> 6, within = r & 63u;
    const uint32_t ty = udiv_by(tile, p.div_tiles_m, p.div_tiles_k);
    const uint32_t tx = tile - ty * p.tilesX;
    const int px = (int)(tx * 8u + (within & 7u));
    const int py = (int)(ty * 8u + (within >> 3));
    const uint32_t pixel = (uint32_t)(py * sc.cam.resI + px);
    const uint32_t key = wapt_key_root(p.seed, (uint32_t)p.sample_begin + qs);
    const uint32_t stream = wapt_stream(key, pixel);
    camera_ray(sc.cam, px, py, wapt_rand(stream, WAPT_DIM_CAM0), wapt_rand(stream, WAPT_DIM_CAM1), o, d);
    *pixel_out = pixel;
    *key_out = key;
}

__global__ __launch_bounds__(WAPT_SBLOCK) void k_raygen(KScene sc, KQueues q, KParams p, unsigned long long p0, uint32_t n) {
    GRID_LOOP(i, n) {
        V3 o, d;
        uint32_t pixel, key;
        const bool valid = camera_task(sc, p, p0 + i, &o, &d, &pixel, &key) && i < n;
        if (p.dense_raygen) {
            // every slot of the batch is a pixel sample: the task goes to slot i, no compaction needed
            if (i < n) {
                q.RA[i] = make_float4(o.x, o.y, o.z, u2f(pixel));
                q.RB[i] = make_float4(d.x, d.y, d.z, u2f(WAPT_CAMERA_INFO));
                q.RC[i] = make_float4(1.f, 1.f, 1.f, u2f(key));
            }
        } else {
            push_radiance(q, 0, valid, o, d, v3(1.f, 1.f, 1.f), pixel, WAPT_CAMERA_INFO, key);
        }
    }
}

// traceFAST for every task of the step (content.cpp:91-123).  The waves draw groups of consecutive tasks
// on demand (WaveFeed) and stream them through their lanes: a lane whose ray has ended takes the next one.
// COUNT adds node-visit and triangle-test tallies (rays[1], rays[2]) for the bytes-per-ray figure of
// the roofline.
// CAM: the step's tasks are camera tasks and are not in the queue: their rays are made here (camera_task_at, slot
// start + i of the chunk) instead of read - no ray generation pass, no 48-byte task written and read back twice.
template <bool COUNT, bool CAM = false>
__global__ __launch_bounds__(WAPT_BLOCK, WAPT_TRACE_WAVES) void k_trace(KScene sc, KQueues q, int depth, uint32_t start, uint32_t n,
                                                                        KParams p) {
    __shared__ uint2 s_stack[WAPT_LDS_STACK * WAPT_BLOCK];
    __shared__ CoopLds s_coop[WAPT_BLOCK / 64];
    CoopLds* coop = &s_coop[threadIdx.x >> 6];
    uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;
    TraceCount tc = {};
    WaveFeed feed = wave_feed(n, q.cnt + CNT_FEED);
    Trav t = {};
    bool have = false;
    uint32_t cur = 0, empty = 0;
    const size_t base = (size_t)depth * q.capR + start;
    for (;;) {
        uint32_t idx = feed.take(!have);
        if (idx != 0xffffffffu) {
            if (CAM) {
                V3 o, d;
                uint32_t pixel, key;
                camera_task_at(sc, p, start + idx, &o, &d, &pixel, &key);
                trav_begin(t, o, d, WAPT_TRACE_BIAS, WAPT_TFAR);
            } else {
                float4 a = q.RA[base + idx], b = q.RB[base + idx];
                trav_begin(t, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), WAPT_TRACE_BIAS, WAPT_TFAR);
                // (a slot that holds no task - a march sample without a child - is taken and dropped)
                if (f2u(b.w) == WAPT_INFO_NULL) {
                    empty++;
                    idx = 0xffffffffu;
                }
            }
            if (idx != 0xffffffffu) {
                cur = idx;
                have = true;
            }
        }
        // (a round whose slots were all empty is no reason to leave while the feed has more)
        if (!__any(have) && (CAM || feed.exhausted())) break;
        if (trav_step<COUNT, false, WAPT_COOP>(sc.tm, t, have, &s_stack[threadIdx.x], glane, tc, coop)) {
            q.hits[cur] = make_uint2(f2u(t.tmax), t.hit);  // fp32 distance; k_surface refines it
            have = false;
        }
    }
    if (COUNT) {
        uint32_t a = wave_sum(tc.nodes), b = wave_sum(tc.tris);
        if (lane_id() == 0) {
            atomicAdd(&q.rays[1], (unsigned long long)a);
            atomicAdd(&q.rays[2], (unsigned long long)b);
            atomicAdd(&q.rays[4], (unsigned long long)tc.wsteps);
            atomicAdd(&q.rays[5], (unsigned long long)tc.nsteps);
            atomicAdd(&q.rays[6], (unsigned long long)tc.ttrips);
        }
    }
    if (tc.overflow) atomicAdd(&q.cnt[CNT_OVERFLOW], tc.overflow);
    if (!CAM) {
        empty = wave_sum(empty);
        if (lane_id() == 0 && empty) atomicAdd(&q.rays[13], (unsigned long long)empty);
    }
}

// Does volumeScattering() march the segment of this task (cpp:221-230, 290, 454)?  A miss is lit through the GLOBAL
// medium up to the world sphere, a hit through the medium the task travels in.
__device__ __forceinline__ bool segment_marched(const KScene& sc, const HitCtx& c) {
    if (c.tri == 0xffffffffu) {
        const DMaterial& g = sc.mats[sc.n_objects];
        V3 rel = c.o - ld3(sc.wc);
        return g.scatters && (mag(rel) < sc.wr);
    }
    return medium_of(sc, info_medium(c.info)).scatters != 0u;
}

// The segment of a task that volumeScattering() marches (cpp:221-230, 290, 429-457).
struct Segment {
    V3 A, IBn;
    float len, pV;
    bool doVol, inside;
    int medium;
    int rms;   // march samples (cpp:457), 0 for a degenerate segment
};

__device__ __forceinline__ Segment march_segment(const KScene& sc, const KParams& p, const HitCtx& c, int depth) {
    Segment g;
    g.A = c.o;
    V3 B;
    if (c.tri == 0xffffffffu) {
        // exit point on the world sphere: the '+' root (cpp:226-227)
        V3 rel = c.o - ld3(sc.wc);
        float m = mag(rel);
        float s1 = 0.f, s2 = 0.f;
        solve_quadratic(1.0f, 2.0f * dot(rel, c.dir), m * m - sc.wr * sc.wr, &s1, &s2);
        B = c.o + s1 * c.dir;
        g.medium = -1;
        g.inside = false;
        g.doVol = true;
        g.pV = 1.0f;
    } else {
        B = c.P;
        g.medium = info_medium(c.info);
        g.inside = !c.entering;  // objID != -1 (cpp:240-248)
        g.pV = p_volume(depth);
        g.doVol = wapt_rand(c.stream, WAPT_DIM_VOL) < g.pV;
    }
    V3 IB = B - g.A;
    g.len = mag(IB);
    g.IBn = (1.0f / g.len) * IB;
    g.rms = (int)(g.len * p.mr / ((float)depth + 1.0f)) + 1;  // cpp:457
    if (!(g.len > 0.f)) g.rms = 0;
    return g;
}

// radiance() from the closest hit on (cpp:218-290) for every task of the step, in one pass over the
// queue: miss handling, roulette, emission, then the surface event of the hit material -
// Lambert: directReflection (cpp:345-381) + indirectReflection (cpp:383-405); mirror: the specular
// branch of reflection() (cpp:329-336); dielectric: the reflect-or-transmit choice (cpp:266-280),
// transmission() (cpp:296-311) and the dielectric branch of reflection() (cpp:329-341).
// The kernel moves 48 B of task, 8 B of hit and 48 B of triangle in and 48 B per child or shadow task
// out and is bound by that traffic, not by arithmetic: the materials share one kernel (lanes of a wave
// may run different branches) so that nothing is read twice.
// ONE: one hemisphere and one light sample per event (the reference's defaults): the sample loops are gone.
// MEDIA: the scene has a participating medium (marched segments are listed for the march kernel: KQueues::G0-G3).
// CAM: camera tasks made on the spot, as in k_trace (depth 0 of a scene without media, whole image).
// `mats`, `lights`: the material and light tables (sc.mats, sc.lights) as parameters of their own that nothing else
// aliases.  Read through the scene structure, every field was a vector load of its own at its point of use, each
// followed by a wait for all vector-memory operations in flight - a dozen dependent round trips per task; said this
// way the compiler may gather a task's loads, and fetches the fields of light l (the same in all lanes) with scalar loads.
template <bool ONE, bool MEDIA, bool CAM = false>
__global__ __launch_bounds__(WAPT_SBLOCK, WAPT_SURFACE_WAVES) void k_surface(KScene sc, KQueues q, KParams p, int depth, uint32_t start, uint32_t n,
                                                                              const DMaterial* __restrict__ mats, const DLight* __restrict__ lights) {
    const int lsamples = ONE ? 1 : p.lsamples, hsamples = ONE ? 1 : p.hsamples;
    enum { NONE, LAMBERT, MIRROR, DIELECTRIC };
    // the global medium, in which most tasks travel
    const uint32_t gflags = mats[sc.n_objects].flags;
    const V3 gst = ld3(mats[sc.n_objects].st);
    GRID_LOOP(i, n) {
        const bool valid = i < n;
        HitCtx c = {};
        int kind = NONE;
        bool toV = false;
        V3 w = v3(0, 0, 0), N1f = v3(0, 0, 1), kd = v3(0, 0, 0);
        // the one child of a mirror or dielectric event
        bool cemit = false;
        V3 cdir = v3(0, 0, 1), cb = v3(0, 0, 0);
        uint32_t cinfo = 0, ckey = 0;
        if (valid) {
            if (CAM) {
                camera_task_at(sc, p, start + i, &c.o, &c.dir, &c.pixel, &c.key);
                c.beta = v3(1.f, 1.f, 1.f);
                c.info = WAPT_CAMERA_INFO;
                c.stream = wapt_stream(c.key, c.pixel);
            } else {
                load_task(q, depth, start, i, c);
            }
        }
        // (a slot that holds no task: a march sample that had no child to send, see k_volume)
        if (valid && (!MEDIA || c.info != WAPT_INFO_NULL)) {
            uint2 h = q.hits[i];
            if (h.y != 0xffffffffu) {
                // the closest-hit kernel leaves the fp32 distance of its search; the fp64 re-evaluation runs
                // here, where every lane has a hit to refine
                float tr = refine_distance(sc.tm.tris, h.y, c.o, c.dir, WAPT_TRACE_BIAS, u2f(h.x));
                h.x = f2u(tr);
            }
            set_hit(sc, h, c);
            DBG(c.pixel, "head d=%d key=%08x o=(%.6f %.6f %.6f) dir=(%.7f %.7f %.7f) t=%.6f tri=%u P=(%.6f %.6f %.6f) obj=%d beta=%g\n", depth, c.key, c.o.x, c.o.y, c.o.z, c.dir.x, c.dir.y, c.dir.z, c.t, c.tri, c.P.x, c.P.y, c.P.z, c.obj, c.beta.x);
            if (c.tri == 0xffffffffu) {
                if (MEDIA) toV = segment_marched(sc, c);
            } else {
                // the first two words of the hit object's material, in one go: colour | flags, ior | kt != 0 | scatters
                const float4* mp = (const float4*)(mats + c.obj);
                const float4 ma = mp[0], mb = mp[1];
                const uint32_t mflags = f2u(ma.w);
                const V3 kcol = v3(ma.x, ma.y, ma.z);
                float xs = wapt_rand(c.stream, WAPT_DIM_SURF);
                if (xs < p_surface(depth)) {
                    w = surface_weight(mats, c, gflags, gst);
                    if (info_emit(c.info) && c.obj < sc.n_lights)
                        add_rgb(q.sum, c.pixel, w * ld3(lights[c.obj].Li));  // primitive.cpp:21-29
                    if (mflags & WAPT_MAT_DIFFUSE) kind = LAMBERT;
                    else if (mflags & WAPT_MAT_SPECULAR) kind = MIRROR;
                    else if (mflags & (WAPT_MAT_DIELECTRIC | WAPT_MAT_DIELECTRICSSS)) kind = DIELECTRIC;
                }
                if (MEDIA) toV = segment_marched(sc, c);
                N1f = c.entering ? c.N : -c.N;
                if (kind == LAMBERT) {
                    kd = kcol;
                } else if (kind == MIRROR) {
                    if (info_raytype(c.info) != RAY_DIFFUSE && depth + 1 <= p.td) {
                        cdir = reflect_dir(-c.dir, N1f);
                        cb = w * kcol;
                        cinfo = make_info(depth + 1, info_emit(c.info), RAY_SPECULAR, info_medium(c.info));
                        ckey = wapt_key_child(c.key, WAPT_SLOT_SPEC);
                        cemit = true;
                    }
                } else if (kind == DIELECTRIC) {
                    V3 D = -c.dir;
                    float F;
                    V3 D2 = refract_dir(c.N, D, mb.x, &F);
                    float pR = (1.0f - WAPT_BLEND) * 0.5f + WAPT_BLEND * F;
                    bool transmit = wapt_rand(c.stream, WAPT_DIM_DIEL) < (1.0f - pR);
                    if (depth + 1 <= p.td) {
                        if (transmit) {
                            const bool kt_nonzero = f2u(mb.y) != 0u;
                            if (kt_nonzero && !is_zero(D2)) {
                                cdir = -D2;
                                cb = ((1.0f - F) / (1.0f - pR)) * w;
                                int medium1 = c.entering ? c.obj : -1;
                                cinfo = make_info(depth + 1, info_emit(c.info), RAY_CAMERA, medium1);
                                ckey = wapt_key_child(c.key, WAPT_SLOT_TRANS);
                                cemit = true;
                            }
                        } else if (info_raytype(c.info) != RAY_DIFFUSE) {
                            cdir = reflect_dir(D, N1f);
                            cb = (F / pR) * (w * kcol);
                            cinfo = make_info(depth + 1, info_emit(c.info), RAY_SPECULAR, info_medium(c.info));
                            ckey = wapt_key_child(c.key, WAPT_SLOT_SPEC);
                            cemit = true;
                        }
                    }
                }
            }
        }
        constexpr bool MERGE = ONE && WAPT_SURFACE_MERGE;
        // the segment goes on the list of the march kernel with everything that kernel needs to know about it
        // (MERGE, scenes with lights: the list's slot is reserved below, together with the first light's shadow walk and the child)
        const bool qv_merged = MEDIA && MERGE && sc.n_lights > 0;
        auto write_segment = [&](uint32_t sv) {
            {
                const Segment g = march_segment(sc, p, c, depth);
                const bool inside_ok = g.inside && g.medium >= 0;  // (the global medium has no surface of its own to look for)
                q.G0[sv] = make_float4(g.A.x, g.A.y, g.A.z, g.len);
                q.G1[sv] = make_float4(g.IBn.x, g.IBn.y, g.IBn.z, g.pV);
                q.G2[sv] = make_float4(c.beta.x, c.beta.y, c.beta.z, u2f(c.key));
                q.G3[sv] = make_float4(u2f(c.pixel), u2f((uint32_t)(g.medium + 1) | ((uint32_t)g.doVol << 29) | ((uint32_t)g.inside << 30) | ((uint32_t)inside_ok << 31)),
                                       u2f((uint32_t)g.rms), u2f(c.stream));
            }
        };
        if (MEDIA && !qv_merged) {
            const uint32_t sv = block_push(&q.cnt[CNT_QV], toV);
            if (toV) write_segment(sv);
        }
        const bool lambert = kind == LAMBERT;
        const bool deeper = lambert && (depth + 1 <= p.td);
        // ONE: the event's single child is made first and parked in LDS (this thread's own entry), so that its slot
        // can be reserved together with the slot of the first light's shadow walk: one block-wide reservation (a pair
        // of barriers around an atomic whose answer the whole block waits for) per task instead of two.
        __shared__ float4 s_child[WAPT_SURFACE_MERGE ? 2 * WAPT_SBLOCK : 1];
        bool child_emit = false;
        if (MERGE) {
            V3 dir = cdir, b = cb;
            uint32_t info = cinfo, key = ckey;
            child_emit = cemit;
            if (deeper) {
                dir = hemisphere_uniform(N1f, wapt_rand(c.stream, WAPT_DIM_HEMI), wapt_rand(c.stream, WAPT_DIM_HEMI + 1u));
                float ndw = dot(dir, N1f);
                if (!(ndw < 0.0f)) {
                    b = ((1.0f / WAPT_PI) * ndw * (2.0f * WAPT_PI)) * (w * kd);
                    info = make_info(depth + 1, false, RAY_DIFFUSE, info_medium(c.info));
                    key = wapt_key_child(c.key, WAPT_SLOT_HEMI);
                    child_emit = true;
                }
            }
            child_emit = child_emit && finite_dir(dir);
            s_child[threadIdx.x] = make_float4(dir.x, dir.y, dir.z, u2f(info));
            s_child[WAPT_SBLOCK + threadIdx.x] = make_float4(b.x, b.y, b.z, u2f(key));
        }
        bool child_pending = MERGE;
        // Lambert, next event estimation: one shadow walk per light sample
        float rls = 1.0f / (float)lsamples;
        for (int l = 0; l < sc.n_lights; l++) {
            for (int s = 0; s < lsamples; s++) {
                bool emit = false;
                V3 LP = v3(0, 0, 0), wt = v3(0, 0, 0), Wi = v3(0, 0, 1);
                float Wl = 0.f;
                uint32_t tgt = 0u;
                if (lambert) {
                    uint32_t dim = WAPT_DIM_NEE + 2u * (uint32_t)(l * lsamples + s);
                    float rpdf;
                    const DLight& L = lights[l];
                    if (light_sample(L, c.P, N1f, true, wapt_rand(c.stream, dim), wapt_rand(c.stream, dim + 1), &LP, &rpdf)) {
                        Wi = normalized_len(LP - c.P, &Wl);
                        float ndw = dot(Wi, N1f);
                        if (!(ndw < 0.0f)) {
                            float sc_ = rls * (1.0f / WAPT_PI) * ndw * rpdf;
                            wt = sc_ * (w * kd * ld3(L.Li));
                            emit = true;
                            if (L.type == 0 && fabsf(dot(Wi, ld3(L.N))) >= 0.25f) tgt = shadow_target_steep(L, l);
                        }
                    }
                    DBG(c.pixel, "nee d=%d key=%08x emit=%d LP=(%.5f %.5f %.5f) wt=%g\n", depth, c.key, (int)emit, LP.x, LP.y, LP.z, wt.x);
                }
                if (MERGE && l == 0) {
                    uint32_t* const counters[3] = {&q.cnt[CNT_S], &q.cnt[CNT_R(depth + 1)], MEDIA ? &q.cnt[CNT_QV] : &q.cnt[CNT_R(depth + 1)]};
                    const bool preds[3] = {emit && !is_zero(wt), child_emit, MEDIA && toV};
                    uint32_t slots[3];
                    block_push3(counters, preds, slots);
                    if (MEDIA && toV) write_segment(slots[2]);
                    if (preds[0]) write_shadow(q, slots[0], c.P, Wi, Wl, wt, c.pixel, tgt);
                    if (preds[1]) {
                        const float4 cd = s_child[threadIdx.x], cbk = s_child[WAPT_SBLOCK + threadIdx.x];
                        write_radiance(q, depth + 1, slots[1], c.P, v3(cd.x, cd.y, cd.z), v3(cbk.x, cbk.y, cbk.z), c.pixel, f2u(cd.w), f2u(cbk.w));
                    }
                    child_pending = false;
                    continue;
                }
                push_shadow(q, emit, c.P, Wi, Wl, wt, c.pixel, tgt);
            }
        }
        if (MERGE) {
            if (child_pending) {  // (a scene without lights)
                const float4 cd = s_child[threadIdx.x], cbk = s_child[WAPT_SBLOCK + threadIdx.x];
                push_radiance(q, depth + 1, child_emit, c.P, v3(cd.x, cd.y, cd.z), v3(cbk.x, cbk.y, cbk.z), c.pixel, f2u(cd.w), f2u(cbk.w));
            }
            continue;
        }
        // children: hsamples uniform hemisphere directions of a Lambert event (no emission at the next
        // hit), or the single continuation of a mirror / dielectric event
        float rhs = 1.0f / (float)hsamples;
        for (int s = 0; s < hsamples; s++) {
            bool emit = cemit && s == 0;
            V3 dir = cdir, b = cb;
            uint32_t info = cinfo, key = ckey;
            if (deeper) {
                dir = hemisphere_uniform(N1f, wapt_rand(c.stream, WAPT_DIM_HEMI + 2u * s),
                                         wapt_rand(c.stream, WAPT_DIM_HEMI + 2u * s + 1u));
                float ndw = dot(dir, N1f);
                if (!(ndw < 0.0f)) {
                    b = (rhs * (1.0f / WAPT_PI) * ndw * (2.0f * WAPT_PI)) * (w * kd);
                    info = make_info(depth + 1, false, RAY_DIFFUSE, info_medium(c.info));
                    key = wapt_key_child(c.key, WAPT_SLOT_HEMI | (uint32_t)s);
                    emit = true;
                }
            }
            push_radiance(q, depth + 1, emit, c.P, dir, b, c.pixel, info, key);
        }
    }
}

// Slots for `count` consecutive entries per thread (a wave-wide prefix sum of the counts, then as block_push).
__device__ __forceinline__ uint32_t block_push_count(uint32_t* counter, uint32_t count) {
    __shared__ uint32_t s_cntN[16];
    __shared__ uint32_t s_offN[16];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t nw = blockDim.x >> 6;
    uint32_t x = count;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if ((int)lane >= o) x += y;
    }
    if (lane == 63u) s_cntN[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t c = lane < nw ? s_cntN[lane] : 0u;
        uint32_t t = c;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t y = __shfl_up(t, o);
            if ((int)lane >= o) t += y;
        }
        uint32_t total = __shfl(t, (int)nw - 1);
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(counter, total);
        base = __shfl(base, 0);
        if (lane < nw) s_offN[lane] = base + t - c;
    }
    __syncthreads();
    return s_offN[wave] + x - count;
}

// The number of march samples the step's tasks will produce, before any of them is shaded: the launches of a
// scene with media are sized by bounds on the segment lengths, and a segment inside an OPEN mesh can exceed its
// bound; the host cuts the step down to what fits if the count says so.  Same arithmetic as k_surface (refined hit
// distance, march_segment), so the count is exact.  out[0] += march samples, out[1] += marched segments.
__global__ __launch_bounds__(WAPT_VBLOCK) void k_count(KScene sc, KQueues q, KParams p, int depth, uint32_t start, uint32_t n,
                                                       unsigned long long* out) {
    unsigned long long mine = 0, segs = 0;
    GRID_LOOP(i, n) {
        if (i < n) {
            HitCtx c = {};
            load_task(q, depth, start, i, c);
            if (c.info == WAPT_INFO_NULL) continue;  // (a slot that holds no task)
            uint2 h = q.hits[i];
            if (h.y != 0xffffffffu) h.x = f2u(refine_distance(sc.tm.tris, h.y, c.o, c.dir, WAPT_TRACE_BIAS, u2f(h.x)));
            set_hit(sc, h, c);
            if (segment_marched(sc, c)) {
                mine += (unsigned long long)march_segment(sc, p, c, depth).rms;
                segs++;
            }
        }
    }
    __shared__ unsigned long long s_sum[2];
    if (threadIdx.x == 0) s_sum[0] = s_sum[1] = 0ull;
    __syncthreads();
    if (mine) atomicAdd(&s_sum[0], mine);
    if (segs) atomicAdd(&s_sum[1], segs);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum[1]) {
        atomicAdd(&out[0], s_sum[0]);
        atomicAdd(&out[1], s_sum[1]);
    }
}

// One entry (segment, sample) per march sample of the step's marched segments: the march kernel then has one
// sample per lane, whatever the lengths of the segments (marching a whole segment per lane left most lanes
// idle: a block ran until its longest segment was done).  Two lists: samples in the open (light sampling, sphere-sampling
// probes) from the front of QM, samples inside an object whose inside scatters (16 refracted connections per light)
// from its back; each list is marched by a launch of its own.
// SPLIT: the scene has such an object.
template <bool SPLIT>
__global__ __launch_bounds__(WAPT_VBLOCK) void k_expand(KQueues q) {
    uint32_t n = q.cnt[CNT_QV];
    GRID_LOOP(j, n) {
        uint32_t rms = 0;
        bool in = false;
        if (j < n) {
            const float4 g3 = q.G3[j];
            rms = f2u(g3.z);
            in = SPLIT && (f2u(g3.y) >> 31) != 0u;
        }
        const uint32_t base = block_push_count(&q.cnt[CNT_QM], in ? 0u : rms);
        if (!in)
            for (uint32_t s = 0; s < rms; s++)
                if (base + s < q.capQM) q.QM[base + s] = make_uint2(j, s);  // (the host refuses the step if the counts exceed the list)
        if (SPLIT) {
            const uint32_t ibase = block_push_count(&q.cnt[CNT_QMI], in ? rms : 0u);
            if (in)
                for (uint32_t s = 0; s < rms; s++)
                    if (ibase + s < q.capQM) q.QM[q.capQM - 1u - (ibase + s)] = make_uint2(j, s);
        }
    }
}

// volumeScattering (cpp:429-621): one march sample of one marched segment per lane.  Everything a sample emits goes
// to slots that follow from its number: jj = its number in this launch, `m0 + jj` its number in its list (see
// kernels.h), so the kernel has no compaction, no atomics and no barriers.
// INSIDE: the samples of this launch lie inside an object whose inside scatters (the 16-probe branch, cpp:560-583);
// else in the open (cpp:487-553), or in the global medium behind a back face (no direct light at all).
// `kid0`: slot in R[depth + 1] of the child of sample 0 of this launch's list, or 0xffffffff at the last depth.
// `lights`: the light table (sc.lights) as a parameter of its own that nothing else aliases: the compiler then fetches a
// light's fields with scalar loads (l is the same in all lanes).  Read through sc.lights inside this loop, which also
// stores, they are vector loads - the stores might alias - each followed by a wait on the vector-memory counter, and
// that counter counts the stores too: every light of every march sample waited for all queue writes issued before it.
template <bool INSIDE>
__global__ __launch_bounds__(WAPT_VOLBLOCK, INSIDE ? WAPT_VOLUME_WAVES_IN : WAPT_VOLUME_WAVES) void k_volume(KScene sc, KQueues q, KParams p, int depth, uint32_t m0, uint32_t n,
                                                                              uint32_t kid0, const DLight* __restrict__ lights) {
    uint32_t skipped = 0;  // probes decided here (tallied like those k_probe decides)
    const uint32_t ms = q.vol_ms;
    for (uint32_t jj = blockIdx.x * blockDim.x + threadIdx.x; jj < n; jj += gridDim.x * blockDim.x) {
        const uint2 e = INSIDE ? q.QM[q.capQM - 1u - (m0 + jj)] : q.QM[m0 + jj];
        const int s = (int)e.y;
        const float4 g0 = q.G0[e.x], g1 = q.G1[e.x], g2 = q.G2[e.x], g3 = q.G3[e.x];
        const V3 A = v3(g0.x, g0.y, g0.z), IBn = v3(g1.x, g1.y, g1.z), beta = v3(g2.x, g2.y, g2.z);
        const float len = g0.w, pV = g1.w;
        const uint32_t key = f2u(g2.w), pixel = f2u(g3.x), gw = f2u(g3.y), stream = f2u(g3.w);
        const int rms = (int)f2u(g3.z);
        const int medium = (int)(gw & 0x1fffffffu) - 1;
        const bool doVol = (gw >> 29) & 1u, inside = (gw >> 30) & 1u;
        const DMaterial& M = medium_of(sc, medium);
        const V3 ss = ld3(M.ss), st = ld3(M.st);
        const float rrms = rms > 0 ? fast_rcp((float)rms) : 0.f;
        float x, rpdf_x;
        {
            const float u = wapt_rand(stream, WAPT_DIM_MARCH + (uint32_t)s);
            if (p.me == 0.0f) {  // wa_samplerBeamUniform (sampling.cpp:75-81)
                x = u * len;
                rpdf_x = len;
            } else {  // wa_samplerBeamExponentialAttenuation (sampling.cpp:91-100)
                float ee = 1.0f - expf(-p.me * len);
                x = (-1.0f / p.me) * logf(1.0f - ee * u);
                rpdf_x = 1.0f / ((p.me / ee) * expf(-p.me * x));
            }
        }
        const V3 S = A + x * IBn;
        // common factor of everything gathered at S (cpp:611-612) times the scattering albedo and phase
        const V3 cs = (rpdf_x * rrms * WAPT_PHASE) * (beta * vexp_mul(v3(-st.x * x, -st.y * x, -st.z * x)) * ss);
        // (where the samples have children the record lies in R[depth + 1] itself, see the host: never beyond that queue)
        if (kid0 == 0xffffffffu || kid0 + jj < q.capR) VSTORE(&q.VO[jj], make_float4(S.x, S.y, S.z, u2f(pixel)));
        // (iii) indirect: one uniform direction, no emission at the next hit (cpp:587-609)
        if (kid0 != 0xffffffffu) {
            const uint32_t dim = WAPT_DIM_VCHILD + 2u * (uint32_t)s;
            const V3 cdir = sphere_uniform(wapt_rand(stream, dim), wapt_rand(stream, dim + 1));
            const bool cemit = doVol && finite_dir(cdir);
            const uint32_t slot = kid0 + jj;
            if (slot >= q.capR) {
                atomicAdd(&q.cnt[CNT_QOVERFLOW], 1u);
            } else {
                const size_t k = (size_t)(depth + 1) * q.capR + slot;
                if (cemit) {  // (plane A of the child is the sample's origin record, written above: see the host's choice of VO)
                    const V3 cbeta = ((4.0f * WAPT_PI) * fast_rcp(pV)) * cs;
                    VSTORE(&q.RB[k], make_float4(cdir.x, cdir.y, cdir.z, u2f(make_info(depth + 1, false, RAY_CAMERA, medium))));
                    VSTORE(&q.RC[k], make_float4(cbeta.x, cbeta.y, cbeta.z, u2f(wapt_key_child(key, WAPT_SLOT_VOL | (uint32_t)s))));
                } else {
                    VSTORE(&q.RB[k], make_float4(0.f, 0.f, 0.f, u2f(WAPT_INFO_NULL)));
                }
            }
        }
        if (!INSIDE) {
            // (i) in the open: light sampling and sphere sampling, MIS-weighted (cpp:487-553)
            const float rls = 1.0f / (float)p.lsamples;
            for (int l = 0; l < sc.n_lights; l++) {
                const DLight& L = lights[l];
                for (int k = 0; k < p.lsamples; k++) {
                    bool emit = false;
                    V3 LP = v3(0, 0, 0), wt = v3(0, 0, 0);
                    if (!inside) {
                        const uint32_t dim = WAPT_DIM_VNEE + 2u * (uint32_t)((s * sc.n_lights + l) * p.lsamples + k);
                        float rpdf;
                        if (light_sample(L, S, v3(0, 0, 0), false, wapt_rand(stream, dim), wapt_rand(stream, dim + 1), &LP, &rpdf)) {
                            // pdf / (pdf + c) with pdf = 1 / rpdf (cpp:500-503; 1 for rpdf = 0, as there), a factor only
                            const float mis = fast_rcp(1.0f + WAPT_MIS_CONST * rpdf);
                            wt = (rpdf * mis * rls) * (cs * ld3(L.Li));
                            emit = !is_zero(wt);
                        }
                    }
                    float Wl = 0.f;
                    const V3 Wi = normalized_len(LP - S, &Wl);
                    const uint32_t tgt = emit ? shadow_target(L, l, Wi) : 0u;
                    write_shadow_body(q, q.vol_s1 + (uint32_t)(l * p.lsamples + k) * ms + jj, emit, Wi, Wl, wt, tgt);
                }
                // the sphere-sampling probe (a light without geometry is never the closest hit of a ray: its probes are
                // not sent, and their directions - unit vectors, always usable - not even drawn)
                const V3 pw = (4.0f * WAPT_PI) * cs;
                if (L.probe_slot >= 0) {
                    const uint32_t dim = WAPT_DIM_VPROBE + 2u * (uint32_t)(s * sc.n_lights + l);
                    const V3 pdir = sphere_uniform(wapt_rand(stream, dim), wapt_rand(stream, dim + 1));
                    const bool probe = !inside && finite_dir(pdir) && !is_zero(pw);
                    write_probe_body(q, (uint32_t)L.probe_slot * ms + jj, probe, pdir, pw, (uint32_t)l, medium);
                } else if (!inside && !is_zero(pw)) {
                    skipped++;
                }
            }
        } else {
            // (ii) inside an object: 16 refracted connections per light (cpp:560-583)
            for (int l = 0; l < sc.n_lights; l++) {
                const DLight& L = lights[l];
                const V3 toL = normalized(ld3(L.centre) - S);
                for (int k = 0; k < 16; k++) {
                    const uint32_t dim = WAPT_DIM_VSINGLE + 2u * (uint32_t)((s * sc.n_lights + l) * 16 + k);
                    float rpdf = 0.f;
                    const V3 dir = sphere_biased3(toL, 1.5f, wapt_rand(stream, dim), wapt_rand(stream, dim + 1), &rpdf);
                    const V3 pw = ((1.0f / 16.0f) * rpdf) * cs;
                    write_probe_body(q, (uint32_t)(l * 16 + k) * ms + jj, finite_dir(dir) && !is_zero(pw), dir, pw, (uint32_t)l | (1u << 16), medium);
                }
            }
        }
    }
    skipped = wave_sum(skipped);
    if (lane_id() == 0 && skipped) atomicAdd(&q.rays[7], (unsigned long long)skipped);
}

// Number within its launch of the march sample that entry `rel` of a light-by-light block of the shadow or probe queue
// belongs to (rel % vol_ms; rel < 2^31).
__device__ __forceinline__ uint32_t vol_sample_of(const KQueues& q, uint32_t rel) {
    return rel - udiv_by(rel, q.vol_div_m, q.vol_div_k) * q.vol_ms;
}
// lanes left over by the majority passes that add on their own instead of being staged (fog's shadow kernel 104.2 ->
// 102.9 ms per frame with 8, dragonChrome's 17.2 -> 16.9 ms per 128 spp; 2 and 4 gain half of that)
#ifndef WAPT_ADD_DIRECT_MAX
#define WAPT_ADD_DIRECT_MAX 8
#endif
// Adds the colours of the lanes with `pred` to their pixels' sums.  The red, green and blue of one pixel
// go out from three neighbouring lanes: float atomics are carried out at the memory side, one request per
// 64-byte segment that a wave instruction touches, so this takes one request per pixel where one atomic
// instruction per channel takes three.  `stage` is 64 float4 of LDS owned by the calling wave; all lanes
// of the wave call this together.
__device__ __forceinline__ void wave_add_rgb(float* sum, float4* stage, bool pred, uint32_t pixel, V3 c) {
#ifdef WAPT_EXP_NOADD
    asm volatile("" ::"v"(pixel), "v"(c.x), "v"(c.y), "v"(c.z));
    return;
#endif
    unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    uint32_t lane = lane_id();
    // Tasks ordered pixel by pixel: most lanes of a wave hold samples of ONE pixel, or of two.  The lanes on the pixel
    // of the first of them are summed across the wave (three sums in the lanes' own registers, wave_total_f; one
    // request), twice if need be; what is left goes the general way.
#pragma unroll
    for (int pass = 0; pass < WAPT_ADD_PASSES; pass++) {
        const uint32_t px0 = (uint32_t)__builtin_amdgcn_readlane((int)pixel, __ffsll((long long)m) - 1);
        const bool maj = pred && pixel == px0;
        if (__popcll(__ballot(maj)) <= 2) break;
        const float r = wave_total_f(maj ? c.x : 0.f), g = wave_total_f(maj ? c.y : 0.f), b = wave_total_f(maj ? c.z : 0.f);
        if (lane < 3u) {
            const float v = lane == 0u ? r : lane == 1u ? g : b;
#ifdef WAPT_EXP_NOATOMIC
            asm volatile("" ::"v"(v));
#else
            if (v != 0.f) unsafeAtomicAdd(sum + (size_t)px0 * 3 + lane, v);
#endif
        }
        pred = pred && !maj;
        m = __ballot(pred);
        if (m == 0ull) return;
    }
#if WAPT_ADD_DIRECT_MAX > 0
    // a few lanes left: their own atomics, three requests per lane, cost less than the staging below
    if (__popcll(m) <= WAPT_ADD_DIRECT_MAX) {
        if (pred) {
            float* sp = sum + (size_t)pixel * 3;
            if (c.x != 0.f) unsafeAtomicAdd(sp, c.x);
            if (c.y != 0.f) unsafeAtomicAdd(sp + 1, c.y);
            if (c.z != 0.f) unsafeAtomicAdd(sp + 2, c.z);
        }
        return;
    }
#endif
    if (pred) {
        uint32_t at = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        WAPT_CHECK_IDX(at, 64u, "staging slot of a pixel add");
        stage[at] = make_float4(c.x, c.y, c.z, u2f(pixel));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t n = (uint32_t)__popcll(m), n3 = 3u * n;
    const float* sf = (const float*)stage;
    for (uint32_t idx = lane; idx < n3; idx += 64u) {
        uint32_t e = (idx * 43691u) >> 17;  // idx / 3 for idx < 192
        uint32_t ch = idx - 3u * e;
        uint32_t px = f2u(sf[4u * e + 3u]);
        // neighbouring lanes that hold the same pixel (march samples of one segment sit next to each other
        // in the queues) are summed here, and the first of the run adds the sum
        if (e > 0u && f2u(sf[4u * e - 1u]) == px) continue;
        float v = sf[4u * e + ch];
        for (uint32_t k = e + 1u; k < n && f2u(sf[4u * k + 3u]) == px; k++) v += sf[4u * k + ch];
        if (v != 0.f) unsafeAtomicAdd(sum + (size_t)px * 3 + ch, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// visibility() (cpp:693-790): walk from A to B through every surface in between.  Lanes stream
// tasks like k_trace; a walk that has to continue past a transparent surface keeps its lane.
//
// Most walks end in one of three ways that need no second look at the scene, and those are decided
// from what the search left in registers: (1) the search met an opaque surface clearly before B and
// stopped there: zero visibility (cpp:778); (2) nothing was hit up to B: the walk arrives, attenuated by the medium
// it is in over what is left of the way (cpp:764-772; nothing to do in the clear global medium); (3) same medium, and the only hit is
// the quad of the light that B was sampled on, reached steeply: B lies on that quad, so the hit is B
// itself to within a fraction of TRACE_BIAS, the walk arrives (cpp:775) and a plain (opaque, clear,
// non-dielectric) surface changes nothing there.  Everything else - media, transparent surfaces, hits
// within the search margin of B - takes the general path below, which follows the reference step by step.
// PLAIN: every object of the scene is opaque, clear inside and no dielectric, and the global medium is clear.  A walk
// is then over with its first search - nothing is passed through, nothing attenuates: blocked if the search stopped
// at a surface before B or its closest hit lies before B - TRACE_BIAS, arrived otherwise (cpp:764-778 with V = 1).
template <bool COUNT, bool PLAIN, bool FOGGY = false>
__global__ __launch_bounds__(WAPT_BLOCK, PLAIN ? WAPT_SHADOW_WAVES : WAPT_SHADOW_WAVES_GENERAL) void k_shadow(KScene sc, KQueues q) {
    __shared__ uint2 s_stack[WAPT_LDS_STACK * WAPT_BLOCK];
    __shared__ union WaveLds {
        float4 stage[64];
        CoopLds coop;
    } s_wave[WAPT_BLOCK / 64];  // (the two are used at different points of a loop trip)
    float4* stage = s_wave[threadIdx.x >> 6].stage;
    CoopLds* coop = &s_wave[threadIdx.x >> 6].coop;
    constexpr bool COOP = PLAIN && WAPT_SHADOW_COOP;
    // VOL: beyond the walks that the surface kernel sent (entries below s1, with their own origin) the queue can hold the
    // walks of march samples, whose origin and pixel are their sample's (KQueues::VO)
    constexpr bool VOL = FOGGY || !PLAIN;
    uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;
    const bool with_vol = VOL && q.vol_ms != 0u;
    const uint32_t s1 = with_vol ? q.vol_s1 : min(q.cnt[CNT_S], q.capS);
    const uint32_t n = with_vol ? s1 + q.vol_scount : s1;
    TraceCount tc = {};
    uint32_t rays = 0;
    const int global_m = sc.n_objects;
    const bool gclear = (sc.mats[global_m].flags & WAPT_MAT_CLEAR) != 0u;
    // (the global medium's extinction, fetched once: read where it is used, inside the loop, it is two dependent vector
    //  loads - flags, then coefficients - per trip, each waited for on the spot)
    const V3 gst = FOGGY ? ld3(sc.mats[global_m].st) : v3(0.f, 0.f, 0.f);
    WaveFeed feed = wave_feed(n, q.cnt + CNT_FEED + 32 * WAPT_FEED_STRIDE);
    Trav t = {};
    bool have_task = false, have_ray = false;
    // walk state (names as in the reference).  The segment start P0 and the direction ABn live in t.o / t.d;
    // B stays in the queue and is read again by the few walks that pass a transparent surface.
    V3 V = v3(1, 1, 1);
    float ABl = 0.f, d = 0.f;
    int cur = global_m, it = 0;
    uint32_t target = 0;
    bool cut = false;  // the search of this walk ends short of B (see below)
    uint32_t ctask = 0;   // general form: the task, whose weight is read again when the walk arrives (its walks are long,
                          // and three registers less keep the kernel out of scratch memory at 7 waves per SIMD)
    V3 cw = v3(0, 0, 0);  // weight and pixel of the task (read with it: a read at the walk's end is a memory round trip
    uint32_t cpixel = 0;  // that the whole wave waits for, 4 % of the kernel)
    for (;;) {
        uint32_t idx = feed.take(!have_task);
        if (idx != 0xffffffffu) {
            WAPT_CHECK_IDX(idx, n, "shadow task");
            float4 a, b = q.SB[idx];
            if (VOL && idx >= s1) a = q.VO[vol_sample_of(q, idx - s1)];
            else a = q.SA[idx];
            float4 w4s = q.SC[idx];
            if (PLAIN) cw = v3(w4s.x, w4s.y, w4s.z);
            else ctask = idx;
            cpixel = f2u(a.w);
            ABl = w4s.w;  // (direction and length come with the task: the kernel that sampled the light point had them)
            t.o = v3(a.x, a.y, a.z);
            t.d = v3(b.x, b.y, b.z);
            V = v3(1.f, 1.f, 1.f);
            d = 0.f;
            cur = global_m;
            it = 0;
            target = f2u(b.w);
            cut = PLAIN && sc.fast_shadow && (target & WAPT_TARGET_CLEAR) != 0u;
            target &= ~WAPT_TARGET_CLEAR;
            have_task = ABl > 0.f;
        }
        if (have_task && !have_ray) {
            // Nothing beyond B matters (a hit there and a miss are treated alike, cpp:764-769) and an opaque
            // surface clearly before B ends the walk with zero visibility whatever else is in between
            // (cpp:778), so the search is cut at B and stops at the first such surface.  The margins cover
            // the fp32 error of search distances (about 2e-5 relative); the walk's own tests below use the
            // refined distance and the reference's thresholds.
            float rest = ABl - d, margin = 0.002f + 1e-4f * ABl;
            // A walk that ends on the quad of a light with nothing in front of it (PLAIN scenes): the quad is the only
            // surface within the margin of B, and it is met at B (cpp:775: arrived), so the search ends short of it -
            // whatever it finds blocks, and the nodes and triangles around the light are never visited.
            trav_begin(t, t.o, t.d, WAPT_TRACE_BIAS, cut ? rest - margin : rest + margin);
            // (never negative here: a negative value is how the search reports that it stopped at an opaque surface; with a
            //  remaining way shorter than the margin the stop simply never fires)
            t.tblock = sc.fast_shadow ? fmaxf(rest - margin, 0.0f) : -1.0f;  // (the step-by-step path searches for the closest hit)
            have_ray = true;
            rays++;
        }
        // (a round whose tasks were all finished on the spot is no reason to leave while the feed has more; the
        //  rest of the loop body does nothing then)
        if (!__any(have_task) && feed.exhausted()) break;
        bool lit = false;
        if (trav_step<COUNT, true, COOP>(sc.tm, t, have_ray, &s_stack[threadIdx.x], glane, tc, coop)) {
            have_ray = false;
            const uint32_t tri = t.hit;
            bool blocked = sc.fast_shadow && t.tblock < 0.f;  // (1)
            bool general = false;
            if (PLAIN && cut) {
                lit = !blocked && tri == 0xffffffffu;
                blocked = !lit;
                if (FOGGY && lit) V = gclear ? v3(1.f, 1.f, 1.f) : vexp_mul(v3(-gst.x * ABl, -gst.y * ABl, -gst.z * ABl));
            } else if (PLAIN) {
                if (!blocked) {
                    bool at_target = false;
                    if (COOP) {
                        // (the cooperative phase does not bring the hit triangle's flag word along: the quad of the
                        //  sampled light is known by its two triangles)
                        if (tri != 0xffffffffu && target != 0u) {
                            const DLight& TL = sc.lights[target - 1u];
                            at_target = tri == (uint32_t)TL.tri0 || tri == (uint32_t)TL.tri1;
                        }
                    } else {
                        at_target = tri != 0xffffffffu && target != 0u && (t.hflag >> WAPT_TRI_OBJ_SHIFT) == target - 1u;
                    }
                    // FOGGY (the global medium attenuates): the part of the way it attenuates - all of it unless the walk
                    // ends on a back face, whose last stretch the step-by-step path takes through the object's clear inside
                    float seg = ABl;
                    if (tri == 0xffffffffu || (!FOGGY && at_target)) {
                        lit = true;  // (2), (3)
                    } else {
                        // the distances as the step-by-step path takes them: fp64 for a hit before or near B
                        float tmax = t.tmax;
                        if (!(tmax > ABl + (0.002f + 5e-5f * ABl))) tmax = refine_distance(sc.tm.tris, tri, t.o, t.d, WAPT_TRACE_BIAS, tmax);
                        const float dd = mag(point_at(t.o, tmax, t.d) - t.o);
                        lit = dd >= ABl - WAPT_TRACE_BIAS;
                        // (a front face of glass ends the walk even within TRACE_BIAS of B, and a little beyond: the
                        //  reference looks at the material, cpp:757-759, before it asks whether B is reached, cpp:775)
                        if (!FOGGY && lit && dd <= ABl + WAPT_TRACE_BIAS &&
                            (__float_as_uint(sc.tm.tris[(size_t)tri * 3 + 2].w) & WAPT_TRI_GLASS) != 0u)
                            lit = false;
                        blocked = !lit;
                        if (FOGGY && lit && dd <= ABl + WAPT_TRACE_BIAS) {
                            int obj, prim;
                            const V3 N = tri_normal(sc.tm.tris, tri, &obj, &prim);
                            seg = dot(t.d, N) < 0.0f ? dd : 0.0f;
                        }
                    }
                    if (FOGGY && lit && seg > 0.0f) V = gclear ? v3(1.f, 1.f, 1.f) : vexp_mul(v3(-gst.x * seg, -gst.y * seg, -gst.z * seg));
                }
            } else if (!blocked) {
                const bool in_clear = gclear && cur == global_m;
                const bool at_target = tri != 0xffffffffu && target != 0u && (t.hflag & WAPT_TRI_PLAIN) &&
                                       (t.hflag >> WAPT_TRI_OBJ_SHIFT) == target - 1u;
                if (sc.fast_shadow && in_clear && (tri == 0xffffffffu || at_target)) {
                    lit = true;  // (2), (3)
                } else if (sc.fast_shadow && tri == 0xffffffffu) {
                    // (2) in a medium that attenuates: what is left of the way to B, through the medium the walk is in
                    // (the last step of the general path, cpp:764-775)
                    // (on a continued walk the start of the segment lies on the way, d from A: |B - P0| = |AB| - d)
                    const float seg = ABl - d;
                    WAPT_CHECK_IDX(cur, sc.n_objects + 1, "medium of a shadow walk");
                    V = V * attenuation(sc.mats[cur], seg);
                    lit = true;
                } else {
                    general = true;
                }
            }
            if (general) {
                const V3 P0 = t.o, ABn = t.d;
                float tmax = t.tmax;
                if (tri != 0xffffffffu) {
                    // fp32 search distances are good to about 2e-5 relative; the walk's decisions compare the
                    // distance with |AB| +- 0.001, so a hit clearly beyond B needs no better value (B itself is
                    // used); one before or near B gets the fp64 distance (it may seed the next segment)
                    float dd32 = d + tmax;
                    if (!(dd32 > ABl + (0.002f + 5e-5f * ABl))) tmax = refine_distance(sc.tm.tris, tri, P0, ABn, WAPT_TRACE_BIAS, tmax);
                }
                V3 P1;
                bool entering = false, opaque = false;
                int obj = -1;
                float seg;   // |P1 - P0|
                V3 hp = point_at(P0, tmax, ABn);
                const float d0 = d;  // distance from A to the start of this segment
                if (tri != 0xffffffffu) d += mag(hp - P0);
                else d = 2.0f * ABl;
                if (d <= ABl + WAPT_TRACE_BIAS) {
                    int prim;
                    V3 N = normalized(tri_normal(sc.tm.tris, tri, &obj, &prim));
                    WAPT_CHECK_IDX(obj, sc.n_objects, "object of a shadow walk's hit");
                    const DMaterial& m = sc.mats[obj];
                    P1 = hp;
                    seg = mag(P1 - P0);
                    entering = dot(ABn, N) < 0.0f;
                    opaque = (m.flags & WAPT_MAT_OPAQUE) != 0u;
                    if (entering) {
                        cur = global_m;
                        // glass blocks shadow walks outright (cpp:757-759)
                        blocked = (m.flags & (WAPT_MAT_DIELECTRIC | WAPT_MAT_DIELECTRICSSS)) && m.ior > 1.0f;
                    } else {
                        cur = obj;
                    }
                } else {
                    // the segment ends at B; on the first segment P0 is A and |B - A| is ABl itself
                    P1 = P0;
                    seg = ABl - d0;  // |B - P0|: P0 lies on the way, d0 from A (on the first segment P0 is A and d0 is 0)
                    d = ABl;
                }
                if (!blocked) {
                    V = V * attenuation(sc.mats[cur], seg);
                    if (d >= ABl - WAPT_TRACE_BIAS) lit = true;
                    else if (opaque) blocked = true;
                    else if (V.x < WAPT_LOW_VIS && V.y < WAPT_LOW_VIS && V.z < WAPT_LOW_VIS) blocked = true;
                }
                if (!blocked && !lit && ++it >= 4096) blocked = true;
                if (!lit && !blocked) {
                    t.o = P1;
                    cur = entering ? obj : global_m;
                }
            }
            if (lit || blocked) have_task = false;
        }
        // lanes whose walk has arrived add weight x visibility to their pixel
        V3 contrib = v3(0, 0, 0);
        uint32_t pixel = 0;
        if (lit) {
            pixel = cpixel;
            WAPT_CHECK_IDX(pixel, (uint32_t)(sc.cam.resI * sc.cam.resJ), "pixel of a shadow task");
            if (!PLAIN) {
                const float4 w4 = q.SC[ctask];
                cw = v3(w4.x, w4.y, w4.z);
            }
            contrib = cw * V;
            DBG(pixel, "shadow lit V=%g w=%g d=%.6f ABl=%.6f\n", V.x, cw.x, d, ABl);
        }
        wave_add_rgb(q.sum, stage, lit, pixel, contrib);
    }
    rays = wave_sum(rays);
    if (lane_id() == 0 && rays) atomicAdd(q.rays, (unsigned long long)rays);
    if (COUNT) {
        uint32_t a = wave_sum(tc.nodes), b = wave_sum(tc.tris);
        if (lane_id() == 0) {
            atomicAdd(&q.rays[8], (unsigned long long)a);
            atomicAdd(&q.rays[9], (unsigned long long)b);
            atomicAdd(&q.rays[10], (unsigned long long)tc.wsteps);
            atomicAdd(&q.rays[11], (unsigned long long)tc.nsteps);
            atomicAdd(&q.rays[12], (unsigned long long)tc.ttrips);
        }
    }
    if (tc.overflow) atomicAdd(&q.cnt[CNT_OVERFLOW], tc.overflow);
}

// directLightHit (cpp:670-687) and directLightingSingleScatter (cpp:631-665).  Lanes stream tasks like
// k_trace; a single-scatter probe that reaches its object's surface keeps its lane for the refracted
// second ray.
//
// Both estimators trace a ray and keep the result only if the closest hit is one given light.  The light's
// quad is two triangles of the tree (DLight::tri0/1), so the question is put to them first, with the test the
// search itself uses: a ray that misses them cannot have the light as its closest hit and is not searched at
// all (most rays: the quad is small); one that meets them at distance tL is searched up to tL only, and the
// search stops at the first other triangle it accepts.  A light with no geometry (sphere lights) is never
// hit, one with another number of triangles gets the plain closest-hit search.  Queries answered without a
// search are tallied in rays[7], not in the ray count.
template <bool COUNT>
__global__ __launch_bounds__(WAPT_BLOCK, WAPT_PROBE_WAVES) void k_probe(KScene sc, KQueues q) {
    __shared__ uint2 s_stack[WAPT_LDS_STACK * WAPT_BLOCK];
    __shared__ union WaveLds {
        float4 stage[64];
        CoopLds coop;
    } s_wave[WAPT_BLOCK / 64];
    float4* astage = s_wave[threadIdx.x >> 6].stage;
    CoopLds* coop = &s_wave[threadIdx.x >> 6].coop;
    uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = min(q.vol_pcount, q.capP);
    TraceCount tc = {};
    uint32_t rays = 0, decided = 0;
    WaveFeed feed = wave_feed(n, q.cnt + CNT_FEED + 64 * WAPT_FEED_STRIDE);
    Trav t = {};
    bool have_task = false, have_ray = false;
    uint32_t task = 0;
    int stage = 0;          // 1: the ray in flight is the refracted one
    float seg0 = 0.f, F0 = 0.f;  // length of the first segment and Fresnel term at the surface
    // the ray in flight asks for the quad of a light: its triangle there (0xffffffff: plain closest-hit search)
    uint32_t want = 0xffffffffu;
    int want_light = 0;

    // The lane has a ray to aim at light `want_light`, its origin and direction in t.o / t.d: a new directLightHit probe,
    // or the refracted second ray of a single-scatter probe.  The search is started - or the query decided on the spot -
    // at ONE place of the loop, below.  (With the start written out at both places, ROCm 7.2's compiler merged the
    // copies and lost the assignment of the new direction to t.d on one of the merged paths - the second ray was then
    // searched along the first ray's direction; tests/test_gpu_render.py::test_dragon_milk_same_seed shows it.)
    bool aim_pending = false;

    for (;;) {
        uint32_t idx = feed.take(!have_task);
        if (idx != 0xffffffffu) {
            WAPT_CHECK_IDX(idx, n, "probe task");
            const float4 b = q.PB[idx];
            const uint32_t info = f2u(b.w);
            task = idx;
            stage = 0;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool sent = b.x != 0.f || b.y != 0.f || b.z != 0.f;  // (an entry with a zero direction is a probe that was not sent)
            if (sent) a = q.VO[vol_sample_of(q, idx)];
            if (!sent) {
                have_task = have_ray = false;
            } else if ((info >> 16) == 0u) {
                t.o = v3(a.x, a.y, a.z);
                t.d = v3(b.x, b.y, b.z);
                want_light = (int)(info & 0xffffu);
                aim_pending = true;
                have_task = true;
                have_ray = false;
            } else {
                want = 0xffffffffu;
                trav_begin(t, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), WAPT_TRACE_BIAS, WAPT_TFAR);
                have_task = have_ray = true;
                rays++;
            }
        }
        if (aim_pending) {
            // Starts the search of a ray whose closest hit has to be light `want_light`, or decides it on the spot.
            aim_pending = false;
            const DLight& L = sc.lights[want_light];
            const V3 o = t.o, d = t.d;
            bool started = true;
            want = 0xffffffffu;
            float tfar = WAPT_TFAR;
            if (sc.fast_probe && L.ntri == 0) started = false;
            else if (sc.fast_probe && L.ntri == 2)   // nothing beyond the quad matters
                started = quad_hit(sc.tm.tris, (uint32_t)L.tri0, (uint32_t)L.tri1, o, d, WAPT_TRACE_BIAS, WAPT_TFAR, &tfar, &want);
            if (started) trav_begin(t, o, d, WAPT_TRACE_BIAS, tfar);
            have_task = have_ray = started;
            rays += started ? 1u : 0u;
            decided += started ? 0u : 1u;
        }
        // (a round whose tasks were all finished on the spot is no reason to leave while the feed has more; the
        //  rest of the loop body does nothing then)
        if (!__any(have_task) && feed.exhausted()) break;
        bool lit = false;
        V3 contrib = v3(0, 0, 0);
        bool done = trav_step<COUNT, false, WAPT_COOP>(sc.tm, t, have_ray, &s_stack[threadIdx.x], glane, tc, coop);
        if (have_ray && !done && want != 0xffffffffu && t.hit != 0xffffffffu) {
            // a triangle in front of the quad (or the quad itself, which ends the matter as well)
            const DLight& L = sc.lights[want_light];
            if (t.hit != (uint32_t)L.tri0 && t.hit != (uint32_t)L.tri1) {
                have_ray = have_task = false;
                t.G = make_uint2(0u, 0u);
                t.T = make_uint2(0u, 0u);
                t.sp = 0;
            }
        }
        if (done) {
            have_ray = false;
            have_task = false;
            uint32_t tri = t.hit;
            float tsearch = t.tmax;
            bool on_light = false;
            if (want != 0xffffffffu) {
                // searched up to the quad: the light is the closest hit unless another triangle was accepted
                const DLight& L = sc.lights[want_light];
                on_light = tri == 0xffffffffu || tri == (uint32_t)L.tri0 || tri == (uint32_t)L.tri1;
                if (tri == 0xffffffffu) tri = want;  // (t.tmax is still the quad's distance then)
            }
            if (tri != 0xffffffffu && (want == 0xffffffffu || on_light)) {
                const V3 S = t.o, dir = t.d;
                const float tmax = refine_distance(sc.tm.tris, tri, S, dir, t.tmin, tsearch);
                const uint32_t info = f2u(q.PB[task].w);
                const float4 w4 = q.PC[task];
                const int light = (int)(info & 0xffffu);
                const int kind = (int)(info >> 16);
                const int medium = (int)f2u(w4.w) - 1;
                const DMaterial& M = medium_of(sc, medium);
                const DLight& L = sc.lights[light];
                const V3 w = v3(w4.x, w4.y, w4.z);
                int obj, prim;
                V3 Ng = tri_normal(sc.tm.tris, tri, &obj, &prim);
                V3 hp = point_at(S, tmax, dir);
                if (stage == 1) {
                    if (obj == light) {
                        V3 att = attenuation_mul(sc.mats[sc.n_objects], mag(hp - S)) * attenuation_mul(M, seg0);
                        contrib = (1.0f - F0) * (w * ld3(L.Li) * att);
                        lit = true;
                    }
                } else if (kind == 0) {
                    if (obj == light) {
                        V3 att = attenuation_mul(M, mag(hp - S));
                        float alt = light_pdf(L, hp, S);
                        float mis = WAPT_MIS_CONST / (WAPT_MIS_CONST + alt);
                        if (mis != mis) mis = 1.0f;
                        contrib = mis * (w * ld3(L.Li) * att);
                        lit = true;
                    }
                } else if (obj == medium) {
                    V3 N = normalized(Ng);
                    float F;
                    V3 D1 = refract_dir(N, -dir, M.ior, &F);
                    V3 dir1 = -D1;
                    if (!is_zero(D1) && finite_dir(dir1)) {
                        seg0 = mag(hp - S);
                        F0 = F;
                        stage = 1;
                        t.o = hp;
                        t.d = dir1;
                        want_light = light;
                        aim_pending = true;  // (the lane keeps its task; the second search starts at the top of the next trip)
                        have_task = true;
                    }
                }
            }
        }
        wave_add_rgb(q.sum, astage, lit, lit ? f2u(q.VO[vol_sample_of(q, task)].w) : 0u, contrib);
    }
    rays = wave_sum(rays);
    decided = wave_sum(decided);
    if (lane_id() == 0 && rays) atomicAdd(&q.rays[3], (unsigned long long)rays);
    if (lane_id() == 0 && decided) atomicAdd(&q.rays[7], (unsigned long long)decided);
    if (COUNT) {
        uint32_t a = wave_sum(tc.nodes), b = wave_sum(tc.tris);
        if (lane_id() == 0) {
            atomicAdd(&q.rays[16], (unsigned long long)a);
            atomicAdd(&q.rays[17], (unsigned long long)b);
            atomicAdd(&q.rays[18], (unsigned long long)tc.wsteps);
            atomicAdd(&q.rays[19], (unsigned long long)tc.nsteps);
            atomicAdd(&q.rays[20], (unsigned long long)tc.ttrips);
        }
    }
    if (tc.overflow) atomicAdd(&q.cnt[CNT_OVERFLOW], tc.overflow);
}

// Batch closest-hit for wapt_intersect*
template <bool COUNT>
__global__ __launch_bounds__(WAPT_BLOCK) void k_intersect(TraceMem tm, const float4* rays, uint32_t n, float* hits,
                                                          unsigned long long* counters, uint32_t* feed_counters) {
    __shared__ uint2 s_stack[WAPT_LDS_STACK * WAPT_BLOCK];
    __shared__ CoopLds s_coop[WAPT_BLOCK / 64];
    CoopLds* coop = &s_coop[threadIdx.x >> 6];
    uint32_t glane = blockIdx.x * blockDim.x + threadIdx.x;
    TraceCount tc = {};
    WaveFeed feed = wave_feed(n, feed_counters);
    Trav t = {};
    bool have = false;
    uint32_t cur = 0;
    for (;;) {
        uint32_t idx = feed.take(!have);
        if (idx != 0xffffffffu) {
            float4 a = rays[2 * (size_t)idx], b = rays[2 * (size_t)idx + 1];
            trav_begin(t, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w, b.w);
            cur = idx;
            have = true;
        }
        if (!__any(have)) break;
        if (trav_step<COUNT, false, WAPT_COOP>(tm, t, have, &s_stack[threadIdx.x], glane, tc, coop)) {
            have = false;
            float* h = hits + 6 * (size_t)cur;
            h[0] = trav_finish(tm, t);
            if (t.hit == 0xffffffffu) {
                h[1] = u2f(0xffffffffu);
                h[2] = u2f(0xffffffffu);
                h[3] = h[4] = h[5] = 0.f;
            } else {
                int obj, prim;
                V3 Ng = tri_normal(tm.tris, t.hit, &obj, &prim);
                h[1] = u2f((uint32_t)prim);
                h[2] = u2f((uint32_t)obj);
                h[3] = Ng.x;
                h[4] = Ng.y;
                h[5] = Ng.z;
            }
        }
    }
    if (COUNT) {
        uint32_t a = wave_sum(tc.nodes), b = wave_sum(tc.tris), c = wave_sum(tc.spills);
        if (lane_id() == 0) {
            atomicAdd(&counters[0], (unsigned long long)a);
            atomicAdd(&counters[1], (unsigned long long)b);
            atomicAdd(&counters[2], (unsigned long long)c);
            atomicAdd(&counters[4], (unsigned long long)tc.wsteps);
            atomicAdd(&counters[5], (unsigned long long)tc.nsteps);
            atomicAdd(&counters[6], (unsigned long long)tc.ttrips);
        }
    }
    uint32_t ov = wave_sum(tc.overflow);
    if (lane_id() == 0 && ov) atomicAdd(&counters[3], (unsigned long long)ov);
}

// =============================================================== host side

#define HIPCHK(call)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (call);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return wapt_fail(WAPT_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e));              \
    } while (0)

// ---- device memory that exists without any allocation (wapt_intersect_device_static) ------------------------
// One array per GPU, part of the loaded code object, held until the process ends.  Layout:
//   [0, 8 KiB)        32 work-feed counters of the traversal kernel, one per 256-byte line
//   [8 KiB, 16 KiB)   8 x 64-bit result counters (word 3: traversal-stack pushes dropped)
//   [16 KiB, 32 MiB)  spill area of the traversal stacks: (tree depth - WAPT_LDS_STACK) entries per lane of the launch
// In host code the array's name is only a handle for hipGetSymbolAddress: each GPU has its own copy at its own
// address, looked up once per device (static_scratch_of) and used through that pointer everywhere else.
#define WAPT_STATIC_SCRATCH_BYTES (32u << 20)
#define WAPT_STATIC_HEADER_BYTES (16u << 10)
#define WAPT_STATIC_MAX_RAYS (1u << 30)
static __device__ __align__(256) unsigned char scratch[WAPT_STATIC_SCRATCH_BYTES];

namespace {
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[64] = {};  // per device

// Address of device `device`'s copy of the array; the device must be current.
int static_scratch_of(int device, unsigned char** out) {
    if (device < 0 || device >= 64) return wapt_fail(WAPT_E_LIMIT, "device index beyond 63");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return wapt_fail(WAPT_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return WAPT_OK;
}
}  // namespace

struct wapt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    Bvh8 bvh_info_only;  // nodes/tris vectors are released after upload; counts kept below
    uint64_t n_nodes = 0, n_tris = 0;
    uint32_t bvh_depth = 0;
    float sah = 0.f;
    int builder = 0;
    float build_seconds = 0.f;
    KScene ks;
    // owned device memory
    void *d_nodes = nullptr, *d_tris = nullptr, *d_mats = nullptr, *d_lights = nullptr, *d_spill = nullptr;
    uint32_t* d_cnt = nullptr;
    unsigned long long* d_rays = nullptr;
    bool all_plain = false, all_plain_foggy = false;
    uint32_t* h_cnt = nullptr;  // pinned
    unsigned long long* h_u64 = nullptr;  // pinned: 24 ray counters + 3 words of the march-sample count
    KQueues kq;
    void* d_queue_block = nullptr;
    size_t queue_bytes = 0;
    int queue_depths = 0;
    uint32_t grid = 0, sgrid = 0, cus = 1;
    uint32_t capR = 0, capS = 0, capP = 0;
    bool has_scatter = false, has_inside = false;
    float max_segment = 0.f;   // longest possible segment (everything lies in a ball around the world centre)
    float max_inside = 0.f;    // longest box diagonal of an object whose inside scatters
    bool global_scatters = false;
    int n_lights = 0;
    int n_reach = 0;  // lights that a probe ray can hit (DLight::probe_slot)
    wapt_camera camera;  // as described by the caller (the progressive interface re-derives lower resolutions from it)
    std::vector<hipEvent_t> ev_pool;
    float* d_sum = nullptr;      // W*H*3 sums of wapt_render (kept between calls)
    unsigned char* static_scratch = nullptr;  // this GPU's copy of the file-scope scratch array (wapt_intersect_device_static)
    std::vector<struct wapt_progressive*> progressives;  // live progressive handles on this scene
};

namespace {

uint32_t env_u32(const char* name, uint32_t def) {
    const char* v = getenv(name);
    if (!v || !*v) return def;
    long long x = atoll(v);
    return x > 0 ? (uint32_t)x : def;
}

DMaterial to_device(const wapt_material& m) {
    DMaterial d;
    memset(&d, 0, sizeof(d));
    memcpy(d.kcol, (m.flags & WAPT_MAT_DIFFUSE) ? m.kd : m.ks, 12);
    d.kt_nonzero = (m.kt[0] != 0.f || m.kt[1] != 0.f || m.kt[2] != 0.f) ? 1u : 0u;
    memcpy(d.ss, m.ss, 12);
    memcpy(d.st, m.st, 12);
    d.ior = m.ior;
    d.flags = m.flags;
    bool ss0 = m.ss[0] == 0.f && m.ss[1] == 0.f && m.ss[2] == 0.f;
    d.scatters = (!(m.flags & WAPT_MAT_CLEAR) && !ss0) ? 1u : 0u;
    return d;
}

void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// wa_areaLight / wa_sphereLight constructors (lights.cpp:24-34, 136-143)
DLight to_device(const wapt_light& l) {
    DLight d;
    memset(&d, 0, sizeof(d));
    d.type = l.type;
    memcpy(d.P, l.P, 12);
    const double pi = 3.1415926535;
    if (l.type == 0) {
        memcpy(d.U, l.U, 12);
        memcpy(d.V, l.V, 12);
        float n[3];
        cross3(l.U, l.V, n);
        float m = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        for (int a = 0; a < 3; a++) d.N[a] = n[a] / m;
        d.area = sqrtf(l.U[0] * l.U[0] + l.U[1] * l.U[1] + l.U[2] * l.U[2]) *
                 sqrtf(l.V[0] * l.V[0] + l.V[1] * l.V[1] + l.V[2] * l.V[2]);
        for (int a = 0; a < 3; a++) {
            float irr = (float)(0.5 * l.power[a]) / d.area;
            d.Li[a] = (float)(irr / pi);
            d.centre[a] = l.P[a] + 0.5f * (l.U[a] + l.V[a]);
        }
    } else {
        d.radius = l.radius;
        for (int a = 0; a < 3; a++) {
            float irr = (float)(l.power[a] / (4.0 * pi * (double)l.radius * (double)l.radius));
            d.Li[a] = (float)(irr / pi);
            d.centre[a] = l.P[a];
        }
    }
    return d;
}

// wa_camera constructor + setResolution (fundamentals.cpp:227-263)
DCamera to_device(const wapt_camera& c) {
    DCamera d;
    memset(&d, 0, sizeof(d));
    d.type = c.type;
    d.resI = c.resI;
    d.resJ = c.resJ;
    memcpy(d.P, c.P, 12);
    memcpy(d.D, c.D, 12);
    float side[3], up[3];
    cross3(c.D, c.up, side);
    float m = sqrtf(side[0] * side[0] + side[1] * side[1] + side[2] * side[2]);
    for (int a = 0; a < 3; a++) side[a] /= m;
    cross3(side, c.D, up);
    const double pi = 3.1415926535;
    float sensorDist = (float)(0.5 * c.sensor_width / tan(0.5 * c.fov_deg * (pi / 180.0)));
    float aspect = (float)c.resJ / (float)c.resI;
    for (int a = 0; a < 3; a++) {
        d.side[a] = side[a];
        d.up[a] = up[a];
        d.sensorOrigin[a] = (float)(c.P[a] + sensorDist * c.D[a] - 0.5 * c.sensor_width * (side[a] + aspect * up[a]));
    }
    d.pixelWidth = c.sensor_width / (float)c.resI;
    d.sensorWidth = c.sensor_width;
    return d;
}

const uint64_t WAPT_MAX_TRIS = 0xffffffffull / 3, WAPT_MAX_NODES = 0xffffffffull / (WAPT_NODE_BYTES / 16);

int validate_desc(const wapt_scene_desc* d) {
    if (!d) return wapt_fail(WAPT_E_INVALID, "null scene description");
    if (d->n_tris == 0 || !d->tri_verts || !d->tri_obj) return wapt_fail(WAPT_E_INVALID, "scene has no triangles");
    if (d->n_objects <= 0 || !d->materials) return wapt_fail(WAPT_E_INVALID, "scene has no objects");
    if (d->n_lights < 0 || d->n_lights > d->n_objects || (d->n_lights > 0 && !d->lights))
        return wapt_fail(WAPT_E_INVALID, "bad light table");
    if (d->n_lights > 0xffff) return wapt_fail(WAPT_E_LIMIT, "more than 65535 lights");
    if (d->n_objects >= (1 << 20) - 1) return wapt_fail(WAPT_E_LIMIT, "more than 2^20-2 objects");
    if (d->camera.resI <= 0 || d->camera.resJ <= 0) return wapt_fail(WAPT_E_INVALID, "bad resolution");
    if ((uint64_t)d->camera.resI * (uint64_t)d->camera.resJ >= (1ull << 30))
        return wapt_fail(WAPT_E_LIMIT, "image larger than 2^30 pixels");
    // the traversal addresses triangle and node records with 32-bit arithmetic (3 and 5 sixteen-byte words each)
    if (d->n_tris > WAPT_MAX_TRIS) return wapt_fail(WAPT_E_LIMIT, "more than (2^32-1)/3 triangles");
    for (uint64_t t = 0; t < d->n_tris; t++)
        if (d->tri_obj[t] < 0 || d->tri_obj[t] >= d->n_objects)
            return wapt_fail(WAPT_E_INVALID, "triangle refers to an object outside the material table");
    if (!(d->world_radius > 0.f)) return wapt_fail(WAPT_E_INVALID, "world radius must be positive");
    return WAPT_OK;
}

}  // namespace

// Flag byte of a triangle record (traverse.h) from the flags of its object's material.
static uint32_t tri_flags_of(const wapt_material& m) {
    const uint32_t mf = m.flags;
    uint32_t f = 0;
    if (mf & WAPT_MAT_OPAQUE) {
        f |= WAPT_TRI_OPAQUE;
        if ((mf & WAPT_MAT_CLEAR) && !(mf & (WAPT_MAT_DIELECTRIC | WAPT_MAT_DIELECTRICSSS))) f |= WAPT_TRI_PLAIN;
    }
    if ((mf & (WAPT_MAT_DIELECTRIC | WAPT_MAT_DIELECTRICSSS)) && m.ior > 1.0f) f |= WAPT_TRI_GLASS;
    return f;
}
// A shadow walk in a scene of such objects (under a clear global medium) is over with ONE search: the object is plain
// (opaque, clear inside, no dielectric) or a dielectric that is not opaque and clear inside - nothing attenuates on
// either side of its surface, its front faces end the walk if its ior is above 1 and are passed otherwise.
static bool one_search_object(const wapt_material& m) {
    if (tri_flags_of(m) & WAPT_TRI_PLAIN) return true;
    if (!WAPT_SHADOW_COOP) return false;  // (only the cooperative any-hit phase leaves out the surfaces that are passed)
    return (m.flags & WAPT_MAT_DIELECTRIC) && !(m.flags & WAPT_MAT_DIELECTRICSSS) && (m.flags & WAPT_MAT_CLEAR) && !(m.flags & WAPT_MAT_OPAQUE);
}

extern "C" int wapt_scene_create(const wapt_scene_desc* desc, int32_t device, wapt_scene** out) {
    return wapt_scene_create_with(desc, device, WAPT_BUILDER_AUTO, out);
}

static int scene_create_with(const wapt_scene_desc* desc, int32_t device, int32_t builder, wapt_scene** out);

extern "C" int wapt_scene_create_with(const wapt_scene_desc* desc, int32_t device, int32_t builder, wapt_scene** out) {
    return wapt_guarded([&] { return scene_create_with(desc, device, builder, out); });
}

static int scene_create_with(const wapt_scene_desc* desc, int32_t device, int32_t builder, wapt_scene** out) {
    if (!out) return wapt_fail(WAPT_E_INVALID, "null output");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (builder != WAPT_BUILDER_AUTO && builder != WAPT_BUILDER_HOST_SAH && builder != WAPT_BUILDER_DEVICE_LBVH)
        return wapt_fail(WAPT_E_INVALID, "unknown BVH builder");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return wapt_fail(WAPT_E_NO_DEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= ndev) return wapt_fail(WAPT_E_INVALID, "device index out of range");
    bool fallback_to_host = false;  // the builder was not asked for by name: a device build that fails may be redone on the host
    if (builder == WAPT_BUILDER_AUTO) {
        // The device builder (locally-ordered clustering over the Morton order) is the default at every size: its trees
        // trace as fast as or faster than the host's binned-SAH trees (dragonChrome +4 %, dragonGlass +2 %, Cornell +2 %;
        // dragonMilk -3 %: rays that start inside the mesh) and take milliseconds to seconds instead of seconds to minutes.
        const char* e = getenv("WAPT_BVH_BUILDER");
        if (e && !strcmp(e, "host")) builder = WAPT_BUILDER_HOST_SAH;
        else if (e && !strcmp(e, "device")) builder = WAPT_BUILDER_DEVICE_LBVH;
        else {
            builder = desc->n_tris >= (uint64_t)env_u32("WAPT_DEVICE_BUILD_MIN_TRIS", 1u) ? WAPT_BUILDER_DEVICE_LBVH : WAPT_BUILDER_HOST_SAH;
            fallback_to_host = builder == WAPT_BUILDER_DEVICE_LBVH;
        }
    }

    HIPCHK(hipSetDevice(device));
    // (the holder frees whatever has been allocated so far on every way out, an exception included)
    std::unique_ptr<wapt_scene, void (*)(wapt_scene*)> holder(new wapt_scene(), wapt_scene_destroy);
    wapt_scene* const s = holder.get();
    s->device = device;
    auto bail = [&](int code) { return code; };
#define HIPCHK_S(call)                                                                                       \
    do {                                                                                                     \
        hipError_t _e = (call);                                                                              \
        if (_e != hipSuccess) return bail(wapt_fail(WAPT_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e))); \
    } while (0)

    HIPCHK_S(hipStreamCreate(&s->stream));
    // this GPU's copy of the static scratch array, looked up once per device with the device current (a failure is
    // reported by the one entry point that needs the array, wapt_intersect_device_static)
    if (static_scratch_of(device, &s->static_scratch) != WAPT_OK) s->static_scratch = nullptr;
    auto tb0 = std::chrono::steady_clock::now();
    float blo[3], bhi[3];
    if (builder == WAPT_BUILDER_DEVICE_LBVH) {
        std::vector<uint8_t> opaque((size_t)desc->n_objects);
        for (int i = 0; i < desc->n_objects; i++) opaque[i] = (uint8_t)tri_flags_of(desc->materials[i]);
        DeviceBvh8 db;
        std::string berr;
        const bool built = bvh8_build_device(desc->tri_verts, desc->tri_obj, desc->tri_prim, desc->n_tris, opaque.data(),
                                             (uint32_t)desc->n_objects, s->stream, &db, &berr);
        if (built) {
            s->d_nodes = db.d_nodes;
            s->d_tris = db.d_tris;
            s->n_nodes = db.n_nodes;
            s->n_tris = db.n_tris;
            s->bvh_depth = db.max_depth;
            s->sah = db.sah_cost;
            memcpy(blo, db.lo, 12);
            memcpy(bhi, db.hi, 12);
        }
        if (!built || s->bvh_depth > WAPT_STACK_CAP) {
            // The device builder gave up (clustering stalled, tree too deep to collapse, out of device memory) or its tree
            // is deeper than the traversal stack.  A caller who named the builder gets the error; the default choice
            // falls back to the host's binned-SAH builder, which handles such inputs.
            const std::string why = built ? "tree deeper than the traversal stack (" + std::to_string(s->bvh_depth) + ")" : berr;
            if (!fallback_to_host)
                return bail(wapt_fail(built ? WAPT_E_LIMIT : WAPT_E_INVALID, "device BVH build: " + why));
            (void)hipFree(s->d_nodes);
            (void)hipFree(s->d_tris);
            s->d_nodes = s->d_tris = nullptr;
            if (getenv("WAPT_BVH_VERBOSE")) fprintf(stderr, "wapt: device BVH build failed (%s): building on the host\n", why.c_str());
            builder = WAPT_BUILDER_HOST_SAH;
        }
    }
    if (builder == WAPT_BUILDER_HOST_SAH) {
        Bvh8 bvh;
        const char* err = nullptr;
        if (!bvh8_build(desc->tri_verts, desc->tri_obj, desc->tri_prim, desc->n_tris, &bvh, &err))
            return bail(wapt_fail(WAPT_E_INVALID, std::string("BVH build: ") + err));
        if (bvh.max_depth > WAPT_STACK_CAP)
            return bail(wapt_fail(WAPT_E_LIMIT, "BVH deeper than the traversal stack (" + std::to_string(bvh.max_depth) + ")"));
        s->n_nodes = bvh.nodes.size();
        s->n_tris = bvh.tris.size();
        s->bvh_depth = bvh.max_depth;
        s->sah = bvh.sah_cost;
        memcpy(blo, bvh.lo, 12);
        memcpy(bhi, bvh.hi, 12);
        // last word of a triangle record: what shadow walks need to know about its object without fetching the material
        for (auto& T : bvh.tris) T.pad = tri_flags_of(desc->materials[T.obj]) | ((uint32_t)T.obj << WAPT_TRI_OBJ_SHIFT);
        HIPCHK_S(hipMalloc(&s->d_nodes, bvh.nodes.size() * sizeof(Bvh8Node)));
        HIPCHK_S(hipMalloc(&s->d_tris, bvh.tris.size() * sizeof(Bvh8Tri)));
        HIPCHK_S(hipMemcpy(s->d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(Bvh8Node), hipMemcpyHostToDevice));
        HIPCHK_S(hipMemcpy(s->d_tris, bvh.tris.data(), bvh.tris.size() * sizeof(Bvh8Tri), hipMemcpyHostToDevice));
    }
    if (s->n_nodes > WAPT_MAX_NODES || s->n_tris > WAPT_MAX_TRIS)
        return bail(wapt_fail(WAPT_E_LIMIT, "BVH larger than the traversal's 32-bit record addressing allows"));
    if (getenv("WAPT_BVH_DUMP")) {
        // the first levels of the tree as text (debugging aid): per node its children's kind and box
        const uint32_t n_dump = (uint32_t)std::min<uint64_t>(s->n_nodes, env_u32("WAPT_BVH_DUMP", 9));
        std::vector<Bvh8Node> top(n_dump);
        HIPCHK_S(hipMemcpy(top.data(), s->d_nodes, n_dump * sizeof(Bvh8Node), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_dump; i++) {
            const Bvh8Node& n = top[i];
            fprintf(stderr, "node %u: child_base %u tri_base %u\n", i, n.child_base, n.tri_base);
            const uint8_t* ql[3] = {n.qlo_x, n.qlo_y, n.qlo_z};
            const uint8_t* qh[3] = {n.qhi_x, n.qhi_y, n.qhi_z};
            for (int c = 0; c < 8; c++) {
                if (!n.meta[c]) continue;
                const bool inner = (n.meta[c] >> 5) == 1 && (n.meta[c] & 31) >= 24;
                float lo[3], hi[3];
                for (int a = 0; a < 3; a++) {
                    const float cell = std::ldexp(1.0f, (int)n.e[a] - 127);
                    lo[a] = n.p[a] + ql[a][c] * cell;
                    hi[a] = n.p[a] + qh[a][c] * cell;
                }
                const int cnt = inner ? 0 : ((n.meta[c] >> 5) == 7 ? 3 : (n.meta[c] >> 5) == 3 ? 2 : 1);
                fprintf(stderr, "   slot %d %s%d  [%.3f %.3f %.3f] - [%.3f %.3f %.3f]\n", c, inner ? "inner " : "leaf x", cnt, lo[0], lo[1], lo[2],
                        hi[0], hi[1], hi[2]);
            }
        }
    }
    s->builder = builder;

    std::vector<DMaterial> mats((size_t)desc->n_objects + 1);
    for (int i = 0; i < desc->n_objects; i++) mats[i] = to_device(desc->materials[i]);
    mats[desc->n_objects] = to_device(desc->global_medium);
    s->has_scatter = false;
    s->has_inside = false;
    for (int i = 0; i <= desc->n_objects; i++) {
        if (mats[i].scatters) {
            s->has_scatter = true;
            if (i < desc->n_objects) s->has_inside = true;
        }
    }
    s->global_scatters = mats[desc->n_objects].scatters != 0u;
    if (s->has_inside) {
        // box diagonal of every object with a scattering inside: a segment that lies inside a closed object is no
        // longer (sizes the launches of scenes with media, see chunk_for)
        std::vector<float> lo((size_t)desc->n_objects * 3, FLT_MAX), hi((size_t)desc->n_objects * 3, -FLT_MAX);
        for (uint64_t t = 0; t < desc->n_tris; t++) {
            const int o = desc->tri_obj[t];
            if (!mats[o].scatters) continue;
            const float* v = desc->tri_verts + 9 * t;
            for (int k = 0; k < 9; k++) {
                lo[(size_t)o * 3 + k % 3] = std::min(lo[(size_t)o * 3 + k % 3], v[k]);
                hi[(size_t)o * 3 + k % 3] = std::max(hi[(size_t)o * 3 + k % 3], v[k]);
            }
        }
        double dmax = 0.0;
        for (int o = 0; o < desc->n_objects; o++) {
            if (!mats[o].scatters || lo[(size_t)o * 3] > hi[(size_t)o * 3]) continue;
            double dx = (double)hi[o * 3] - lo[o * 3], dy = (double)hi[o * 3 + 1] - lo[o * 3 + 1], dz = (double)hi[o * 3 + 2] - lo[o * 3 + 2];
            dmax = std::max(dmax, std::sqrt(dx * dx + dy * dy + dz * dz));
        }
        s->max_inside = (float)(dmax * 1.001);
    }
    HIPCHK_S(hipMalloc(&s->d_mats, mats.size() * sizeof(DMaterial)));
    HIPCHK_S(hipMemcpy(s->d_mats, mats.data(), mats.size() * sizeof(DMaterial), hipMemcpyHostToDevice));
    // longest possible segment: everything lives inside the ball around the world
    // centre that holds the world sphere, the geometry and the camera
    {
        double rb = desc->world_radius;
        for (int cx = 0; cx < 8; cx++) {
            double dx = ((cx & 1) ? bhi[0] : blo[0]) - desc->world_center[0];
            double dy = ((cx & 2) ? bhi[1] : blo[1]) - desc->world_center[1];
            double dz = ((cx & 4) ? bhi[2] : blo[2]) - desc->world_center[2];
            rb = std::max(rb, std::sqrt(dx * dx + dy * dy + dz * dz));
        }
        double dx = desc->camera.P[0] - desc->world_center[0], dy = desc->camera.P[1] - desc->world_center[1],
               dz = desc->camera.P[2] - desc->world_center[2];
        rb = std::max(rb, std::sqrt(dx * dx + dy * dy + dz * dz));
        s->max_segment = (float)(2.0 * rb * 1.001);
    }

    std::vector<DLight> lights((size_t)std::max(desc->n_lights, 1));
    for (int i = 0; i < desc->n_lights; i++) lights[i] = to_device(desc->lights[i]);
    // DLight::front_clear.  A shadow search resolves distances to about 0.002 + 1e-4 x its length; a walk towards a point
    // B on the quad may end its search that far short of B if nothing but the quad can be met there.  Walks arrive from
    // the front (lights.cpp:91 rejects the rest) at |cos| >= 0.25, so a surface that stays on or behind the quad's plane
    // (to within 2.5e-4: TRACE_BIAS x 0.25) is met after B or with it, which counts as arrived (cpp:775).  The light is
    // clear if every other object's triangle near the quad (boxes within the margin) is such a surface.
    // (the test itself runs on the GPU over the tree's triangle records, below: k_front_clear)
    const double fc_reach = 0.002 + 1e-4 * (double)s->max_segment + 1e-3;
    bool fc_test = false;
    {
        const uint64_t budget = 4000000000ull;  // triangle-light pairs looked at (a scene with thousands of lights skips this)
        int n_area = 0;
        for (int i = 0; i < desc->n_lights; i++) n_area += desc->lights[i].type == 0;
        const char* cut_env = getenv("WAPT_SHADOW_CUT");  // WAPT_SHADOW_CUT=0: every walk searches up to its end point
        if (n_area > 0 && (uint64_t)n_area * desc->n_tris <= budget && !(cut_env && cut_env[0] == '0')) {
            fc_test = true;
            for (int l = 0; l < desc->n_lights; l++) {
                const DLight& L = lights[l];
                lights[l].front_clear = (L.type == 0 && std::isfinite(L.N[0]) && std::isfinite(L.N[1]) && std::isfinite(L.N[2])) ? 1 : 0;
            }
        }
    }
    HIPCHK_S(hipMalloc(&s->d_lights, lights.size() * sizeof(DLight)));
    HIPCHK_S(hipMemcpy(s->d_lights, lights.data(), lights.size() * sizeof(DLight), hipMemcpyHostToDevice));
    s->n_lights = desc->n_lights;
    if (desc->n_lights > 0)
        hipLaunchKernelGGL(k_light_tris, dim3((unsigned)((s->n_tris + 255) / 256)), dim3(256), 0, s->stream, (const float4*)s->d_tris,
                           (uint32_t)s->n_tris, desc->n_lights, (DLight*)s->d_lights);
    if (desc->n_lights > 0 && fc_test)
        hipLaunchKernelGGL(k_front_clear, dim3((unsigned)((s->n_tris + 255) / 256)), dim3(256), 0, s->stream, (const float4*)s->d_tris,
                           (uint32_t)s->n_tris, desc->n_lights, (DLight*)s->d_lights, fc_reach);
    // DLight::probe_slot: the lights that a probe ray can have as its closest hit (those with triangles in the tree;
    // all of them when WAPT_GENERAL_PROBE=1 asks for plain searches), numbered in order
    if (desc->n_lights > 0) {
        HIPCHK_S(hipStreamSynchronize(s->stream));
        HIPCHK_S(hipMemcpy(lights.data(), s->d_lights, lights.size() * sizeof(DLight), hipMemcpyDeviceToHost));
        const bool general_probe = env_u32("WAPT_GENERAL_PROBE", 0) != 0;
        for (int i = 0; i < desc->n_lights; i++) lights[i].probe_slot = (general_probe || lights[i].ntri != 0) ? s->n_reach++ : -1;
        HIPCHK_S(hipMemcpy(s->d_lights, lights.data(), lights.size() * sizeof(DLight), hipMemcpyHostToDevice));
    }

    hipDeviceProp_t prop;
    HIPCHK_S(hipGetDeviceProperties(&prop, device));
    uint32_t cus = (uint32_t)std::max(prop.multiProcessorCount, 1);
    s->cus = cus;
    // (never fewer blocks than work-feed shards: the groups of a shard that no block maps to would never be drawn)
    s->grid = std::max<uint32_t>(cus * env_u32("WAPT_BLOCKS_PER_CU", 8), WAPT_FEED_SHARDS);
    s->sgrid = std::max(1u, cus * env_u32("WAPT_SBLOCKS_PER_CU", 2) * (1024u / WAPT_SBLOCK));  // blocks of the shading kernels
    HIPCHK_S(hipMalloc((void**)&s->d_cnt, CNT_TOTAL * sizeof(uint32_t)));
    HIPCHK_S(hipMemset(s->d_cnt, 0, CNT_TOTAL * sizeof(uint32_t)));
    HIPCHK_S(hipMalloc((void**)&s->d_rays, 24 * sizeof(unsigned long long)));
    HIPCHK_S(hipMemset(s->d_rays, 0, 24 * sizeof(unsigned long long)));
    HIPCHK_S(hipHostMalloc((void**)&s->h_cnt, CNT_TOTAL * sizeof(uint32_t) + 64));
    HIPCHK_S(hipHostMalloc((void**)&s->h_u64, 28 * sizeof(unsigned long long)));

    memset(&s->ks, 0, sizeof(s->ks));
    s->ks.tm.nodes = (const uint4*)s->d_nodes;
    s->ks.tm.tris = (const float4*)s->d_tris;
    s->ks.tm.spill = nullptr;  // allocated with the queues (ensure_spill)
    s->ks.tm.spill_stride = 0;
    s->ks.tm.stack_cap = WAPT_LDS_STACK;
    s->ks.tm.tri_quorum = (int32_t)env_u32("WAPT_TRI_QUORUM", 1);
    s->ks.tm.n_nodes = (uint32_t)s->n_nodes;
    s->ks.tm.n_tris = (uint32_t)s->n_tris;
    s->ks.mats = (const DMaterial*)s->d_mats;
    s->ks.lights = (const DLight*)s->d_lights;
    s->ks.n_lights = desc->n_lights;
    s->ks.n_objects = desc->n_objects;
    s->ks.has_inside = s->has_inside ? 1 : 0;
    // WAPT_GENERAL_SHADOW=1 sends every shadow walk through the step-by-step path (the tests compare the two)
    s->ks.fast_shadow = env_u32("WAPT_GENERAL_SHADOW", 0) ? 0 : 1;
    // every object opaque, clear and no dielectric, the global medium clear: the one-search shadow kernel
    s->all_plain = s->ks.fast_shadow != 0 && (desc->global_medium.flags & WAPT_MAT_CLEAR) != 0;
    s->all_plain_foggy = s->ks.fast_shadow != 0 && (desc->global_medium.flags & WAPT_MAT_CLEAR) == 0;  // the same under a medium
    for (int i = 0; i < desc->n_objects; i++) {
        if (!one_search_object(desc->materials[i])) s->all_plain = false;
        if (!(tri_flags_of(desc->materials[i]) & WAPT_TRI_PLAIN)) s->all_plain_foggy = false;  // (a medium outside glass: the segments differ)
    }
    // WAPT_GENERAL_PROBE=1: every probe ray gets the plain closest-hit search (the tests compare the two)
    s->ks.fast_probe = env_u32("WAPT_GENERAL_PROBE", 0) ? 0 : 1;
    s->ks.cam = to_device(desc->camera);
    s->camera = desc->camera;
    memcpy(s->ks.wc, desc->world_center, 12);
    s->ks.wr = desc->world_radius;
    memset(&s->kq, 0, sizeof(s->kq));
    HIPCHK_S(hipDeviceSynchronize());
    // (tree, upload, and the tests of the lights against the tree's triangles)
    s->build_seconds = (float)std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
    *out = holder.release();
    return WAPT_OK;
#undef HIPCHK_S
}

// Cuts a progressive handle loose from a scene that is going away: its device buffers are freed and every later
// call on it reports an error instead of touching the freed scene (wapt_progressive_destroy still deletes it).
static void progressive_orphan(wapt_progressive* p);

extern "C" void wapt_scene_destroy(wapt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (wapt_progressive* p : s->progressives) progressive_orphan(p);
    s->progressives.clear();
    (void)hipFree(s->d_sum);
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    (void)hipFree(s->d_nodes);
    (void)hipFree(s->d_tris);
    (void)hipFree(s->d_mats);
    (void)hipFree(s->d_lights);
    (void)hipFree(s->d_spill);
    (void)hipFree(s->d_cnt);
    (void)hipFree(s->d_rays);
    (void)hipFree(s->d_queue_block);
    if (s->h_cnt) (void)hipHostFree(s->h_cnt);
    if (s->h_u64) (void)hipHostFree(s->h_u64);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" int wapt_scene_bvh_info(const wapt_scene* s, wapt_bvh_info* o) {
    if (!s || !o) return wapt_fail(WAPT_E_INVALID, "null argument");
    o->n_nodes = s->n_nodes;
    o->n_tris = s->n_tris;
    o->node_size = sizeof(Bvh8Node);
    o->tri_size = sizeof(Bvh8Tri);
    o->node_bytes = s->n_nodes * sizeof(Bvh8Node);
    o->tri_bytes = s->n_tris * sizeof(Bvh8Tri);
    o->max_depth = s->bvh_depth;
    o->sah_cost = s->sah;
    o->builder = (uint32_t)s->builder;
    o->build_seconds = s->build_seconds;
    return WAPT_OK;
}

namespace {

// Spill area of the traversal stacks: (tree depth - LDS entries) slots for every lane of a traversal launch
// (all of them use the same fixed grid).
int ensure_spill(wapt_scene* s) {
    uint32_t lanes = s->grid * WAPT_BLOCK;
    if (s->d_spill && s->ks.tm.spill_stride >= lanes) return WAPT_OK;
    if (s->d_spill) HIPCHK(hipFree(s->d_spill));
    s->d_spill = nullptr;
    int depth_slots = std::max(1, (int)s->bvh_depth + 1 - WAPT_LDS_STACK);
    depth_slots = std::min(depth_slots, WAPT_STACK_CAP - WAPT_LDS_STACK);
    HIPCHK(hipMalloc(&s->d_spill, (size_t)depth_slots * lanes * sizeof(uint2)));
    s->ks.tm.spill = (uint2*)s->d_spill;
    s->ks.tm.spill_stride = lanes;
    s->ks.tm.stack_cap = WAPT_LDS_STACK + depth_slots;
    return WAPT_OK;
}

// What one task of depth d emits at most.  A task marches one segment: through the global medium (any length up to
// max_segment) or inside an object (closed objects: no longer than the object's box diagonal); a march sample in the
// open sends a shadow walk per light sample and a probe per light that has geometry, one inside an object whose inside
// scatters 16 probes per light.  A segment that lies inside an OPEN mesh can exceed that bound: the render loop checks
// the exact number of march samples of a step before it emits anything and refuses the step then.
struct FanOut {
    uint64_t marches, kids, shadows, probes;
};
FanOut fan_out(const wapt_scene* s, const wapt_render_params* prm, int d) {
    FanOut f;
    const uint64_t r_out = s->global_scatters ? (uint64_t)((double)s->max_segment * prm->mr / (d + 1.0)) + 1 : 0;
    const uint64_t r_in = s->has_inside ? (uint64_t)((double)s->max_inside * prm->mr / (d + 1.0)) + 1 : 0;
    const uint64_t L = (uint64_t)s->n_lights;
    f.marches = std::max(r_out, r_in);
    f.kids = (uint64_t)std::max(prm->hsamples, 1) + f.marches;
    f.shadows = L * prm->lsamples * (1 + f.marches);  // (a segment in an object's medium that ends on a front face samples the lights too)
    f.probes = std::max((uint64_t)s->n_reach * std::max(r_out, r_in), 16 * L * r_in);
    return f;
}

// Device memory a render call may use for its queues: five eighths of what is free (its own block included), or
// WAPT_QUEUE_BUDGET_GB if that is less.
uint64_t queue_budget(const wapt_scene* s, uint64_t* avail_out) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    const uint64_t avail = (uint64_t)(free_b + s->queue_bytes);
    if (avail_out) *avail_out = avail;
    uint64_t budget = avail / 8 * 5;
    if (const char* e = getenv("WAPT_QUEUE_BUDGET_GB")) {
        const double gb = atof(e);
        if (gb > 0.0) budget = std::min<uint64_t>(budget, (uint64_t)(gb * 1e9));
    }
    return budget;
}

// (Re)allocates the work queues for a trace depth and a call that issues `paths` top-level tasks; `f0` is the fan-out
// of a task of depth 0.
// Capacities: large launches matter more than anything else for the traversal kernels (a launch ends
// with a tail in which most lanes have run out of work: at 2^22 tasks per launch the lanes of the
// closest-hit kernel were busy 55 % of the time, at 2^26 90 %; whole frame, dragonChrome 1080p: 2^26 2931,
// 2^27 3023, 2^28 3055 Mpaths/s; later kernels: 2^28 3400, 2^29 - the 256-spp frame in one piece - 3520), and HBM is
// plentiful, so the default is 2^29 tasks per depth (24 GiB per depth, 135 GB with the shadow queue and the hit
// records at td 3); scenes with media get the largest launch that the budget allows, with the queues in the
// proportions of a task's worst-case fan-out; all of it is cut down to what the call can use and to the budget
// (queue_budget).
int ensure_queues(wapt_scene* s, int td, uint64_t paths, const FanOut& f0, int lsamples) {
    int depths = td + 1;
    const uint64_t L = (uint64_t)std::max(s->n_lights, 0);
    const uint64_t paths_r = (paths + 4095) & ~(uint64_t)4095;
    // (64 KB of it set aside for the small fixed lists and the alignment of the blocks, so that the allocation stays
    //  within what queue_budget() names)
    const uint64_t budget_all = queue_budget(s, nullptr);
    const uint64_t budget = budget_all > (1ull << 17) ? budget_all - (1ull << 16) : budget_all;
    auto cap32 = [](uint64_t v) { return (uint32_t)std::min<uint64_t>(v, 0x7ffff000u); };
    // capacities in entries: R per depth, S bodies, SA (origins of the surface kernel's walks), P bodies, march samples
    // of a step (QM, VO), segments / tasks of a step (G0-G3, hit records)
    uint32_t capR, capS, capSA, capP, capM, capG;
    if (!s->has_scatter) {
        capR = env_u32("WAPT_RAD_CAP", 1u << 29);
        capR = (uint32_t)std::min<uint64_t>(capR, paths_r * std::max<uint64_t>(f0.kids, 1));
        capR = std::max(capR, 64u * 64u) & ~63u;
        // cut to the budget: R per depth, the shadow queue (one entry per task and light, four lights' worth at most)
        // and the hit records
        const uint64_t per_light = (uint64_t)std::max(1, std::min(s->n_lights, 4));
        const uint64_t per_task = (uint64_t)depths * 48 + per_light * 48 + 8;
        if ((uint64_t)capR * per_task > budget) capR = std::max<uint32_t>((uint32_t)(budget / per_task) & ~63u, 64u * 64u);
        capS = env_u32("WAPT_SHADOW_CAP", (uint32_t)std::min<uint64_t>((uint64_t)capR * per_light, 0xfffff000u));
        capSA = capS;
        capP = capM = 64u;
        capG = capR;
    } else {
        // A launch of c tasks of depth 0 needs c x kids slots in R[1], c x shadows in S, c x probes in P and c x marches
        // entries for the march samples (the deeper launches need less per task and are as many tasks larger): the
        // budget goes to the queues in these proportions, which gives the largest launch the memory allows.
        const uint64_t kids = std::max<uint64_t>(f0.kids, 1), sh = std::max<uint64_t>(f0.shadows, 1), pr = f0.probes,
                       ma = std::max<uint64_t>(f0.marches, 1);
        const uint64_t g_per = 4;  // segments and hit records: the launches of the deepest level are about td + 1 times as large
        const uint64_t per_task = (uint64_t)depths * kids * 48 + sh * 32 + g_per * L * lsamples * 16 + pr * 32 + ma * 24 + g_per * (64 + 8);
        // (not more than the call can use: its paths x kids tasks is what the second level can hold at most - the
        //  launches of the deeper levels, where most of the work of a low-spp frame is, grow with c too)
        uint64_t c = std::min<uint64_t>(paths_r * kids, budget / per_task);
        if (const uint32_t rc = env_u32("WAPT_RAD_CAP", 0)) c = std::min<uint64_t>(c, std::max<uint64_t>(rc / kids, 1));
        c = std::max<uint64_t>(c, 4096);
        capR = std::max(cap32(c * kids) & ~63u, 64u * 64u);
        capS = env_u32("WAPT_SHADOW_CAP", cap32(c * sh));
        capP = std::max(env_u32("WAPT_PROBE_CAP", cap32(c * pr)), 64u);
        capM = std::max(cap32(c * ma), 64u);
        capG = (uint32_t)std::min<uint64_t>(capR, std::max<uint64_t>(c * g_per, 4096));
        capSA = (uint32_t)std::min<uint64_t>(capS, (uint64_t)capG * std::max<uint64_t>(L * lsamples, 1));
    }
    // A block that is a little smaller than what this call would get is kept (the call runs in slightly smaller
    // chunks): the capacities of a scene with media follow the free memory, which moves between calls (memory of
    // processes that have ended is returned in the background), and hipFree of a block of this size took 0.2-2 s.
    auto enough = [](uint32_t have, uint32_t want) { return have >= want - want / 4; };
    if (s->d_queue_block && s->queue_depths >= depths && enough(s->capR, capR) && enough(s->capS, capS) && enough(s->capP, capP) &&
        enough(s->kq.capSA, capSA) && enough(s->kq.capQM, capM) && enough(s->kq.capG, capG))
        return WAPT_OK;
    if (s->d_queue_block) {
        // keep what an earlier, larger call allocated for the dimensions that do not grow
        capR = std::max(capR, s->capR);
        capS = std::max(capS, s->capS);
        capP = std::max(capP, s->capP);
        capSA = std::max(capSA, s->kq.capSA);
        capM = std::max(capM, s->kq.capQM);
        capG = std::max(capG, s->kq.capG);
        depths = std::max(depths, s->queue_depths);
        HIPCHK(hipFree(s->d_queue_block));
        s->d_queue_block = nullptr;
        s->queue_bytes = 0;
    }
    size_t bytes = 0;
    auto take = [&](size_t b) {
        size_t off = bytes;
        bytes += (b + 255) & ~(size_t)255;
        return off;
    };
    const bool media = s->has_scatter;
    size_t oRA = take((size_t)depths * capR * 16), oRB = take((size_t)depths * capR * 16), oRC = take((size_t)depths * capR * 16);
    size_t oSA = take((size_t)capSA * 16), oSB = take((size_t)capS * 16), oSC = take((size_t)capS * 16);
    size_t oPB = take((size_t)capP * 16), oPC = take((size_t)capP * 16);
    size_t oH = take((size_t)capG * 8);
    size_t oQM = take((size_t)capM * 8), oVO = take((size_t)capM * 16);
    const size_t gN = media ? capG : 16;
    size_t oG0 = take(gN * 16), oG1 = take(gN * 16), oG2 = take(gN * 16), oG3 = take(gN * 16);
    HIPCHK(hipMalloc(&s->d_queue_block, bytes));
    char* b = (char*)s->d_queue_block;
    KQueues& q = s->kq;
    q.RA = (float4*)(b + oRA);
    q.RB = (float4*)(b + oRB);
    q.RC = (float4*)(b + oRC);
    q.SA = (float4*)(b + oSA);
    q.SB = (float4*)(b + oSB);
    q.SC = (float4*)(b + oSC);
    q.PB = (float4*)(b + oPB);
    q.PC = (float4*)(b + oPC);
    q.hits = (uint2*)(b + oH);
    q.QM = (uint2*)(b + oQM);
    q.VO = (float4*)(b + oVO);
    q.G0 = (float4*)(b + oG0);
    q.G1 = (float4*)(b + oG1);
    q.G2 = (float4*)(b + oG2);
    q.G3 = (float4*)(b + oG3);
    q.capQM = capM;
    q.capG = capG;
    q.capR = capR;
    q.capS = capS;
    q.capSA = capSA;
    q.capP = capP;
    q.cnt = s->d_cnt;
    q.rays = s->d_rays;
    s->capR = capR;
    s->capS = capS;
    s->capP = capP;
    s->queue_bytes = bytes;
    s->queue_depths = depths;
    return ensure_spill(s);
}

struct Timer {
    wapt_scene* s;
    bool on;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> recs;
    size_t next = 0;
    hipEvent_t get() {
        if (next == s->ev_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            s->ev_pool.push_back(e);
        }
        return s->ev_pool[next++];
    }
    template <class F>
    void run(int family, F&& f) {
        if (!on) {
            f();
            return;
        }
        hipEvent_t a = get(), b = get();
        (void)hipEventRecord(a, s->stream);
        f();
        (void)hipEventRecord(b, s->stream);
        recs.push_back({family, {a, b}});
    }
};

enum { FAM_RAYGEN, FAM_TRACE, FAM_HEAD, FAM_SHADE, FAM_VOLUME, FAM_SHADOW, FAM_PROBE, FAM_N };

}  // namespace

namespace {
// What a render call looks at: the scene's camera and the whole image, unless the progressive
// interface has lowered the resolution or set a region; `abort` is polled between launches.
struct RenderView {
    DCamera cam;
    int l, r, b, t;  // pixel rectangle [l, r) x [b, t)
    const std::atomic<int>* abort;
};
int render_accumulate_view(wapt_scene* s, const wapt_render_params* prm, float* d_sum, wapt_stats* stats, const RenderView* view);
}  // namespace

extern "C" int wapt_render_accumulate(wapt_scene* s, const wapt_render_params* prm, float* d_sum, wapt_stats* stats) {
    return wapt_guarded([&] { return render_accumulate_view(s, prm, d_sum, stats, nullptr); });
}

namespace {
int render_accumulate_view(wapt_scene* s, const wapt_render_params* prm, float* d_sum, wapt_stats* stats, const RenderView* view) {
    if (!s || !prm || !d_sum) return wapt_fail(WAPT_E_INVALID, "null argument");
    if (prm->psamples <= 0 || prm->hsamples <= 0 || prm->lsamples <= 0)
        return wapt_fail(WAPT_E_INVALID, "psamples, hsamples and lsamples must be positive");
    if (prm->max_depth < 0 || prm->max_depth > 62) return wapt_fail(WAPT_E_LIMIT, "max_depth must be in [0, 62]");
    if (prm->sample_begin < 0 || prm->sample_count < 0) return wapt_fail(WAPT_E_INVALID, "bad sample range");
    if (!(prm->mr >= 0.f) || !(prm->me >= 0.f)) return wapt_fail(WAPT_E_INVALID, "mr and me must be >= 0");
    HIPCHK(hipSetDevice(s->device));
    const DCamera cam = view ? view->cam : s->ks.cam;
    const uint64_t n_paths = (uint64_t)((cam.resI + 7) / 8) * (uint64_t)((cam.resJ + 7) / 8) * 64u * (uint64_t)prm->sample_count;
    {
        const FanOut f0 = fan_out(s, prm, 0);
        int rc = ensure_queues(s, prm->max_depth, n_paths, f0, prm->lsamples);
        if (rc) return rc;
    }
    KScene ks = s->ks;  // (after ensure_queues: it sets the spill area and stack capacity of ks.tm)
    ks.cam = cam;
    auto t0 = std::chrono::steady_clock::now();

    const int td = prm->max_depth;
    const int W = ks.cam.resI, H = ks.cam.resJ;
    KParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.hsamples = prm->hsamples;
    kp.lsamples = prm->lsamples;
    kp.td = td;
    kp.me = prm->me;
    kp.mr = prm->mr;
    kp.seed = prm->seed;
    kp.sample_begin = prm->sample_begin;
    kp.tilesX = (uint32_t)((W + 7) / 8);
    kp.slots_per_sample = kp.tilesX * (uint32_t)((H + 7) / 8) * 64u;
    kp.rl = view ? view->l : 0;
    kp.rr = view ? view->r : W;
    kp.rb = view ? view->b : 0;
    kp.rt = view ? view->t : H;
    const bool whole = kp.rl <= 0 && kp.rb <= 0 && kp.rr >= W && kp.rt >= H;
    kp.dense_raygen = (W % 8 == 0 && H % 8 == 0 && whole) ? 1 : 0;
    kp.sample_count = (uint32_t)std::max(prm->sample_count, 1);
    const char* pm_env = getenv("WAPT_PIXEL_MAJOR");
    kp.write_hits = s->has_scatter ? 1 : 0;
    KQueues q = s->kq;
    q.sum = d_sum;
    q.debug_pixel = env_u32("WAPT_DEBUG_PIXEL", 0xffffffffu);
    hipStream_t st = s->stream;
    const dim3 grid(s->grid), block(WAPT_BLOCK);
    const dim3 sgrid(s->sgrid), sblock(WAPT_SBLOCK);
    const dim3 vgrid(s->cus * env_u32("WAPT_VBLOCKS_PER_CU", 2048u / WAPT_VBLOCK)), vblock(WAPT_VBLOCK);
    const dim3 volgrid(s->cus * env_u32("WAPT_VOLBLOCKS_PER_CU", 2048u / WAPT_VOLBLOCK)), volblock(WAPT_VOLBLOCK);

    // fan-out of one task of depth d (sizes the chunk taken from R[d])
    const int L = s->n_lights;
    auto chunk_for = [&](int d) -> uint32_t {
        const FanOut f = fan_out(s, prm, d);
        uint64_t c = std::min(s->capR, s->kq.capG);
        if (d < td) c = std::min<uint64_t>(c, s->capR / f.kids);
        if (f.shadows) c = std::min<uint64_t>(c, s->capS / f.shadows);
        if (L > 0) c = std::min<uint64_t>(c, s->kq.capSA / ((uint64_t)L * prm->lsamples));
        if (f.probes) c = std::min<uint64_t>(c, s->capP / f.probes);
        if (f.marches) c = std::min<uint64_t>(c, s->kq.capQM / f.marches);
        return (uint32_t)c;
    };
    for (int d = 0; d <= td; d++)
        if (chunk_for(d) == 0)
            return wapt_fail(WAPT_E_LIMIT, "queue capacity too small for the fan-out of this scene and these settings");

    // Camera tasks of a scene without media (whole image, dimensions multiples of 8) are never stored: the kernels of
    // depth 0 make them from their slot number.  A chunk of camera tasks then holds whole samples
    // (WAPT_FUSE_CAMERA=0 keeps the ray generation pass, for comparison).
    const char* fuse_env = getenv("WAPT_FUSE_CAMERA");
    bool fuse_camera = kp.dense_raygen && !s->has_scatter && !(fuse_env && fuse_env[0] == '0') && s->capR < (1u << 31);
    // Task order: pixel by pixel (dragonChrome 1080p 256 spp: closest-hit and surface kernel -3 %, shadow kernel -6 %
    // against sample by sample; glass +2 %, fog and milk +1 %).  WAPT_PIXEL_MAJOR=0 / 1 overrides.
    kp.pixel_major = pm_env ? (pm_env[0] != '0') : (kp.sample_count > 1u);
    if (fuse_camera && (kp.pixel_major ? s->capR < kp.sample_count : s->capR < kp.slots_per_sample)) fuse_camera = false;
    udiv_magic(kp.pixel_major ? kp.sample_count : kp.slots_per_sample, &kp.div_slots_m, &kp.div_slots_k);
    udiv_magic(kp.tilesX, &kp.div_tiles_m, &kp.div_tiles_k);
    Timer tm{s, (prm->profile & 1) != 0, {}, 0};
    std::vector<uint32_t> have((size_t)td + 2, 0u);
    const unsigned long long total = (unsigned long long)kp.slots_per_sample * (unsigned long long)prm->sample_count;
    unsigned long long issued = 0;
    uint64_t rays_trace = 0, n_trace_launches = 0, n_shadow_launches = 0, empties_seen = 0, empties_kept = 0;
    bool aborted = false;
    HIPCHK(hipMemsetAsync(s->d_rays, 0, 24 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_reset, dim3(1), dim3(96), 0, st, s->d_cnt, CNT_OVERFLOW, 0u);
    hipLaunchKernelGGL(k_reset, dim3(1), dim3(96), 0, st, s->d_cnt, CNT_QOVERFLOW, 0u);

    for (;;) {
        int d = -1;
        for (int k = td; k >= 0; k--)
            if (have[k]) {
                d = k;
                break;
            }
        if (d < 0) {
            if (view && view->abort && view->abort->load()) aborted = true;  // in-flight paths are finished, no new ones start
            if (issued >= total || aborted) break;
            uint32_t n = (uint32_t)std::min<unsigned long long>(s->capR, total - issued);
            if (fuse_camera) {
                // (chunks hold whole samples or, pixel by pixel, all samples of whole image slots)
                const unsigned long long unit = kp.pixel_major ? kp.sample_count : kp.slots_per_sample;
                n = (uint32_t)std::min<unsigned long long>((unsigned long long)(s->capR / unit) * unit, total - issued);
                kp.cam_s0 = (uint32_t)(issued / unit);
            }
            if (!fuse_camera) {
                hipLaunchKernelGGL(k_reset, dim3(1), dim3(96), 0, st, s->d_cnt, CNT_R0, kp.dense_raygen ? n : 0u);
                tm.run(FAM_RAYGEN, [&] { hipLaunchKernelGGL(k_raygen, sgrid, sblock, 0, st, ks, q, kp, issued, n); });
            }
            issued += n;
            if (kp.dense_raygen) {
                have[0] = n;
            } else {
                HIPCHK(hipMemcpyAsync(s->h_cnt, s->d_cnt, CNT_FEED * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                have[0] = std::min(s->h_cnt[CNT_R0], s->capR);
            }
            continue;
        }
        uint32_t n = std::min(have[d], chunk_for(d));
        uint32_t start = have[d] - n;
        have[d] = start;
        for (;;) {
            hipLaunchKernelGGL(k_reset, dim3(1), dim3(96), 0, st, s->d_cnt, d < td ? CNT_R(d + 1) : -1, 0u);
            const bool cam = fuse_camera && d == 0;
            tm.run(FAM_TRACE, [&] {
                if (cam && (prm->profile & 2)) hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, st, ks, q, d, start, n, kp);
                else if (cam) hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, st, ks, q, d, start, n, kp);
                else if (prm->profile & 2) hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, st, ks, q, d, start, n, kp);
                else hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, st, ks, q, d, start, n, kp);
            });
            if (!s->has_inside) break;
            // Scenes with a scattering object: chunk_for() bounds a segment inside an object by the object's box,
            // which an open mesh can exceed.  The exact number of march samples of the step is counted before anything
            // is shaded; a step that would not fit is cut down and traced again.
            unsigned long long* h_m = s->h_u64 + 24;  // (a copy into pageable memory goes through the runtime's staging path)
            HIPCHK(hipMemsetAsync(s->d_rays + 14, 0, 2 * sizeof(unsigned long long), st));
            tm.run(FAM_VOLUME, [&] { hipLaunchKernelGGL(k_count, vgrid, vblock, 0, st, ks, q, kp, d, start, n, s->d_rays + 14); });
            HIPCHK(hipMemcpyAsync(h_m, s->d_rays + 13, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const uint64_t M = h_m[1], kids_surface = (uint64_t)n * (uint64_t)std::max(prm->hsamples, 1);
            const bool fits = M <= s->kq.capQM && (d >= td || kids_surface + M <= s->capR);
            // (slots without a task that the closest-hit launch met: those of a launch that is repeated do not count)
            if (fits) empties_kept += h_m[0] - empties_seen;
            empties_seen = h_m[0];
            if (fits) break;
            if (n <= 64)
                return wapt_fail(WAPT_E_LIMIT, "the march samples of 64 tasks (" + std::to_string(M) + ") do not fit the queues");
            double ratio = (double)s->kq.capQM / (double)M;
            if (d < td) ratio = std::min(ratio, (double)s->capR / (double)(kids_surface + M));
            uint32_t n_new = std::max(64u, (uint32_t)((double)n * ratio * 0.8) & ~63u);
            n_new = std::min(n_new, n - 64u);
            start += n - n_new;
            have[d] = start;
            n = n_new;
        }
        rays_trace += n;
        n_trace_launches++;
        tm.run(FAM_SHADE, [&] {
            const bool one = prm->lsamples == 1 && prm->hsamples == 1;
            const bool cam = fuse_camera && d == 0;
            if (cam && one) hipLaunchKernelGGL((k_surface<true, false, true>), sgrid, sblock, 0, st, ks, q, kp, d, start, n, ks.mats, ks.lights);
            else if (cam) hipLaunchKernelGGL((k_surface<false, false, true>), sgrid, sblock, 0, st, ks, q, kp, d, start, n, ks.mats, ks.lights);
            else if (one && !kp.write_hits) hipLaunchKernelGGL((k_surface<true, false>), sgrid, sblock, 0, st, ks, q, kp, d, start, n, ks.mats, ks.lights);
            else if (one) hipLaunchKernelGGL((k_surface<true, true>), sgrid, sblock, 0, st, ks, q, kp, d, start, n, ks.mats, ks.lights);
            else if (!kp.write_hits) hipLaunchKernelGGL((k_surface<false, false>), sgrid, sblock, 0, st, ks, q, kp, d, start, n, ks.mats, ks.lights);
            else hipLaunchKernelGGL((k_surface<false, true>), sgrid, sblock, 0, st, ks, q, kp, d, start, n, ks.mats, ks.lights);
        });
        // the walks of the shadow queue: those the surface kernel sent and, beyond the first s1 entries, `vol_count` entries
        // of the march samples of the current slice (vol_ms samples, or 0: no such entries, the queue counter says it all)
        float4* vo_now = q.VO;  // origin records of the slice of march samples whose walks are in the shadow queue
        auto run_shadow = [&](uint32_t s1, uint32_t vol_ms, uint32_t vol_count) {
            KQueues qs = q;
            if (vol_ms) qs.VO = vo_now;
            qs.vol_s1 = s1;
            qs.vol_ms = vol_ms;
            qs.vol_scount = vol_count;
            if (vol_ms) udiv_magic(vol_ms, &qs.vol_div_m, &qs.vol_div_k);
            tm.run(FAM_SHADOW, [&] {
                if (s->all_plain) {
                    if (prm->profile & 2) hipLaunchKernelGGL((k_shadow<true, true>), grid, block, 0, st, ks, qs);
                    else hipLaunchKernelGGL((k_shadow<false, true>), grid, block, 0, st, ks, qs);
                } else if (s->all_plain_foggy) {
                    if (prm->profile & 2) hipLaunchKernelGGL((k_shadow<true, true, true>), grid, block, 0, st, ks, qs);
                    else hipLaunchKernelGGL((k_shadow<false, true, true>), grid, block, 0, st, ks, qs);
                } else {
                    if (prm->profile & 2) hipLaunchKernelGGL((k_shadow<true, false>), grid, block, 0, st, ks, qs);
                    else hipLaunchKernelGGL((k_shadow<false, false>), grid, block, 0, st, ks, qs);
                }
            });
            n_shadow_launches++;
        };
        bool shadows_pending = true;  // (what the surface kernel sent)
        uint64_t kids_total = 0;
        if (s->has_scatter) {
            // The march samples of the step are listed before any of them emits: the step's queues were sized by
            // bounds (fan_out), the counts are exact.  The samples are marched in slices that fit the shadow and probe
            // queues (normally one per list), each followed by the kernels that drain those queues.
            tm.run(FAM_VOLUME, [&] {
                if (s->has_inside) hipLaunchKernelGGL(k_expand<true>, vgrid, vblock, 0, st, q);
                else hipLaunchKernelGGL(k_expand<false>, vgrid, vblock, 0, st, q);
            });
            HIPCHK(hipMemcpyAsync(s->h_cnt, s->d_cnt, CNT_FEED * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const uint64_t M_out = s->h_cnt[CNT_QM], M_in = s->h_cnt[CNT_QMI], M = M_out + M_in;
            uint64_t s_used = std::min(s->h_cnt[CNT_S], s->capS);
            const uint64_t kids_now = d < td ? s->h_cnt[CNT_R(d + 1)] : 0;
            if (M > s->kq.capQM || (d < td && kids_now + M > s->capR))
                return wapt_fail(WAPT_E_LIMIT, "a step has more march samples (" + std::to_string(M) +
                                                   ") than the queues were sized for: a segment inside an object is longer than the object's box");
            kids_total = kids_now + (d < td ? M : 0);
            const uint64_t sh_per = (uint64_t)L * prm->lsamples;
            for (int inside = 0; inside < 2; inside++) {
                const uint64_t Ml = inside ? M_in : M_out;
                const uint64_t pr_per = inside ? (uint64_t)L * 16 : (uint64_t)s->n_reach;
                uint64_t m0 = 0;
                while (m0 < Ml) {
                    uint64_t ms = Ml - m0;
                    if (!inside && sh_per) ms = std::min(ms, (s->capS - s_used) / sh_per);
                    if (pr_per) ms = std::min(ms, (uint64_t)s->capP / pr_per);
                    if (ms == 0) {
                        if (inside || s_used == 0) return wapt_fail(WAPT_E_LIMIT, "shadow or probe queue smaller than the fan-out of one march sample");
                        run_shadow((uint32_t)s_used, 0u, 0u);  // make room: drain what the surface kernel sent
                        hipLaunchKernelGGL(k_reset_sp, dim3(1), dim3(64), 0, st, s->d_cnt, 1);
                        s_used = 0;
                        shadows_pending = false;
                        continue;
                    }
                    KQueues qv = q;
                    qv.vol_s1 = (uint32_t)s_used;
                    qv.vol_ms = (uint32_t)ms;
                    udiv_magic(qv.vol_ms, &qv.vol_div_m, &qv.vol_div_k);
                    qv.vol_scount = inside ? 0u : (uint32_t)(ms * sh_per);
                    qv.vol_pcount = (uint32_t)(ms * pr_per);
                    const uint32_t kid0 = d < td ? (uint32_t)(kids_now + (inside ? M_out : 0) + m0) : 0xffffffffu;
                    // origin and pixel of a sample are plane A of its child task: where children exist the samples'
                    // records ARE that plane of R[d + 1] (kids_now + M <= capR was checked above), written once
                    if (d < td) qv.VO = q.RA + (size_t)(d + 1) * s->capR + kid0;
                    vo_now = qv.VO;
                    tm.run(FAM_VOLUME, [&] {
                        if (inside) hipLaunchKernelGGL(k_volume<true>, volgrid, volblock, 0, st, ks, qv, kp, d, (uint32_t)m0, (uint32_t)ms, kid0, ks.lights);
                        else hipLaunchKernelGGL(k_volume<false>, volgrid, volblock, 0, st, ks, qv, kp, d, (uint32_t)m0, (uint32_t)ms, kid0, ks.lights);
                    });
                    if (qv.vol_pcount)
                        tm.run(FAM_PROBE, [&] {
                            if (prm->profile & 2) hipLaunchKernelGGL(k_probe<true>, grid, block, 0, st, ks, qv);
                            else hipLaunchKernelGGL(k_probe<false>, grid, block, 0, st, ks, qv);
                        });
                    m0 += ms;
                    // (the walks read their origins from the slice's samples: they run before the next slice is marched)
                    if (!inside) {
                        run_shadow((uint32_t)s_used, (uint32_t)ms, qv.vol_scount);
                        s_used = 0;
                        shadows_pending = false;
                    }
                    hipLaunchKernelGGL(k_reset_sp, dim3(1), dim3(64), 0, st, s->d_cnt, shadows_pending ? 0 : 1);
                }
            }
        }
        if (shadows_pending) run_shadow(0u, 0u, 0u);
        if (d < td) {
            if (s->has_scatter) {
                have[d + 1] = (uint32_t)std::min<uint64_t>(kids_total, s->capR);
            } else {
                HIPCHK(hipMemcpyAsync(s->h_cnt, s->d_cnt, CNT_FEED * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                have[d + 1] = std::min(s->h_cnt[CNT_R(d + 1)], s->capR);
            }
        }
    }
    unsigned long long* h_rays = s->h_u64;
    HIPCHK(hipMemcpyAsync(s->h_cnt, s->d_cnt, CNT_FEED * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_rays, s->d_rays, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->paths = (uint64_t)W * H * (uint64_t)prm->sample_count;
        rays_trace -= s->has_inside ? empties_kept : h_rays[13];  // (slots of the deeper queues that held no task)
        stats->rays = rays_trace + h_rays[0] + h_rays[3];
        stats->stack_overflows = s->h_cnt[CNT_OVERFLOW] + s->h_cnt[CNT_QOVERFLOW];
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->n_trace_launches = n_trace_launches;
        stats->n_trace_rays = rays_trace;
        stats->n_shadow_launches = n_shadow_launches;
        stats->n_shadow_rays = h_rays[0];
        stats->n_probe_rays = h_rays[3];
        stats->queries_decided = h_rays[7];
        stats->trace_node_visits = h_rays[1];
        stats->trace_tri_tests = h_rays[2];
        stats->trace_wave_steps = h_rays[4];
        stats->trace_node_steps = h_rays[5];
        stats->trace_tri_trips = h_rays[6];
        stats->shadow_node_visits = h_rays[8];
        stats->shadow_tri_tests = h_rays[9];
        stats->shadow_wave_steps = h_rays[10];
        stats->shadow_node_steps = h_rays[11];
        stats->shadow_tri_trips = h_rays[12];
        stats->probe_node_visits = h_rays[16];
        stats->probe_tri_tests = h_rays[17];
        stats->probe_wave_steps = h_rays[18];
        stats->probe_node_steps = h_rays[19];
        stats->probe_tri_trips = h_rays[20];
        stats->queue_bytes = s->queue_bytes;
        stats->queue_tasks = s->capR;
        if (tm.on) {
            double ms[FAM_N] = {0};
            for (auto& r : tm.recs) {
                float e = 0.f;
                (void)hipEventElapsedTime(&e, r.second.first, r.second.second);
                ms[r.first] += e;
            }
            stats->ms_raygen = ms[FAM_RAYGEN];
            stats->ms_trace = ms[FAM_TRACE];
            stats->ms_head = ms[FAM_HEAD];
            stats->ms_shade = ms[FAM_SHADE];
            stats->ms_volume = ms[FAM_VOLUME];
            stats->ms_shadow = ms[FAM_SHADOW];
            stats->ms_probe = ms[FAM_PROBE];
        }
    }
    if (s->h_cnt[CNT_OVERFLOW] || s->h_cnt[CNT_QOVERFLOW])
        return wapt_fail(WAPT_E_LIMIT, "work dropped: " + std::to_string(s->h_cnt[CNT_OVERFLOW]) + " traversal-stack pushes, " +
                                           std::to_string(s->h_cnt[CNT_QOVERFLOW]) + " queue pushes");
    if (aborted) return wapt_fail(WAPT_E_ABORTED, "render aborted");
    return WAPT_OK;
}
}  // namespace

__global__ void k_resolve(const float* sum, float* out, size_t n, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sum[i] * scale;
}

static int render_impl(wapt_scene* s, const wapt_render_params* prm, float* fb, wapt_stats* stats) {
    if (!s || !prm || !fb) return wapt_fail(WAPT_E_INVALID, "null argument");
    if (prm->sample_begin != 0 || prm->sample_count != prm->psamples)
        return wapt_fail(WAPT_E_INVALID, "wapt_render renders all samples: sample_begin=0, sample_count=psamples");
    HIPCHK(hipSetDevice(s->device));
    const size_t n = (size_t)s->ks.cam.resI * s->ks.cam.resJ * 3;
    if (!s->d_sum) HIPCHK(hipMalloc((void**)&s->d_sum, n * sizeof(float)));  // (the resolution of a scene is fixed)
    HIPCHK(hipMemsetAsync(s->d_sum, 0, n * sizeof(float), s->stream));
    int rc = render_accumulate_view(s, prm, s->d_sum, stats, nullptr);
    if (rc != WAPT_OK) return rc;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->d_sum, s->d_sum, n,
                       1.0f / (float)prm->psamples);
    HIPCHK(hipMemcpyAsync(fb, s->d_sum, n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return WAPT_OK;
}

extern "C" int wapt_render(wapt_scene* s, const wapt_render_params* prm, float* fb, wapt_stats* stats) {
    return wapt_guarded([&] { return render_impl(s, prm, fb, stats); });
}

// ---- progressive interface (udp_renderer::reset / setIterationCount / setRenderResolution / setRegion /
// abort and the running average of renderTile, cpp:68-76, 141-161, 907-972)

// internal = it/(it+1) * internal + 1/(it+1) * (sum / psamples) for the pixels of the region (cpp:141-145)
__global__ void k_blend(float* avg, const float* sum, int renW, int dispW, int l, int r, int b, int t, float w_old, float w_new) {
    int i = l + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = b + (int)blockIdx.y;
    if (i >= r || j >= t) return;
    size_t src = ((size_t)j * renW + i) * 3, dst = ((size_t)j * dispW + i) * 3;
    for (int c = 0; c < 3; c++) avg[dst + c] = w_old * avg[dst + c] + w_new * sum[src + c];
}

// frame buffer block [i*rf, (i+1)*rf) x [j*rf, (j+1)*rf) <- internal(i, j) for the pixels of the region (cpp:152-159)
__global__ void k_replicate(float* fb, const float* avg, int dispW, int dispH, int rf, int l, int r, int b, int t) {
    int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y;
    if (x >= dispW || y >= dispH) return;
    int i = x / rf, j = y / rf;
    if (i < l || i >= r || j < b || j >= t) return;
    size_t src = ((size_t)j * dispW + i) * 3, dst = ((size_t)y * dispW + x) * 3;
    for (int c = 0; c < 3; c++) fb[dst + c] = avg[src + c];
}

struct wapt_progressive {
    wapt_scene* s = nullptr;   // null once the scene has been destroyed
    int device = 0;
    wapt_render_params prm;
    wapt_camera cam;           // display camera
    int dispW = 0, dispH = 0, renW = 0, renH = 0, rf = 1;
    int l = 0, r = 0, b = 0, t = 0;
    int it = 0;
    float *d_avg = nullptr, *d_sum = nullptr, *d_fb = nullptr;  // display-sized
    std::atomic<int> abort_flag{0};
};

static void progressive_free_buffers(wapt_progressive* p) {
    (void)hipSetDevice(p->device);
    (void)hipFree(p->d_avg);
    (void)hipFree(p->d_sum);
    (void)hipFree(p->d_fb);
    p->d_avg = p->d_sum = p->d_fb = nullptr;
}

static void progressive_orphan(wapt_progressive* p) {
    progressive_free_buffers(p);
    p->s = nullptr;
}

#define PROG_ALIVE(p)                                                                              \
    do {                                                                                           \
        if (!(p)) return wapt_fail(WAPT_E_INVALID, "null argument");                               \
        if (!(p)->s) return wapt_fail(WAPT_E_INVALID, "the scene of this progressive handle has been destroyed"); \
    } while (0)

static int progressive_create(wapt_scene* s, const wapt_render_params* prm, wapt_progressive** out) {
    if (!s || !prm || !out) return wapt_fail(WAPT_E_INVALID, "null argument");
    *out = nullptr;
    if (prm->psamples <= 0) return wapt_fail(WAPT_E_INVALID, "psamples must be positive");
    HIPCHK(hipSetDevice(s->device));
    std::unique_ptr<wapt_progressive, void (*)(wapt_progressive*)> p(new wapt_progressive(), [](wapt_progressive* q) {
        progressive_free_buffers(q);
        delete q;
    });
    p->s = s;
    p->device = s->device;
    p->prm = *prm;
    p->cam = s->camera;
    p->dispW = p->renW = s->camera.resI;
    p->dispH = p->renH = s->camera.resJ;
    p->r = p->renW;
    p->t = p->renH;
    size_t bytes = (size_t)p->dispW * p->dispH * 3 * sizeof(float);
    HIPCHK(hipMalloc((void**)&p->d_avg, bytes));
    HIPCHK(hipMalloc((void**)&p->d_sum, bytes));
    HIPCHK(hipMalloc((void**)&p->d_fb, bytes));
    HIPCHK(hipMemset(p->d_avg, 0, bytes));
    HIPCHK(hipMemset(p->d_fb, 0, bytes));
    s->progressives.push_back(p.get());
    *out = p.release();
    return WAPT_OK;
}

extern "C" int wapt_progressive_create(wapt_scene* s, const wapt_render_params* prm, wapt_progressive** out) {
    return wapt_guarded([&] { return progressive_create(s, prm, out); });
}

extern "C" void wapt_progressive_destroy(wapt_progressive* p) {
    if (!p) return;
    if (p->s) {
        auto& v = p->s->progressives;
        v.erase(std::remove(v.begin(), v.end(), p), v.end());
        (void)hipSetDevice(p->device);
        if (p->s->stream) (void)hipStreamSynchronize(p->s->stream);
    }
    progressive_free_buffers(p);
    delete p;
}

extern "C" int wapt_progressive_reset(wapt_progressive* p) {
    PROG_ALIVE(p);
    HIPCHK(hipSetDevice(p->s->device));
    p->it = 0;
    HIPCHK(hipMemsetAsync(p->d_avg, 0, (size_t)p->dispW * p->dispH * 3 * sizeof(float), p->s->stream));
    return WAPT_OK;
}

extern "C" int wapt_progressive_set_iteration(wapt_progressive* p, int32_t it) {
    if (!p || it < 0) return wapt_fail(WAPT_E_INVALID, "bad iteration count");
    p->it = it;
    return WAPT_OK;
}

extern "C" int wapt_progressive_set_resolution(wapt_progressive* p, int32_t downres_pow2) {
    PROG_ALIVE(p);
    if (downres_pow2 < 0 || downres_pow2 > 12) return wapt_fail(WAPT_E_INVALID, "bad resolution exponent");
    // lowerResolution (cpp:924-934)
    int w_new = (int)((float)p->dispW / powf(2.0f, (float)downres_pow2));
    int h_new = (int)((float)p->dispH / powf(2.0f, (float)downres_pow2));
    if (w_new < 1 || h_new < 1) return wapt_fail(WAPT_E_INVALID, "resolution exponent too large for this image");
    if ((p->dispW % w_new) != 0) w_new++;
    if ((p->dispH % h_new) != 0) h_new++;
    p->renW = w_new;
    p->renH = h_new;
    p->rf = 1 << downres_pow2;
    p->l = p->b = 0;
    p->r = w_new;
    p->t = h_new;
    return WAPT_OK;
}

extern "C" int wapt_progressive_set_region(wapt_progressive* p, int32_t l, int32_t r, int32_t b, int32_t t) {
    PROG_ALIVE(p);
    if (l < 0 || b < 0 || r > p->renW || t > p->renH || l >= r || b >= t)
        return wapt_fail(WAPT_E_INVALID, "region outside the render resolution");
    p->l = l;
    p->r = r;
    p->b = b;
    p->t = t;
    return WAPT_OK;
}

static int progressive_render(wapt_progressive* p, float* fb, wapt_stats* stats);

extern "C" int wapt_progressive_abort(wapt_progressive* p) {
    if (!p) return wapt_fail(WAPT_E_INVALID, "null argument");
    p->abort_flag.store(1);
    return WAPT_OK;
}

extern "C" int wapt_progressive_render(wapt_progressive* p, float* fb, wapt_stats* stats) {
    return wapt_guarded([&] { return progressive_render(p, fb, stats); });
}

static int progressive_render(wapt_progressive* p, float* fb, wapt_stats* stats) {
    PROG_ALIVE(p);
    if (!fb) return wapt_fail(WAPT_E_INVALID, "null argument");
    wapt_scene* s = p->s;
    HIPCHK(hipSetDevice(s->device));
    p->abort_flag.store(0);
    RenderView view;
    wapt_camera cam = p->cam;
    cam.resI = p->renW;
    cam.resJ = p->renH;
    view.cam = to_device(cam);  // wa_camera::setResolution (fundamentals.cpp:256-263)
    view.l = p->l;
    view.r = p->r;
    view.b = p->b;
    view.t = p->t;
    view.abort = &p->abort_flag;
    wapt_render_params prm = p->prm;
    // every iteration draws its own block of sample indices, so iterations are independent estimates
    long long begin = (long long)p->it * prm.psamples;
    if (begin + prm.psamples > 0x7fffffffLL) return wapt_fail(WAPT_E_LIMIT, "sample index overflow");
    prm.sample_begin = (int32_t)begin;
    prm.sample_count = prm.psamples;
    HIPCHK(hipMemsetAsync(p->d_sum, 0, (size_t)p->renW * p->renH * 3 * sizeof(float), s->stream));
    int rc = render_accumulate_view(s, &prm, p->d_sum, stats, &view);
    if (rc) return rc;
    float itf = (float)p->it;
    dim3 blk(256), g1((unsigned)((p->r - p->l + 255) / 256), (unsigned)(p->t - p->b));
    hipLaunchKernelGGL(k_blend, g1, blk, 0, s->stream, p->d_avg, p->d_sum, p->renW, p->dispW, p->l, p->r, p->b, p->t,
                       itf / (itf + 1.0f), (1.0f / (itf + 1.0f)) / (float)prm.psamples);
    dim3 g2((unsigned)((p->dispW + 255) / 256), (unsigned)p->dispH);
    hipLaunchKernelGGL(k_replicate, g2, blk, 0, s->stream, p->d_fb, p->d_avg, p->dispW, p->dispH, p->rf, p->l, p->r, p->b, p->t);
    HIPCHK(hipMemcpyAsync(fb, p->d_fb, (size_t)p->dispW * p->dispH * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    return WAPT_OK;
}

// generateTiles (cpp:941-960): 2^level x 2^level tiles over renW x renH, the last row / column takes the remainder
extern "C" int wapt_generate_tiles(int32_t renW, int32_t renH, int32_t subdiv_level, wapt_tile* tiles, int32_t* n_out) {
    if (!tiles || !n_out || renW <= 0 || renH <= 0 || subdiv_level < 0 || subdiv_level > 12)
        return wapt_fail(WAPT_E_INVALID, "bad argument");
    int dim = 1 << subdiv_level;
    int tw = (int)((float)renW / (float)dim), th = (int)((float)renH / (float)dim);
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) {
            wapt_tile& T = tiles[j * dim + i];
            T.l = i * tw;
            T.r = (i + 1) * tw;
            T.b = j * th;
            T.t = (j + 1) * th;
            if (i == dim - 1) T.r = renW;
            if (j == dim - 1) T.t = renH;
            T.w = T.h = 0;  // the reference leaves w and h of generated tiles unset; they are zeroed here
        }
    *n_out = dim * dim;
    return WAPT_OK;
}

static int intersect_device(wapt_scene* s, const wapt_ray* d_rays, uint32_t n, wapt_hit* d_hits,
                            wapt_trace_counters* counters, float* ms) {
    if (!s || (n && (!d_rays || !d_hits))) return wapt_fail(WAPT_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemsetAsync(s->d_rays, 0, 16 * sizeof(unsigned long long), s->stream));
    hipEvent_t a = nullptr, b = nullptr;
    if (ms) {
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        HIPCHK(hipEventRecord(a, s->stream));
    }
    {
        const uint32_t batch = 1u << 30;  // one launch unless the batch is enormous
        int rc = ensure_spill(s);
        if (rc) return rc;
        for (uint64_t off = 0; off < n; off += batch) {
            uint32_t m = (uint32_t)std::min<uint64_t>(batch, n - off);
            hipLaunchKernelGGL(k_reset, dim3(1), dim3(96), 0, s->stream, s->d_cnt, -1, 0u);
            const float4* r = (const float4*)d_rays + 2 * off;
            float* h = (float*)d_hits + 6 * off;
            if (counters)
                hipLaunchKernelGGL(k_intersect<true>, dim3(s->grid), dim3(WAPT_BLOCK), 0, s->stream, s->ks.tm, r, m, h, s->d_rays, s->d_cnt + CNT_FEED);
            else
                hipLaunchKernelGGL(k_intersect<false>, dim3(s->grid), dim3(WAPT_BLOCK), 0, s->stream, s->ks.tm, r, m, h, s->d_rays, s->d_cnt + CNT_FEED);
        }
    }
    if (ms) HIPCHK(hipEventRecord(b, s->stream));
    unsigned long long h[8] = {0};
    HIPCHK(hipMemcpyAsync(h, s->d_rays, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    if (ms) {
        (void)hipEventElapsedTime(ms, a, b);
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
    if (counters) {
        counters->rays = n;
        counters->node_visits = h[0];
        counters->tri_tests = h[1];
        counters->stack_spills = h[2];
        counters->wave_steps = h[4];
        counters->node_steps = h[5];
        counters->tri_trips = h[6];
    }
    if (h[3]) return wapt_fail(WAPT_E_LIMIT, "traversal stack overflow");
    return WAPT_OK;
}

extern "C" int wapt_intersect_device(wapt_scene* s, const wapt_ray* d_rays, uint32_t n, wapt_hit* d_hits,
                                     wapt_trace_counters* counters, float* ms) {
    return wapt_guarded([&] { return intersect_device(s, d_rays, n, d_hits, counters, ms); });
}

static int intersect_host(wapt_scene* s, const wapt_ray* rays, uint32_t n, wapt_hit* hits);

extern "C" int wapt_intersect(wapt_scene* s, const wapt_ray* rays, uint32_t n, wapt_hit* hits) {
    return wapt_guarded([&] { return intersect_host(s, rays, n, hits); });
}

static int intersect_host(wapt_scene* s, const wapt_ray* rays, uint32_t n, wapt_hit* hits) {
    if (!s || (n && (!rays || !hits))) return wapt_fail(WAPT_E_INVALID, "null argument");
    if (n == 0) return WAPT_OK;
    HIPCHK(hipSetDevice(s->device));
    wapt_ray* d_r = nullptr;
    wapt_hit* d_h = nullptr;
    HIPCHK(hipMalloc((void**)&d_r, (size_t)n * sizeof(wapt_ray)));
    hipError_t e = hipMalloc((void**)&d_h, (size_t)n * sizeof(wapt_hit));
    if (e != hipSuccess) {
        (void)hipFree(d_r);
        return wapt_fail(WAPT_E_HIP, hipGetErrorString(e));
    }
    int rc = WAPT_OK;
    e = hipMemcpy(d_r, rays, (size_t)n * sizeof(wapt_ray), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = wapt_fail(WAPT_E_HIP, hipGetErrorString(e));
    if (rc == WAPT_OK) rc = intersect_device(s, d_r, n, d_h, nullptr, nullptr);
    if (rc == WAPT_OK) {
        e = hipMemcpy(hits, d_h, (size_t)n * sizeof(wapt_hit), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = wapt_fail(WAPT_E_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_r);
    (void)hipFree(d_h);
    return rc;
}

// The closest-hit batch of wapt_intersect_device for callers that cannot allocate: everything the launch needs
// beyond the caller's rays and hits lives in this GPU's copy of `scratch`.  No hipMalloc / hipFree here or below.
static int intersect_device_static(wapt_scene* s, const wapt_ray* d_rays, uint32_t n, wapt_hit* d_hits) {
    if (!s || (n && (!d_rays || !d_hits))) return wapt_fail(WAPT_E_INVALID, "null argument");
    if (n > WAPT_STATIC_MAX_RAYS)
        return wapt_fail(WAPT_E_LIMIT, "wapt_intersect_device_static takes at most 2^30 rays per call (got " + std::to_string(n) + ")");
    if (n == 0) return WAPT_OK;
    HIPCHK(hipSetDevice(s->device));
    unsigned char* const base = s->static_scratch;
    if (!base) return wapt_fail(WAPT_E_HIP, "the static scratch array of this device was not found at scene creation");
    // the stack spill area is cut to the lanes that fit: a deeper tree runs with fewer blocks
    int depth_slots = std::max(1, (int)s->bvh_depth + 1 - WAPT_LDS_STACK);
    depth_slots = std::min(depth_slots, WAPT_STACK_CAP - WAPT_LDS_STACK);
    const size_t spill_bytes = WAPT_STATIC_SCRATCH_BYTES - WAPT_STATIC_HEADER_BYTES;
    const uint32_t blocks = (uint32_t)std::min<size_t>(s->grid, spill_bytes / ((size_t)depth_slots * sizeof(uint2) * WAPT_BLOCK));
    // (a launch with fewer blocks than work-feed shards would leave the groups of the unmapped shards undrawn)
    static_assert((WAPT_STATIC_SCRATCH_BYTES - WAPT_STATIC_HEADER_BYTES) / ((size_t)(WAPT_STACK_CAP - WAPT_LDS_STACK) * sizeof(uint2) * WAPT_BLOCK) >=
                      WAPT_FEED_SHARDS,
                  "the static scratch array must hold the stack spill area of at least one block per work-feed shard");
    if (blocks < WAPT_FEED_SHARDS) return wapt_fail(WAPT_E_LIMIT, "BVH too deep for the static scratch array");
    TraceMem tm = s->ks.tm;
    tm.spill = (uint2*)(base + WAPT_STATIC_HEADER_BYTES);
    tm.spill_stride = blocks * WAPT_BLOCK;
    tm.stack_cap = WAPT_LDS_STACK + depth_slots;
    // the array keeps what the last call left: feed counters and result counters start at zero
    HIPCHK(hipMemsetAsync(base, 0, WAPT_STATIC_HEADER_BYTES, s->stream));
    unsigned long long* const counters = (unsigned long long*)(base + (8u << 10));
    hipLaunchKernelGGL(k_intersect<false>, dim3(blocks), dim3(WAPT_BLOCK), 0, s->stream, tm, (const float4*)d_rays, n, (float*)d_hits,
                       counters, (uint32_t*)base);
    unsigned long long* const h = s->h_u64;  // pinned, allocated with the scene
    HIPCHK(hipMemcpyAsync(h, counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    if (h[3]) return wapt_fail(WAPT_E_LIMIT, "traversal stack overflow");
    return WAPT_OK;
}

extern "C" int wapt_intersect_device_static(wapt_scene* s, const wapt_ray* d_rays, uint32_t n, wapt_hit* d_hits) {
    return wapt_guarded([&] { return intersect_device_static(s, d_rays, n, d_hits); });
}
