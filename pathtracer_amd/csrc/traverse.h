// Closest-hit traversal of the 8-wide quantised BVH (bvh8.h) for gfx950:
// one ray per lane, the traversal stack split between an LDS short stack and
// a spill area in global memory.
//
// This is the device-side stand-in for rtcIntersect1 as wa_content::traceFAST
// uses it (src/content.cpp:91-123): closest hit with t in (tmin, tmax],
// two-sided triangles, no culling.
//
// Stack entries are 64-bit "node groups" (first child index, pending inner
// hits in the top byte + inner-child mask in the low byte), so the stack never
// holds more than one entry per tree level; the host refuses trees deeper than
// WAPT_STACK_CAP.  Entries [0, WAPT_LDS_STACK) of a lane live in LDS, laid out
// [entry][lane] so that a wave's 64-bit accesses hit distinct banks; deeper
// entries go to spill[(entry-WAPT_LDS_STACK)*stride + global lane].
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"

#define WAPT_BLOCK 256
#define WAPT_LDS_STACK 8
#define WAPT_STACK_CAP 32

struct TraceMem {
    const uint4* __restrict__ nodes;   // 5 x uint4 per node
    const float4* __restrict__ tris;   // 3 x float4 per triangle
    uint2* spill;
    uint32_t spill_stride;             // lanes of the launch (grid*block)
};

struct TraceCount {
    uint32_t nodes, tris, spills, overflow;
};

__device__ __forceinline__ float ubyte_f(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xffu); }

// Hit bits of four children (one meta word, six box words).
__device__ __forceinline__ uint32_t hit4(uint32_t nlx, uint32_t nly, uint32_t nlz, uint32_t nhx, uint32_t nhy,
                                         uint32_t nhz, uint32_t meta4, float ax, float ay, float az, float ox,
                                         float oy, float oz, float tmin, float tmax, uint32_t octinv4) {
    uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
    uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
    uint32_t bit_index4 = (meta4 ^ (octinv4 & inner_mask4)) & 0x1f1f1f1fu;
    uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float lx = fmaf(ubyte_f(nlx, j), ax, ox);
        float ly = fmaf(ubyte_f(nly, j), ay, oy);
        float lz = fmaf(ubyte_f(nlz, j), az, oz);
        float hx = fmaf(ubyte_f(nhx, j), ax, ox);
        float hy = fmaf(ubyte_f(nhy, j), ay, oy);
        float hz = fmaf(ubyte_f(nhz, j), az, oz);
        float tn = fmaxf(fmaxf(lx, ly), fmaxf(lz, tmin));
        float tf = fminf(fminf(hx, hy), fminf(hz, tmax));
        uint32_t bits = (child_bits4 >> (8 * j)) & 0xffu;
        uint32_t idx = (bit_index4 >> (8 * j)) & 0xffu;
        if (tn <= tf) mask |= bits << idx;
    }
    return mask;
}

// Point on a ray, rounded like the scalar expression org + t*dir without a fused
// multiply-add (content.cpp:98-100 as a non-FMA build evaluates it).  Hit points seed the next
// ray; at grazing angles one ulp decides whether that ray re-hits the surface it starts on
// beyond tnear = 0.001, so the rounding is kept the same as in the CPU checker.
__device__ __forceinline__ V3 point_at(V3 o, float t, V3 d) {
#pragma clang fp contract(off)
    float mx = t * d.x, my = t * d.y, mz = t * d.z;
    return v3(o.x + mx, o.y + my, o.z + mz);
}

// Unnormalised geometric normal (v1-v0)x(v2-v0) and ids of triangle `ti`.
__device__ __forceinline__ V3 tri_normal(const float4* __restrict__ tris, uint32_t ti, int* obj, int* prim) {
    const float4* tp = tris + (size_t)ti * 3;
    float4 a = tp[0], b = tp[1], c = tp[2];
    V3 p0 = v3(a.x, a.y, a.z), p1 = v3(a.w, b.x, b.y), p2 = v3(b.z, b.w, c.x);
    *obj = (int)__float_as_uint(c.y);
    *prim = (int)__float_as_uint(c.z);
    return cross(p1 - p0, p2 - p0);
}

// Traces one ray.  On return tmax is the hit distance (unchanged on a miss)
// and the result is the index of the hit triangle in the BVH's triangle array
// or 0xffffffff.  `lds` points at this lane's column of the block's short
// stack (element stride WAPT_BLOCK), `glane` is the lane's index in the launch.
template <bool COUNT>
__device__ __forceinline__ uint32_t trace_ray(const TraceMem& m, V3 o, V3 d, float tmin, float& tmax, uint2* lds,
                                              uint32_t glane, TraceCount& cnt) {
    const float tiny = 1e-18f;
    float idx_ = 1.0f / (fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x));
    float idy_ = 1.0f / (fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y));
    float idz_ = 1.0f / (fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z));
    const bool nx = d.x < 0.f, ny = d.y < 0.f, nz = d.z < 0.f;
    const uint32_t octinv = (nx ? 0u : 4u) | (ny ? 0u : 2u) | (nz ? 0u : 1u);
    const uint32_t octinv4 = octinv * 0x01010101u;

    uint32_t hit = 0xffffffffu;
    uint2 G = make_uint2(0u, 0x80000000u);  // root: "child 0 of group 0 is pending"
    int sp = 0;

    for (;;) {
        uint2 T;
        if (G.y & 0xff000000u) {
            uint32_t hits_imask = G.y;
            uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
            G.y &= ~(1u << bit);
            if (G.y & 0xff000000u) {
                if (sp < WAPT_LDS_STACK) {
                    lds[sp * WAPT_BLOCK] = G;
                    sp++;
                } else if (sp < WAPT_STACK_CAP) {
                    m.spill[(size_t)(sp - WAPT_LDS_STACK) * m.spill_stride + glane] = G;
                    sp++;
                    if (COUNT) cnt.spills++;
                } else {
                    cnt.overflow++;
                }
            }
            uint32_t slot = (bit - 24u) ^ octinv;
            uint32_t rel = __popc(hits_imask & 0xffu & ((1u << slot) - 1u));
            uint32_t ni = G.x + rel;
            const uint4* np = m.nodes + (size_t)ni * 5;
            uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
            if (COUNT) cnt.nodes++;

            float ax = __uint_as_float((n0.w & 0xffu) << 23) * idx_;
            float ay = __uint_as_float(((n0.w >> 8) & 0xffu) << 23) * idy_;
            float az = __uint_as_float(((n0.w >> 16) & 0xffu) << 23) * idz_;
            float ox = (__uint_as_float(n0.x) - o.x) * idx_;
            float oy = (__uint_as_float(n0.y) - o.y) * idy_;
            float oz = (__uint_as_float(n0.z) - o.z) * idz_;
            // near/far plane words per axis, by ray direction sign
            uint32_t lx0 = nx ? n3.z : n2.x, lx1 = nx ? n3.w : n2.y;
            uint32_t hx0 = nx ? n2.x : n3.z, hx1 = nx ? n2.y : n3.w;
            uint32_t ly0 = ny ? n4.x : n2.z, ly1 = ny ? n4.y : n2.w;
            uint32_t hy0 = ny ? n2.z : n4.x, hy1 = ny ? n2.w : n4.y;
            uint32_t lz0 = nz ? n4.z : n3.x, lz1 = nz ? n4.w : n3.y;
            uint32_t hz0 = nz ? n3.x : n4.z, hz1 = nz ? n3.y : n4.w;
            uint32_t hm = hit4(lx0, ly0, lz0, hx0, hy0, hz0, n1.z, ax, ay, az, ox, oy, oz, tmin, tmax, octinv4);
            hm |= hit4(lx1, ly1, lz1, hx1, hy1, hz1, n1.w, ax, ay, az, ox, oy, oz, tmin, tmax, octinv4);
            G = make_uint2(n1.x, (hm & 0xff000000u) | (n0.w >> 24));
            T = make_uint2(n1.y, hm & 0x00ffffffu);
        } else {
            T = G;
            G = make_uint2(0u, 0u);
        }

        while (T.y) {
            uint32_t k = (uint32_t)__ffs((int)T.y) - 1u;
            T.y &= T.y - 1u;
            uint32_t ti = T.x + k;
            const float4* tp = m.tris + (size_t)ti * 3;
            float4 a = tp[0], b = tp[1], c = tp[2];
            if (COUNT) cnt.tris++;
            // Ray/triangle test in the form Embree 3 uses for its default (non-robust) triangle
            // intersector (kernels/geometry/triangle_intersector_moeller.h, Moeller-Trumbore with
            // C = v0 - org, R = C x D, unnormalised edge tests against |den|): the reference's
            // closest hits come from that code (src/content.cpp:97), so its edge behaviour is the
            // one to follow.  Two-sided, interval (tmin, tmax].
            V3 p0 = v3(a.x, a.y, a.z), p1 = v3(a.w, b.x, b.y), p2 = v3(b.z, b.w, c.x);
            V3 e1 = p0 - p1, e2 = p2 - p0;
            V3 Ng = cross(e2, e1);
            V3 Cv = p0 - o;
            V3 R = cross(Cv, d);
            float den = dot(Ng, d);
            float absDen = fabsf(den);
            float sgn = den < 0.f ? -1.0f : 1.0f;
            float U = dot(R, e2) * sgn;
            float V = dot(R, e1) * sgn;
            float T = dot(Ng, Cv) * sgn;
            if (den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen && absDen * tmin < T && T <= absDen * tmax) {
                // The accept tests run in fp32; the distance that is kept is the same quotient
                // T/|den| evaluated in fp64 (a few dozen operations on accepted hits only), which
                // removes the 1e-5-level error fp32 shows on thin, distant triangles.
                double e1x = (double)p0.x - p1.x, e1y = (double)p0.y - p1.y, e1z = (double)p0.z - p1.z;
                double e2x = (double)p2.x - p0.x, e2y = (double)p2.y - p0.y, e2z = (double)p2.z - p0.z;
                double nx_ = e2y * e1z - e2z * e1y, ny_ = e2z * e1x - e2x * e1z, nz_ = e2x * e1y - e2y * e1x;
                double num = nx_ * ((double)p0.x - o.x) + ny_ * ((double)p0.y - o.y) + nz_ * ((double)p0.z - o.z);
                double dd = nx_ * d.x + ny_ * d.y + nz_ * d.z;
                float tr = (float)(num / dd);
                tmax = (tr > tmin) ? tr : T / absDen;
                hit = ti;
            }
        }

        if ((G.y & 0xff000000u) == 0u) {
            if (sp == 0) break;
            sp--;
            G = (sp < WAPT_LDS_STACK) ? lds[sp * WAPT_BLOCK]
                                      : m.spill[(size_t)(sp - WAPT_LDS_STACK) * m.spill_stride + glane];
        }
    }
    return hit;
}
