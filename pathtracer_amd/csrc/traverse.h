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

#include "bvh8.h"
#include "device_math.h"

#define WAPT_BLOCK 256
#ifndef WAPT_LDS_STACK
#define WAPT_LDS_STACK 8
#endif
#define WAPT_STACK_CAP 32
// waves per SIMD the register allocator must leave room for (second argument of __launch_bounds__).
// Measured on dragonChrome 1080p (ms per 64 spp, closest-hit / shadow kernel): 6 and 5 waves with 6 blocks
// per CU 18.0 / 14.7, 8 and 6 waves with 8 blocks 16.8 / 13.4, 8 and 7 waves with 8 blocks 16.9 / 12.6.
#ifndef WAPT_TRACE_WAVES
#define WAPT_TRACE_WAVES 8
#endif
#ifndef WAPT_SHADOW_WAVES
#define WAPT_SHADOW_WAVES 7
#endif
// the general shadow kernel (walks through media and transparent surfaces)
#ifndef WAPT_SHADOW_WAVES_GENERAL
#define WAPT_SHADOW_WAVES_GENERAL 7
#endif

// Last word of a triangle record: flags of its object in the low byte, the object index above.
#define WAPT_TRI_OPAQUE 1u   // the object is opaque: a shadow walk that meets it before its end point is over
#define WAPT_TRI_PLAIN 2u    // opaque, clear inside and no dielectric: reaching it AT the end point changes nothing
#define WAPT_TRI_GLASS 4u    // a dielectric with ior > 1: a shadow walk that ENTERS it (meets a front face) is over
                             // (src/unidirectionalPathTracer.cpp:757-759: glass casts opaque shadows)
#define WAPT_TRI_OBJ_SHIFT 8

struct TraceMem {
    const uint4* __restrict__ nodes;   // 5 x uint4 per node
    const float4* __restrict__ tris;   // 3 x float4 per triangle
    uint2* spill;
    uint32_t spill_stride;             // lanes of the largest launch
    int32_t stack_cap;                 // WAPT_LDS_STACK + spill entries per lane (>= depth of the tree)
    int32_t tri_quorum;                // lanes with pending triangles that trigger the triangle loop
    uint32_t n_nodes, n_tris;          // record counts (read by the index-checking build only)
};

// -DWAPT_CHECK_INDICES: every index that the traversal and the shadow / probe kernels compute is compared with its
// limit before it is used; a violation is printed and the index replaced by 0, so that the launch finishes and
// the message is seen instead of a memory fault.  A debugging build: the product build compiles none of it.
#ifdef WAPT_CHECK_INDICES
#define WAPT_CHECK_IDX(i, n, what)                                                                         \
    do {                                                                                                   \
        if (!((unsigned long long)(i) < (unsigned long long)(n))) {                                        \
            printf("wapt index check: %s = %llu, limit %llu (block %u thread %u)\n", what,                 \
                   (unsigned long long)(i), (unsigned long long)(n), (unsigned)blockIdx.x, (unsigned)threadIdx.x); \
            (i) = 0;                                                                                       \
        }                                                                                                  \
    } while (0)
#else
#define WAPT_CHECK_IDX(i, n, what) \
    do {                           \
    } while (0)
#endif

struct TraceCount {
    uint32_t nodes, tris, spills, overflow;
    // wave-level tallies of the counting build (equal in all lanes of a wave): loop iterations,
    // iterations whose node phase ran, trips of the triangle loop
    uint32_t wsteps, nsteps, ttrips;
};

__device__ __forceinline__ float ubyte_f(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xffu); }

// All ones if the sign bit of x is set, else zero.  (Written out, the compiler turns "sign spread over the word, then
// and / or" back into a compare and a select, which cost twice the issue cycles of the shift and the logic operation.)
__device__ __forceinline__ uint32_t sign_mask(float x) {
    uint32_t m;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(x));
    return m;
}

// Hit bits of four children (one meta word, six box words).
__device__ __forceinline__ uint32_t hit4(uint32_t nlx, uint32_t nly, uint32_t nlz, uint32_t nhx, uint32_t nhy,
                                         uint32_t nhz, uint32_t meta4, float ax, float ay, float az, float ox,
                                         float oy, float oz, float tmin, float tmax, uint32_t octinv4) {
    uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
    // bits 0-2 in the bytes of inner children (the octant has three bits).  Two shift-ors: written as one
    // expression the compiler turns them into a multiply by 7, which runs at quarter rate.
    uint32_t inner1 = is_inner4 >> 4;
    uint32_t inner_mask4 = inner1 | (inner1 << 1);
    asm volatile("" : "+v"(inner_mask4));
    inner_mask4 |= inner1 << 2;
    // On this chip a wave64 instruction occupies its SIMD for 2 cycles if it is an f32 add / mul / fma, a logic
    // operation of up to three inputs, an integer add, a shift RIGHT or a move, and for 4 if it is anything else
    // (compare, select, bit-field extract, shift left, min / max, conversion).  The kernels that run this are bound
    // by exactly those cycles, so the mask is put together from the cheap kinds: a shift count needs no masking (the
    // shifter reads five bits), and "hit" is the sign of tf - tn spread over the word instead of a compare and a
    // select (no NaN can arise: all terms are finite; a difference flushed to zero keeps its sign).
    uint32_t bit_index4 = meta4 ^ (octinv4 & inner_mask4);
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float lx = fmaf(ubyte_f(nlx, j), ax, ox);
        float ly = fmaf(ubyte_f(nly, j), ay, oy);
        float lz = fmaf(ubyte_f(nlz, j), az, oz);
        float hx = fmaf(ubyte_f(nhx, j), ax, ox);
        float hy = fmaf(ubyte_f(nhy, j), ay, oy);
        float hz = fmaf(ubyte_f(nhz, j), az, oz);
        // the box is met within [tmin, tmax] iff max(lx, ly, lz, tmin) <= min(hx, hy, hz, tmax), that is iff none of
        // tf - tn, tmax - tn, tf - tmin is negative
        float tn = fmaxf(fmaxf(lx, ly), lz);
        float tf = fminf(fminf(hx, hy), hz);
        uint32_t bits = (meta4 >> (8 * j + 5)) & 7u;
        uint32_t idx = (bit_index4 >> (8 * j)) & 31u;
        uint32_t miss = sign_mask(__uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(tf - tn), __float_as_uint(tmax - tn),
                                                                              __float_as_uint(tf - tmin), 0xfe)));  // (a | b | c)
        mask |= (bits << idx) & ~miss;
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

// State of one ray in flight.  A kernel keeps one per lane, calls trav_begin when the lane takes a
// new ray, trav_step until it reports completion, and trav_finish for the final distance.  Keeping
// the loop in the kernel lets a lane whose ray has ended pick up the next ray at once instead of
// idling until the slowest ray of its wave is done (the waves' rays differ a lot in length once
// paths have bounced).
struct Trav {
    V3 o, d;
    float idx, idy, idz;
    float tmin, tmax;
    uint32_t octinv;  // 7 ^ (bit 2: runs down x, bit 1: down y, bit 0: down z), the same in all four bytes
    uint32_t hit;   // triangle index or 0xffffffff
    uint2 G;        // current node group
    uint2 T;        // pending triangles: first index, hit bits
    int sp;
    float tblock;   // shadow walks: an OPAQUE surface hit closer than this ends the search at once;
                    // negative afterwards: the search was ended that way
    uint32_t hflag; // shadow walks: last word of the hit triangle's record (flags | object << 8)
};

__device__ __forceinline__ void trav_begin(Trav& t, V3 o, V3 d, float tmin, float tmax) {
    const float tiny = 1e-18f;
    t.o = o;
    t.d = d;
#ifndef WAPT_EXACT_DIV
    // 1-ulp reciprocals: the box test is not exact either way (quantised, conservative boxes)
    t.idx = __builtin_amdgcn_rcpf(fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x));
    t.idy = __builtin_amdgcn_rcpf(fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y));
    t.idz = __builtin_amdgcn_rcpf(fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z));
#else
    t.idx = 1.0f / (fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x));
    t.idy = 1.0f / (fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y));
    t.idz = 1.0f / (fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z));
#endif
    t.tmin = tmin;
    t.tmax = tmax;
    // (the sign BIT, as for the reciprocals above: a component of -0.0 runs down its axis like any other negative one)
    t.octinv = (7u ^ (((__float_as_uint(d.x) >> 31) << 2) | ((__float_as_uint(d.y) >> 31) << 1) | (__float_as_uint(d.z) >> 31))) * 0x01010101u;
    t.hit = 0xffffffffu;
    t.G = make_uint2(0u, 0x80000000u);  // root: "child 0 of group 0 is pending"
    t.T = make_uint2(0u, 0u);
    t.sp = 0;
    t.tblock = 0.f;
    t.hflag = 0u;
}

// Wave-private LDS of the cooperative triangle phase: the closest hit found so far per lane's ray
// ((distance bits << 32) | triangle, so a 64-bit minimum keeps the nearest) and the list of pending
// (lane, triangle-of-its-set) pairs of the step.
// 1: the triangle phase of the kernels is the cooperative one (tri_phase_coop); 0: the sequential loop
#ifndef WAPT_COOP
#define WAPT_COOP 1
#endif
#ifndef WAPT_COOP_PER_LANE
#define WAPT_COOP_PER_LANE 4
#endif
// 1: the one-search shadow kernel (scenes of opaque objects) uses the cooperative phase in its any-hit form.
// With tasks ordered sample by sample it cost the shadow kernel 3 %; ordered pixel by pixel (the lanes' walks start
// close together and meet the same leaves) it gains: Cornell box +3.5 %, fog's shadow kernel -7 %, dragonChrome's -2.6 %.
#ifndef WAPT_SHADOW_COOP
#define WAPT_SHADOW_COOP 1
#endif

struct CoopLds {
    unsigned long long best[64];
    uint16_t items[64 * WAPT_COOP_PER_LANE];
};

// Triangle phase with the tests spread over the wave: every pending (ray, triangle) pair of the step becomes
// one item, the items are dealt to the lanes 64 at a time, a lane fetches the ray of the item's owner across
// the wave and tests the triangle, and a 64-bit minimum per owner keeps the nearest hit.  (One ray's leaf with
// three triangles costs the sequential loop three trips with one lane busy; here the step's ~57 tests take one
// round.)  A lane contributes at most WAPT_COOP_PER_LANE triangles per step and keeps the rest pending.
// (Closest-hit searches only.  Shadow walks keep the sequential loop: they stop at the first opaque hit, keep 41 % of
// that loop's lanes busy and need the flag word of the hit; spread over the wave they were 14 % slower.)
// ANY (shadow walks of scenes whose objects are all opaque): a hit closer than the owner's t.tblock ends that ray's
// search whatever else lies on its way, so it is recorded with distance bits zero - nothing beats it in the minimum,
// every later test of the same ray fails against a far limit of zero - and the owner drops its stack.
template <bool COUNT, bool ANY = false>
__device__ __forceinline__ void tri_phase_coop(const TraceMem& m, Trav& t, bool pending, CoopLds* cl, TraceCount& cnt) {
    const uint32_t lane = __lane_id();
    const uint32_t bits = pending ? t.T.y : 0u;
    uint32_t rest = bits;
#pragma unroll
    for (int j = 0; j < WAPT_COOP_PER_LANE; j++) rest &= rest - 1u;
    const uint32_t mine = bits ^ rest;  // the lowest WAPT_COOP_PER_LANE pending triangles
    const uint32_t n_mine = (uint32_t)__popc(mine);
    // exclusive prefix sum of the (small) counts from one ballot per bit of the count: no LDS round trips
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int bit = 0; (1 << bit) <= WAPT_COOP_PER_LANE; bit++) {
        const unsigned long long mk = __ballot((n_mine >> bit) & 1u);
        off += (uint32_t)__popcll(mk & ((1ull << lane) - 1ull)) << bit;
        total += (uint32_t)__popcll(mk) << bit;
    }
    for (uint32_t b = mine; b; b &= b - 1u) cl->items[off++] = (uint16_t)(lane | (((uint32_t)__ffs((int)b) - 1u) << 6));
    cl->best[lane] = ((unsigned long long)__float_as_uint(t.tmax) << 32) | (unsigned long long)t.hit;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (COUNT) cnt.ttrips += (total + 63u) >> 6;
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t idx = base + lane;
        const bool valid = idx < total;
        const uint32_t item = valid ? (uint32_t)cl->items[idx] : 0u;
        const int src = (int)(item & 63u);
        const uint32_t k = item >> 6;
        const uint32_t tbase = (uint32_t)__shfl((int)t.T.x, src);
        const V3 o = v3(__shfl(t.o.x, src), __shfl(t.o.y, src), __shfl(t.o.z, src));
        const V3 d = v3(__shfl(t.d.x, src), __shfl(t.d.y, src), __shfl(t.d.z, src));
        const float tmin = __shfl(t.tmin, src);
        const float tblock = ANY ? __shfl(t.tblock, src) : 0.f;
        const float tmax = __uint_as_float((uint32_t)(cl->best[src] >> 32));
        if (valid) {
            uint32_t ti = tbase + k;
            WAPT_CHECK_IDX(ti, m.n_tris, "triangle (cooperative phase)");
            const float4* tp = m.tris + (uint32_t)(ti * 3u);
            float4 a = tp[0], b = tp[1], c = tp[2];
            if (COUNT) cnt.tris++;
            // the triangle test of the sequential loop below (Embree's form, see there), on the fetched ray
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
            float Tn = dot(Ng, Cv) * sgn;
            // ANY: only what ends a walk counts - opaque surfaces, and front faces (den < 0: Ng is the geometric normal)
            // of glass; back faces of glass and dielectrics with ior <= 1 are passed through as if they were not there
            // (the scenes that use this form have nothing but such objects: clear inside, no attenuation either side)
            bool counts = true;
            if (ANY) {
                const uint32_t fl = __float_as_uint(c.w);
                counts = (fl & WAPT_TRI_OPAQUE) != 0u || ((fl & WAPT_TRI_GLASS) != 0u && den < 0.f);
            }
            if (counts && den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen && absDen * tmin < Tn && Tn <= absDen * tmax) {
                // search distance only (the final one is re-evaluated in fp64, see refine_distance): the 1-ulp
                // reciprocal saves the ten-instruction IEEE division
                if (ANY && Tn < absDen * tblock) {
                    atomicMin(&cl->best[src], (unsigned long long)ti);
                } else {
                    const unsigned long long key = (unsigned long long)__float_as_uint(Tn * __builtin_amdgcn_rcpf(absDen)) << 32;
                    atomicMin(&cl->best[src], key | (unsigned long long)ti);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (pending) {
        unsigned long long r = cl->best[lane];
        t.tmax = __uint_as_float((uint32_t)(r >> 32));
        t.hit = (uint32_t)r;
        t.T.y = rest;
        if (ANY && (uint32_t)(r >> 32) == 0u && t.hit != 0xffffffffu) {
            // stopped at a surface before the end point: the search is over (no accepted distance is zero: tmin > 0)
            t.G = make_uint2(0u, 0u);
            t.T.y = 0u;
            t.sp = 0;
            t.tblock = -1.0f;
        }
    }
}

// One traversal step of a wave.  A lane without pending triangles visits its nearest pending inner
// child (popping the stack when its current group is exhausted); triangles that the visited nodes
// expose are collected in t.T and tested when at least `tri_quorum` lanes of the wave have some or no
// lane can descend any further.  (Measured on the dragon scene: postponing the triangle loop to gather
// more lanes loses more to waiting than it saves - closest-hit kernel 11.8 / 12.8 / 14.4 / 15.7 ms per
// frame at quorum 1 / 8 / 32 / 48 - so the default quorum is 1; the knob stays.)  Returns true when the lane's
// ray is finished.  Must be called by all lanes of the wave together (`have` = lane holds a ray).
// `lds` points at this lane's column of the block's short stack (element stride WAPT_BLOCK), `glane`
// is the lane's index in the launch.
// SHADOW: the caller only wants to know what a shadow walk needs, so the search stops at the first
// opaque triangle (bit 0 of the triangle record's last word) closer than t.tblock: whatever else lies
// in front of it, the walk from A to B ends with zero visibility.
template <bool COUNT, bool SHADOW = false, bool COOP = false>
__device__ __forceinline__ bool trav_step(const TraceMem& m, Trav& t, bool have, uint2* lds, uint32_t glane,
                                          TraceCount& cnt, CoopLds* coop = nullptr) {
    // (SHADOW && COOP: the any-hit form of the cooperative phase, for scenes whose objects are all opaque)
    const int tri_quorum = m.tri_quorum;
    bool descend = have && t.T.y == 0u;
    if (COUNT) {
        cnt.wsteps++;
        cnt.nsteps += __any(descend) ? 1u : 0u;
    }
    if (descend && (t.G.y & 0xff000000u) == 0u) {
        // current group exhausted: pop, or finish
        if (t.sp == 0) return true;
        t.sp--;
        WAPT_CHECK_IDX(t.sp, m.stack_cap, "stack pointer (pop)");
        WAPT_CHECK_IDX(glane, m.spill_stride > 0u ? m.spill_stride : 0xffffffffu, "lane of the launch");
        // (two separate loads: written as one conditional expression the compiler selects between the two ADDRESSES
        //  and issues a flat load, which is slower than an LDS read and is waited for together with every vector-memory
        //  operation the wave has in flight - the hit record it has just stored, for one)
        if (t.sp < WAPT_LDS_STACK) {
            t.G = lds[t.sp * WAPT_BLOCK];
            asm volatile("" : "+v"(t.G.x), "+v"(t.G.y));  // (keeps the two loads apart)
        } else {
            const uint2* sp_ = m.spill + ((size_t)(t.sp - WAPT_LDS_STACK) * m.spill_stride + glane);
            t.G = *sp_;
        }
    }
    if (descend) {
        // all ones where the ray runs down the axis (the sign of the reciprocal: it is never zero)
        const uint32_t nx = sign_mask(t.idx), ny = sign_mask(t.idy), nz = sign_mask(t.idz);
        uint32_t hits_imask = t.G.y;
        uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
        t.G.y &= ~(1u << bit);
        if (t.G.y & 0xff000000u) {
            WAPT_CHECK_IDX(t.sp, m.stack_cap + 1, "stack pointer (push)");
            if (t.sp < WAPT_LDS_STACK) {
                lds[t.sp * WAPT_BLOCK] = t.G;
                t.sp++;
            } else if (t.sp < m.stack_cap) {
                m.spill[(size_t)(t.sp - WAPT_LDS_STACK) * m.spill_stride + glane] = t.G;
                t.sp++;
                if (COUNT) cnt.spills++;
            } else {
                cnt.overflow++;
            }
        }
        uint32_t slot = (bit - 24u) ^ (t.octinv & 7u);
        uint32_t rel = __popc(hits_imask & 0xffu & ((1u << slot) - 1u));
        uint32_t ni = t.G.x + rel;
        WAPT_CHECK_IDX(ni, m.n_nodes, "node");
        const uint4* np = m.nodes + (uint32_t)(ni * (uint32_t)(WAPT_NODE_BYTES / 16));  // (32-bit index arithmetic: a 64-bit multiply runs at quarter rate)
        uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        if (COUNT) cnt.nodes++;

        float ax = __uint_as_float((n0.w & 0xffu) << 23) * t.idx;
        float ay = __uint_as_float(((n0.w >> 8) & 0xffu) << 23) * t.idy;
        float az = __uint_as_float(((n0.w >> 16) & 0xffu) << 23) * t.idz;
        float ox = (__uint_as_float(n0.x) - t.o.x) * t.idx;
        float oy = (__uint_as_float(n0.y) - t.o.y) * t.idy;
        float oz = (__uint_as_float(n0.z) - t.o.z) * t.idz;
        // near/far plane words per axis, by ray direction sign
#define WAPT_SEL(m, a, b) __builtin_amdgcn_bitop3_b32(a, b, m, 0xe4)  // m ? a : b bit by bit, one three-input logic instruction
        uint32_t lx0 = WAPT_SEL(nx, n3.z, n2.x), lx1 = WAPT_SEL(nx, n3.w, n2.y);
        uint32_t hx0 = WAPT_SEL(nx, n2.x, n3.z), hx1 = WAPT_SEL(nx, n2.y, n3.w);
        uint32_t ly0 = WAPT_SEL(ny, n4.x, n2.z), ly1 = WAPT_SEL(ny, n4.y, n2.w);
        uint32_t hy0 = WAPT_SEL(ny, n2.z, n4.x), hy1 = WAPT_SEL(ny, n2.w, n4.y);
        uint32_t lz0 = WAPT_SEL(nz, n4.z, n3.x), lz1 = WAPT_SEL(nz, n4.w, n3.y);
        uint32_t hz0 = WAPT_SEL(nz, n3.x, n4.z), hz1 = WAPT_SEL(nz, n3.y, n4.w);
#undef WAPT_SEL
        const uint32_t octinv4 = t.octinv;
        uint32_t hm = hit4(lx0, ly0, lz0, hx0, hy0, hz0, n1.z, ax, ay, az, ox, oy, oz, t.tmin, t.tmax, octinv4);
        hm |= hit4(lx1, ly1, lz1, hx1, hy1, hz1, n1.w, ax, ay, az, ox, oy, oz, t.tmin, t.tmax, octinv4);
        t.G = make_uint2(n1.x, (hm & 0xff000000u) | (n0.w >> 24));
        t.T = make_uint2(n1.y, hm & 0x00ffffffu);
    }

    // triangles: now, or with more company later
    bool pending = have && t.T.y != 0u;
    bool can_descend = have && t.T.y == 0u && ((t.G.y & 0xff000000u) != 0u || t.sp > 0);
    int waiting = (int)__popcll(__ballot(pending));
    if (COOP) {
        if (waiting > 0) tri_phase_coop<COUNT, SHADOW>(m, t, pending, coop, cnt);
    } else if (waiting >= tri_quorum || (waiting > 0 && !__any(can_descend))) {
        // (lanes without a ray, whose state is stale, stay out: `bits` is zero for them)
        uint32_t bits = pending ? t.T.y : 0u;
        if (COUNT) {
            uint32_t trips = (uint32_t)__popc(bits);
            for (int o = 32; o > 0; o >>= 1) trips = max(trips, (uint32_t)__shfl_xor((int)trips, o));
            cnt.ttrips += trips;
        }
        while (bits) {
            uint32_t k = (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            uint32_t ti = t.T.x + k;
            WAPT_CHECK_IDX(ti, m.n_tris, "triangle");
            const float4* tp = m.tris + (uint32_t)(ti * 3u);
            float4 a = tp[0], b = tp[1], c = tp[2];
            // (the flag word is only looked at for accepted hits: without this the compiler moves its load
            //  into that branch, where a whole memory round trip is then waited for with most lanes idle)
            if (SHADOW) asm volatile("" ::"v"(c.w));
            if (COUNT) cnt.tris++;
            // Ray/triangle test in the form Embree 3 uses for its default (non-robust) triangle
            // intersector (kernels/geometry/triangle_intersector_moeller.h, Moeller-Trumbore with
            // C = v0 - org, R = C x D, unnormalised edge tests against |den|): the reference's
            // closest hits come from that code (src/content.cpp:97), so its edge behaviour is the
            // one to follow.  Two-sided, interval (tmin, tmax].
            V3 p0 = v3(a.x, a.y, a.z), p1 = v3(a.w, b.x, b.y), p2 = v3(b.z, b.w, c.x);
            V3 e1 = p0 - p1, e2 = p2 - p0;
            V3 Ng = cross(e2, e1);
            V3 Cv = p0 - t.o;
            V3 R = cross(Cv, t.d);
            float den = dot(Ng, t.d);
            float absDen = fabsf(den);
            float sgn = den < 0.f ? -1.0f : 1.0f;
            float U = dot(R, e2) * sgn;
            float V = dot(R, e1) * sgn;
            float Tn = dot(Ng, Cv) * sgn;
            if (den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen && absDen * t.tmin < Tn && Tn <= absDen * t.tmax) {
#ifdef WAPT_EXACT_DIV
                t.tmax = Tn / absDen;
#else
                // search distance only (the final one is re-evaluated in fp64, see refine_distance): the
                // 1-ulp reciprocal saves the ten-instruction IEEE division in the hottest loop
                t.tmax = Tn * __builtin_amdgcn_rcpf(absDen);
#endif
                t.hit = ti;
                if (SHADOW) {
                    t.hflag = __float_as_uint(c.w);
                    if (((t.hflag & WAPT_TRI_OPAQUE) || ((t.hflag & WAPT_TRI_GLASS) && den < 0.f)) && Tn < absDen * t.tblock) {
                        t.G = make_uint2(0u, 0u);
                        t.sp = 0;
                        bits = 0u;
                        t.tblock = -1.0f;
                    }
                }
            }
        }
        if (pending) t.T.y = 0u;
    }
    // a lane is finished when nothing is pending anywhere
    return have && t.T.y == 0u && (t.G.y & 0xff000000u) == 0u && t.sp == 0;
}

// One triangle against one ray with the arithmetic of the search loops above (Embree's form): true and the
// search-precision distance if the ray hits it in (tmin, tmax].
__device__ __forceinline__ bool tri_hit_one(const float4* __restrict__ tris, uint32_t ti, V3 o, V3 d, float tmin, float tmax, float* th) {
    const float4* tp = tris + (uint32_t)(ti * 3u);
    float4 a = tp[0], b = tp[1], c = tp[2];
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
    float Tn = dot(Ng, Cv) * sgn;
    if (den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen && absDen * tmin < Tn && Tn <= absDen * tmax) {
        *th = Tn * __builtin_amdgcn_rcpf(absDen);
        return true;
    }
    return false;
}

// Where a ray meets the two triangles tri0, tri1 of a light's quad, if it does: the nearer hit.
__device__ __forceinline__ bool quad_hit(const float4* __restrict__ tris, uint32_t tri0, uint32_t tri1, V3 o, V3 d, float tmin,
                                         float tmax, float* th, uint32_t* tri) {
    float t0 = 0.f, t1 = 0.f;
    const bool h0 = tri_hit_one(tris, tri0, o, d, tmin, tmax, &t0);
    const bool h1 = tri_hit_one(tris, tri1, o, d, tmin, tmax, &t1);
    if (!h0 && !h1) return false;
    const bool first = h0 && (!h1 || t0 <= t1);
    *th = first ? t0 : t1;
    *tri = first ? tri0 : tri1;
    return true;
}

// Distance of a hit: the same quotient T/|den| as in the search, re-evaluated in fp64 from the float
// vertices (fp32 shows 1e-5-level relative error on thin, distant triangles; the acceptance bar for
// hit distances is 1e-5).  `t32` is the fp32 distance the search produced (returned if the fp64 value
// falls out of the interval by rounding).
__device__ __forceinline__ float refine_distance(const float4* __restrict__ tris, uint32_t tri, V3 o, V3 d, float tmin,
                                                 float t32) {
    const float4* tp = tris + (size_t)tri * 3;
    float4 a = tp[0], b = tp[1], c = tp[2];
    double e1x = (double)a.x - a.w, e1y = (double)a.y - b.x, e1z = (double)a.z - b.y;
    double e2x = (double)b.z - a.x, e2y = (double)b.w - a.y, e2z = (double)c.x - a.z;
    double nx_ = e2y * e1z - e2z * e1y, ny_ = e2z * e1x - e2x * e1z, nz_ = e2x * e1y - e2y * e1x;
    double num = nx_ * ((double)a.x - o.x) + ny_ * ((double)a.y - o.y) + nz_ * ((double)a.z - o.z);
    double dd = nx_ * d.x + ny_ * d.y + nz_ * d.z;
    float tr = (float)(num / dd);
    return (tr > tmin) ? tr : t32;
}

__device__ __forceinline__ float trav_finish(const TraceMem& m, const Trav& t) {
    if (t.hit == 0xffffffffu) return t.tmax;
    return refine_distance(m.tris, t.hit, t.o, t.d, t.tmin, t.tmax);
}

// Traces one ray to completion (kernels whose lanes do not stream rays).  On return tmax is the hit
// distance (unchanged on a miss) and the result is the index of the hit triangle or 0xffffffff.
template <bool COUNT>
__device__ __forceinline__ uint32_t trace_ray(const TraceMem& m, V3 o, V3 d, float tmin, float& tmax, uint2* lds,
                                              uint32_t glane, TraceCount& cnt) {
    // (lanes of a wave may arrive here with different trip counts: keep every lane in the loop until
    //  the whole wave is done, as trav_step votes across the wave)
    Trav t;
    trav_begin(t, o, d, tmin, tmax);
    bool have = true;
    while (__any(have)) {
        if (trav_step<COUNT>(m, t, have, lds, glane, cnt)) have = false;
    }
    tmax = trav_finish(m, t);
    return t.hit;
}

// Hands work items to the lanes of one wave of a persistent launch.  The queue is dealt out in groups
// of 2^gshift consecutive items; group g belongs to shard g % WAPT_FEED_SHARDS, every shard has a counter
// of its own, and a wave whose current group is used up draws the next group of the shard its block
// maps to (one returning atomic per group, spread over 32 counters).  The cost of a ray varies by
// more than 10x (wall vs. mesh): with a fixed share per wave the average wave was resident for 43 % of
// the kernel's duration, with groups drawn on demand the waves finish together.  Neighbouring groups
// are in flight at the same time, which keeps the BVH lines they touch in L1/L2.
// The wave waits for the atomic's answer, so the group grows with the launch (64 items below 2^23, up to
// 512 from 2^27 on: dragonChrome +2 %, the short shadow rays of the fog scene +20 %), and stays small
// where the tail of a small launch would otherwise be left to a few waves (dragonMilk: -5 % at 256 throughout).
// A lane that asks (`want`) gets the next unassigned item or 0xffffffff; take() must be called by all
// lanes of the wave together, and the counters must be zero when the kernel starts.
#define WAPT_FEED_SHARDS 32
#define WAPT_FEED_STRIDE 64
struct WaveFeed {
    uint32_t next, end;   // unassigned part of the current group
    uint32_t n, shard, gshift;
    uint32_t* counter;
    bool dry;
    __device__ __forceinline__ uint32_t take(bool want) {
        unsigned long long mk = __ballot(want);
        if (mk == 0ull) return 0xffffffffu;
        if (next >= end && !dry) {
            uint32_t k = 0;
            if (__lane_id() == 0) k = atomicAdd(counter, 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);  // (all lanes are here together: the first one is lane 0)
            unsigned long long b = ((unsigned long long)k * WAPT_FEED_SHARDS + shard) << gshift;
            if (b >= n) {
                dry = true;
            } else {
                next = (uint32_t)b;
                end = (uint32_t)min(b + (1ull << gshift), (unsigned long long)n);
            }
        }
        uint32_t lane = __lane_id();
        uint32_t idx = next + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
        uint32_t got = (want && idx < end) ? idx : 0xffffffffu;
        uint32_t c = (uint32_t)__popcll(mk);
        next = (end - next < c) ? end : next + c;
        return got;
    }
    // Nothing left to hand out, now or later (the same in all lanes of the wave).  A kernel whose lanes can finish
    // an item at the moment they take it must not leave its loop on "no lane holds an item" alone: the rest of
    // the wave's current group would be lost.
    __device__ __forceinline__ bool exhausted() const { return dry && next >= end; }
};

__device__ __forceinline__ WaveFeed wave_feed(uint32_t n, uint32_t* counters) {
    WaveFeed f;
    f.shard = blockIdx.x % WAPT_FEED_SHARDS;
    f.n = n;
    f.counter = counters + f.shard * WAPT_FEED_STRIDE;  // one 256-byte line per counter: atomics on words of
                                                        // one line queue behind each other like atomics on one word
    f.next = f.end = 0;
#ifndef WAPT_FEED_BIG_SHIFT
#define WAPT_FEED_BIG_SHIFT 9u
#endif
    f.gshift = n >= (1u << 27) ? WAPT_FEED_BIG_SHIFT : n >= (1u << 25) ? 8u : n >= (1u << 23) ? 7u : 6u;
    f.dry = n == 0;
    return f;
}
