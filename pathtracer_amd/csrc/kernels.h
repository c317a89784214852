// Wavefront form of udp_renderer::radiance (src/unidirectionalPathTracer.cpp:198-293)
// for gfx950.
//
// The reference evaluates a pixel sample as a recursion tree: every node is
// one radiance() call, and a node's value enters its parent's through a
// product of known factors.  Here every node is a task carrying that product
// (beta) and its pixel; a task adds beta * (local term) to the pixel with a
// float atomic and emits its children as new tasks.  Tasks of depth d live in
// queue R[d]; shadow walks (visibility(), cpp:693-790) and the two probe-ray
// estimators (directLightHit cpp:670-687, directLightingSingleScatter
// cpp:631-665) have queues of their own.  All queues are three planes of
// 16-byte words so that a wave reads and writes whole 1 KiB lines.
//
//   R  A = org.xyz | pixel     B = dir.xyz | info     C = beta.rgb | key
//   S  A = from.xyz | pixel    B = to.xyz  | -        C = weight.rgb | -
//   P  A = org.xyz | pixel     B = dir.xyz | info     C = weight.rgb | medium
//
// R.info  = depth | includeEmission<<8 | rayType<<9 | (medium+1)<<12   (medium -1 = global)
// P.info  = light | kind<<16   (kind 0 = directLightHit, 1 = directLightingSingleScatter)
#pragma once
#include "traverse.h"

// Counter words.  Every counter that kernels add to sits on a 256-byte line of its own: atomics on words of
// one line queue behind each other like atomics on one word, and the shading kernels add to two or three
// queue counters per step.
#define WAPT_CNT_STRIDE 64
enum {
    CNT_OVERFLOW = 1 * WAPT_CNT_STRIDE,   // traversal-stack pushes dropped (must stay 0)
    CNT_QOVERFLOW = 2 * WAPT_CNT_STRIDE,  // queue pushes dropped (must stay 0)
    CNT_QV = 3 * WAPT_CNT_STRIDE,         // segments of the current step that scatter (indices into the step)
    CNT_S = 4 * WAPT_CNT_STRIDE,          // shadow-walk tasks
    CNT_P = 5 * WAPT_CNT_STRIDE,          // probe tasks
    CNT_QM = 6 * WAPT_CNT_STRIDE,         // march samples of the current step ((task, sample) pairs)
    CNT_R0 = 8 * WAPT_CNT_STRIDE,         // CNT_R(d): tasks in R[d], d = 0..63
    CNT_FEED = (8 + 64) * WAPT_CNT_STRIDE, // 3 x 32 work-feed counters of k_trace, k_shadow and k_probe (zeroed by k_reset)
    CNT_TOTAL = CNT_FEED + 3 * 32 * 64
};
#define CNT_R(d) (CNT_R0 + (d) * WAPT_CNT_STRIDE)

enum { RAY_CAMERA = 0, RAY_DIFFUSE = 1, RAY_SPECULAR = 2 };  // unidirectionalPathTracer.h:9-11

struct KScene {
    TraceMem tm;
    const DMaterial* mats;   // [n_objects + 1], the last one is the global medium
    const DLight* lights;
    int32_t n_lights, n_objects;
    int32_t has_inside;      // some object's inside scatters (the march kernel's 16-probe branch can be needed)
    int32_t fast_probe;      // probe rays may be aimed at their light's quad (k_probe); 0: plain closest-hit searches
    int32_t fast_shadow;     // shadow walks may be finished from registers (k_shadow); 0: always the general path
    DCamera cam;
    float wc[3], wr;         // world bound sphere
};

struct KQueues {
    float4 *RA, *RB, *RC;    // radiance tasks, depth d at [d*capR, (d+1)*capR)
    float4 *SA, *SB, *SC;
    float4 *PA, *PB, *PC;
    uint2* hits;             // (t bits, triangle) per task of the current step
    uint32_t* QV;            // tasks of the current step whose segment is marched (scenes with media)
    uint2* QM;               // their march samples: (task of the step, sample number)
    uint32_t capQM;
    uint32_t capR, capS, capP;
    uint32_t* cnt;           // CNT_*
    unsigned long long* rays;  // closest-hit queries issued by shadow and probe kernels
    float* sum;              // W*H*3 radiance sums
    uint32_t debug_pixel;    // only read by builds with -DWAPT_DEBUG_PIXEL (per-pixel printf trace)
};

struct KParams {
    int32_t hsamples, lsamples, td;
    float me, mr;
    uint64_t seed;
    int32_t sample_begin;
    uint32_t slots_per_sample;  // tilesX*tilesY*64
    uint32_t tilesX;
    int32_t dense_raygen;       // W and H are multiples of 8: every slot is a pixel sample
    uint32_t sample_count;      // samples of this call
    int32_t rl, rr, rb, rt;     // pixel rectangle [rl, rr) x [rb, rt) that is rendered (setRegion, cpp:962-972)
    int32_t write_hits;         // the scene has media: keep refined hits and the list of marched segments for k_volume
    int32_t pixel_major;        // task order: all samples of a pixel next to each other (the lanes of a wave then work on
                                // one pixel: their camera rays, hit triangles and materials coincide, and the pixel's sum
                                // takes one atomic request per wave, see wave_add_rgb); else sample by sample
    // camera tasks made inside the depth-0 kernels.  Sample by sample: slot x of the step's chunk is sample
    // cam_s0 + x / slots_per_sample, image slot x % slots_per_sample.  Pixel by pixel: image slot cam_s0 + x / sample_count,
    // sample x % sample_count.  The divisions by launch constants as multiply-and-shift (udiv_magic).
    uint32_t cam_s0;
    uint32_t div_slots_m, div_slots_k, div_tiles_m, div_tiles_k;
};

// x / d for x < 2^31 as (x * m) >> k, with m = ceil(2^k / d), k = 31 + ceil(log2 d): the error term x * (m d - 2^k)
// stays below 2^k for such x, so the quotient is exact.
inline void udiv_magic(uint32_t d, uint32_t* m, uint32_t* k) {
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    *k = 31u + l;
    *m = (uint32_t)(((1ull << *k) + d - 1u) / d);
}
__device__ __forceinline__ uint32_t udiv_by(uint32_t x, uint32_t m, uint32_t k) {
    return (uint32_t)(((unsigned long long)x * (unsigned long long)m) >> k);
}
