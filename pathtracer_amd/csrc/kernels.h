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
//   S  A = from.xyz | pixel    B = dir.xyz | target   C = weight.rgb | length
//   P  (origin in VO)          B = dir.xyz | info     C = weight.rgb | medium
//
// R.info  = depth | includeEmission<<8 | rayType<<9 | (medium+1)<<12   (medium -1 = global); WAPT_INFO_NULL: no task
// P.info  = light | kind<<16   (kind 0 = directLightHit, 1 = directLightingSingleScatter)
//
// What a march sample of volumeScattering() (cpp:429-621) emits goes to slots that follow from the sample's number
// in its launch - no compaction, no atomics, no barriers: its origin and pixel once (VO[jj]), the bodies of its shadow
// walks (S.B, S.C, light by light: slot = vol_s1 + (light sample) x vol_ms + jj), of its probes (P.B, P.C, the same
// way) and its child (R[d+1], slot = first free + number of the sample in the step).  Whatever a sample does not emit
// is written as an empty entry (length 0, zero direction, WAPT_INFO_NULL), which the kernel that reads it skips.
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
    CNT_QM = 6 * WAPT_CNT_STRIDE,         // march samples of the current step in the open ((segment, sample) pairs)
    CNT_QMI = 7 * WAPT_CNT_STRIDE,        // march samples of the current step inside an object whose inside scatters
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

#define WAPT_INFO_NULL 0xffffffffu  // info word of an R slot that holds no task (a march sample without a child)

struct KQueues {
    float4 *RA, *RB, *RC;    // radiance tasks, depth d at [d*capR, (d+1)*capR)
    float4 *SA, *SB, *SC;
    float4 *PB, *PC;
    float4* VO;              // origin | pixel of the march samples of the current launch of the march kernel; where the
                             // samples have children it points into RA: the record is plane A of the sample's child
    uint2* hits;             // (t bits, triangle) per task of the current step
    // the marched segments of the current step (scenes with media), written by the surface kernel:
    // A.xyz | length, unit direction | pV, beta.rgb | key, pixel | medium+1, doVol<<30, inside<<31 | march samples | stream
    float4 *G0, *G1, *G2, *G3;
    uint2* QM;               // their march samples: (segment, sample number); in the open from the front, inside objects
                             // from the back of the list
    uint32_t capQM;
    // the march samples whose shadow walks / probes the current launch of k_shadow / k_probe reads: entry i >= vol_s1
    // of the shadow queue has its origin in VO[(i - vol_s1) % vol_ms], every entry i of the probe queue in VO[i % vol_ms]
    // (the division as multiply-and-shift, udiv_magic)
    uint32_t vol_s1, vol_ms, vol_div_m, vol_div_k;
    uint32_t vol_scount, vol_pcount;  // entries of the shadow queue beyond vol_s1; entries of the probe queue
    uint32_t capR, capS, capP;
    uint32_t capSA;          // entries of SA (origins of the walks that the surface kernel sends; the first entries of S)
    uint32_t capG;           // entries of G0-G3 and of the hit records: tasks of a step
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
