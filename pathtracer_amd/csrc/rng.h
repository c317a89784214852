// Counter-based random numbers for the radiance loop.
//
// Replaces the process-global std::default_random_engine of the reference
// (src/randomnumbers.cpp:11, 30-41), which four threads share without a lock.
// A number is a pure function of (node key, pixel, dimension): any kernel, any
// GPU and the CPU checker can regenerate it, so a path-tree node may be
// processed in any order and the image does not depend on the GPU count.
//
// key   : 32-bit id of a node of the path tree.  The root key is a bijection of
//         the global sample index (for a fixed seed), so no two samples of a
//         pixel share a stream; a child's key hashes its parent's key with the
//         child's slot.
// pixel : j*W+i, mixed in so that the same key in another pixel is unrelated.
// dim   : (category << 20) | index, which draw of the node this is.
//
// The mixers are the 32-bit MurmurHash3 block mix and finaliser.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WAPT_HD __host__ __device__ inline
#else
#define WAPT_HD inline
#endif

WAPT_HD uint32_t wapt_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

WAPT_HD uint32_t wapt_fmix(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

WAPT_HD uint32_t wapt_mix(uint32_t h, uint32_t w) {
    w *= 0xcc9e2d51u;
    w = wapt_rotl(w, 15);
    w *= 0x1b873593u;
    h ^= w;
    h = wapt_rotl(h, 13);
    return h * 5u + 0xe6546b64u;
}

// Root key of global sample index `sample`.
WAPT_HD uint32_t wapt_key_root(uint64_t seed, uint32_t sample) {
    uint32_t h = wapt_mix((uint32_t)seed, (uint32_t)(seed >> 32));
    return wapt_fmix(wapt_mix(h, sample));
}

// Key of the child in slot `slot` of the node `key`.
WAPT_HD uint32_t wapt_key_child(uint32_t key, uint32_t slot) {
    return wapt_fmix(wapt_mix(key, slot) ^ 0x9e3779b9u);
}

// Per-node prefix: hash of key and pixel, computed once per node.
WAPT_HD uint32_t wapt_stream(uint32_t key, uint32_t pixel) { return wapt_mix(key, pixel); }

// Uniform float in [0,1) with 24 random bits.
WAPT_HD float wapt_rand(uint32_t stream, uint32_t dim) {
    uint32_t h = wapt_fmix(wapt_mix(stream, dim));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// Dimension categories of one radiance() node.
enum {
    WAPT_DIM_FIXED = 0u << 20,     // 0,1 camera jitter; 2 surface roulette; 3 volume roulette; 4 reflect/transmit
    WAPT_DIM_HEMI = 1u << 20,      // 2*i+{0,1}               indirectReflection sample i
    WAPT_DIM_NEE = 2u << 20,       // 2*(l*ls+i)+{0,1}        directReflection light l sample i
    WAPT_DIM_MARCH = 3u << 20,     // s                       march distance of step s
    WAPT_DIM_VNEE = 4u << 20,      // 2*((s*L+l)*ls+i)+{0,1}  light sample at march step s
    WAPT_DIM_VPROBE = 5u << 20,    // 2*(s*L+l)+{0,1}         directLightHit direction
    WAPT_DIM_VSINGLE = 6u << 20,   // 2*((s*L+l)*16+i)+{0,1}  directLightingSingleScatter direction
    WAPT_DIM_VCHILD = 7u << 20     // 2*s+{0,1}               indirect direction at march step s
};
enum { WAPT_DIM_CAM0 = 0, WAPT_DIM_CAM1 = 1, WAPT_DIM_SURF = 2, WAPT_DIM_VOL = 3, WAPT_DIM_DIEL = 4 };

// Child slots of one radiance() node.
enum {
    WAPT_SLOT_HEMI = 1u << 24,     // | i
    WAPT_SLOT_SPEC = 2u << 24,
    WAPT_SLOT_TRANS = 3u << 24,
    WAPT_SLOT_VOL = 4u << 24       // | s
};
