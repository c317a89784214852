// 8-wide quantised BVH: the acceleration structure that stands in for the
// Embree scene of the reference (rtcCommitScene, src/raytracer.cpp:172-175).
//
// Node = 80 bytes = five 16-byte loads per lane.  The eight child boxes are
// stored as 8-bit offsets on a per-node grid (origin p, power-of-two cell size
// 2^e per axis), children that are inner nodes sit contiguously from
// child_base (in slot order), the triangles of all leaf children sit
// contiguously from tri_base (at most 3 per child, 24 per node).
//
// meta[s] of slot s:
//   empty      : 0
//   inner node : (1<<5) | (24+s)
//   leaf       : (unary triangle count: 1->001 2->011 3->111)<<5 | offset from tri_base
// so "child_bits << (meta & 31)" drops a hit child's bits into a 32-bit hit
// mask whose top byte indexes inner children and whose low 24 bits index
// triangles.  Slots are assigned at build time so that visiting inner hits in
// descending (slot ^ octant_inverse) order is front to back.
//
// Triangle = 48 bytes = three 16-byte loads: the three vertices exactly as
// given (the watertight edge test needs bit-identical copies of shared
// vertices) and the (geomID, primID) pair of the reference's hit record
// (src/content.cpp:110-111).
#pragma once
#include <stdint.h>
#include <vector>

// Bytes from one node record to the next: 80 (packed) or 128 (one record per 128-byte line of L2 / HBM: a packed
// record straddles two lines in five cases of eight, which costs nothing while the tree sits in cache and a second
// line fetch per visit where it does not).
#ifndef WAPT_NODE_BYTES
#define WAPT_NODE_BYTES 80
#endif

struct Bvh8Node {
    float p[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t child_base;
    uint32_t tri_base;
    uint8_t meta[8];
    uint8_t qlo_x[8], qlo_y[8];
    uint8_t qlo_z[8], qhi_x[8];
    uint8_t qhi_y[8], qhi_z[8];
#if WAPT_NODE_BYTES > 80
    uint8_t pad[WAPT_NODE_BYTES - 80];
#endif
};
static_assert(sizeof(Bvh8Node) == WAPT_NODE_BYTES && WAPT_NODE_BYTES % 16 == 0, "node record size");

struct Bvh8Tri {
    float v0[3];
    float v1[3];
    float v2[3];
    int32_t obj;
    int32_t prim;
    uint32_t pad;
};
static_assert(sizeof(Bvh8Tri) == 48, "triangle must be 48 bytes");

struct Bvh8 {
    std::vector<Bvh8Node> nodes;
    std::vector<Bvh8Tri> tris;
    uint32_t max_depth = 0;  // levels of inner nodes on the longest root-to-leaf path
    float sah_cost = 0.f;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
};

// Builds the BVH over n triangles (verts: 9 floats each; obj/prim: ids copied
// into the leaf records).  Returns false and fills err for unusable input.
bool bvh8_build(const float* verts, const int32_t* obj, const int32_t* prim, uint64_t n, Bvh8* out,
                const char** err);
