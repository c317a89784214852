// Device-side builder of the 8-wide BVH (bvh_device.hip): same node and triangle layout as the host
// builder (bvh8.h), result left in device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

struct DeviceBvh8 {
    void* d_nodes = nullptr;  // Bvh8Node[n_nodes], owned by the caller after a successful build (hipFree)
    void* d_tris = nullptr;   // Bvh8Tri[n_tris]
    uint64_t n_nodes = 0, n_tris = 0;
    uint32_t max_depth = 0;
    float sah_cost = 0.f;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double seconds = 0.0;
};

// verts: 9 floats per triangle (host); obj / prim: ids per triangle (host; prim may be null = running
// index); obj_opaque[o] = flag byte of object o (goes into the low byte of the triangle record's last word, the object index above it).
// The current device must be the one the scene lives on.  Returns false and fills err on failure.
bool bvh8_build_device(const float* verts, const int32_t* obj, const int32_t* prim, uint64_t n, const uint8_t* obj_opaque,
                       uint32_t n_objects, hipStream_t st, DeviceBvh8* out, std::string* err);
