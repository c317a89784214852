// Internals shared by the translation units of libwapt.so (not installed).
#pragma once
#include <string>
#include <vector>

#include "../../include/wapt.h"

// Records the message returned by wapt_last_error() and returns `code`.
int wapt_fail(int code, const std::string& msg);

// Parsed .waff file + loaded meshes, flattened the way wa_content::init
// (src/content.cpp:13-40) numbers objects.
struct wapt_waff {
    wapt_camera camera;
    wapt_material global_medium;
    std::vector<wapt_light> lights;
    std::vector<wapt_material> materials;     // lights first, then meshes
    std::vector<std::string> mesh_files;      // one per mesh object
    std::vector<float> tri_verts;
    std::vector<int32_t> tri_obj, tri_prim;
    int32_t td = 0;
    float me = 0.f, mr = 0.f;
};

// Mesh readers (src/raytracer.cpp:80-165).  verts: xyz per vertex; tris: 3 indices per triangle.
bool wapt_read_fast(const std::string& path, std::vector<float>& verts, std::vector<int32_t>& tris,
                    std::string& err);
bool wapt_read_obj(const std::string& path, std::vector<float>& verts, std::vector<int32_t>& tris,
                   std::string& err);
