// Internals shared by the translation units of libwapt.so (not installed).
#pragma once
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/wapt.h"

// Records the message returned by wapt_last_error() and returns `code`.
int wapt_fail(int code, const std::string& msg);

// Runs the body of an extern "C" entry point: no C++ exception leaves the library.  Running out of host memory
// becomes WAPT_E_LIMIT, anything else WAPT_E_INVALID with the exception's text.
template <class F>
int wapt_guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try {
            return wapt_fail(WAPT_E_LIMIT, "out of memory");  // (short enough to need no allocation itself)
        } catch (...) {
            return WAPT_E_LIMIT;
        }
    } catch (const std::exception& e) {
        try {
            return wapt_fail(WAPT_E_INVALID, std::string("internal error: ") + e.what());
        } catch (...) {
            return WAPT_E_INVALID;
        }
    } catch (...) {
        try {
            return wapt_fail(WAPT_E_INVALID, "internal error");
        } catch (...) {
            return WAPT_E_INVALID;
        }
    }
}

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
