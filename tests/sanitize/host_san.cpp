// Driver of tests/test_sanitizers.py: runs the host side of the library (scene file and mesh readers, BVH builder,
// tone map, BMP writer) and the CPU checker under AddressSanitizer/UndefinedBehaviorSanitizer (or ThreadSanitizer,
// mode "threads").  Built from the sources where they lie; nothing here touches a GPU.
//   host_san <mode> <scene dir with test.waff, box.obj> : mode = host | fuzz | oracle | threads
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/wapt.h"
#include "../../pathtracer_amd/csrc/bvh8.h"

extern "C" {
struct oracle_scene;
oracle_scene* oracle_scene_create(const void* desc);
void oracle_scene_destroy(oracle_scene*);
int oracle_render(const oracle_scene*, const void* params, float* fb, int threads, uint64_t* rays_out);
}

#define CHECK(x)                                                              \
    do {                                                                      \
        if (!(x)) {                                                           \
            fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #x, __LINE__, wapt_last_error()); \
            exit(2);                                                          \
        }                                                                     \
    } while (0)

static std::string slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static void spit(const std::string& p, const std::string& s) {
    std::ofstream f(p, std::ios::binary);
    f.write(s.data(), (std::streamsize)s.size());
}

static int run_host(const std::string& dir) {
    wapt_waff* w = nullptr;
    CHECK(wapt_waff_load((dir + "/test.waff").c_str(), &w) == WAPT_OK);
    wapt_scene_desc d;
    CHECK(wapt_waff_desc(w, &d) == WAPT_OK);
    int32_t td;
    float me, mr;
    CHECK(wapt_waff_settings(w, &td, &me, &mr) == WAPT_OK);
    CHECK(d.n_tris > 0 && d.n_objects > d.n_lights);
    Bvh8 bvh;
    const char* err = nullptr;
    CHECK(bvh8_build(d.tri_verts, d.tri_obj, d.tri_prim, d.n_tris, &bvh, &err));
    CHECK(bvh.tris.size() == d.n_tris && !bvh.nodes.empty());
    // degenerate inputs of the builder: no triangles, a NaN, all triangles identical
    Bvh8 b2;
    CHECK(!bvh8_build(d.tri_verts, d.tri_obj, d.tri_prim, 0, &b2, &err));
    std::vector<float> v(d.tri_verts, d.tri_verts + 9 * d.n_tris);
    v[4] = NAN;
    CHECK(!bvh8_build(v.data(), d.tri_obj, d.tri_prim, d.n_tris, &b2, &err));
    std::vector<float> same(9 * 1000);
    for (size_t i = 0; i < same.size(); i++) same[i] = d.tri_verts[i % 9];
    std::vector<int32_t> zeros(1000, 0);
    CHECK(bvh8_build(same.data(), zeros.data(), zeros.data(), 1000, &b2, &err));
    wapt_waff_free(w);
    CHECK(wapt_convert_obj_to_fast((dir + "/box.obj").c_str(), (dir + "/box_out.fast").c_str()) == WAPT_OK);
    CHECK(wapt_convert_obj_to_fast((dir + "/nope.obj").c_str(), (dir + "/x.fast").c_str()) != WAPT_OK);
    std::vector<float> fb(7 * 5 * 3);
    for (size_t i = 0; i < fb.size(); i++) fb[i] = (float)i * 0.02f;
    fb[3] = -1.0f;
    fb[4] = 1e30f;
    CHECK(wapt_tonemap(fb.data(), 7, 5) == WAPT_OK);
    CHECK(wapt_write_bmp((dir + "/o.bmp").c_str(), 7, 5, fb.data()) == WAPT_OK);
    CHECK(wapt_write_bmp((dir + "/no/such/dir/o.bmp").c_str(), 7, 5, fb.data()) != WAPT_OK);
    return 0;
}

// Truncated and corrupted scene and mesh files: every outcome but a status code is a bug (the reference hangs or
// reads unchecked on such input, src/myRenderer.cpp:355-363, 651).
static int run_fuzz(const std::string& dir) {
    const std::string waff = slurp(dir + "/test.waff"), obj = slurp(dir + "/box.obj"), fast = slurp(dir + "/ball.fast");
    int ok = 0, bad = 0;
    auto try_waff = [&](const std::string& text) {
        spit(dir + "/f.waff", text);
        wapt_waff* w = nullptr;
        int rc = wapt_waff_load((dir + "/f.waff").c_str(), &w);
        if (rc == WAPT_OK) {
            wapt_scene_desc d;
            CHECK(wapt_waff_desc(w, &d) == WAPT_OK);
            wapt_waff_free(w);
            ok++;
        } else {
            CHECK(w == nullptr);
            bad++;
        }
    };
    for (size_t cut = 0; cut <= waff.size(); cut += 7) try_waff(waff.substr(0, cut));
    uint32_t x = 12345u;
    for (int k = 0; k < 200; k++) {
        std::string t = waff;
        for (int m = 0; m < 3; m++) {
            x = x * 1664525u + 1013904223u;
            t[(x >> 8) % t.size()] = (char)(x >> 24);
        }
        try_waff(t);
    }
    // the mesh files the scene names, damaged
    for (size_t cut = 0; cut <= fast.size(); cut += 97) {
        spit(dir + "/ball.fast", fast.substr(0, cut));
        try_waff(waff);
    }
    for (int k = 0; k < 100; k++) {
        std::string t = fast;
        for (int m = 0; m < 4; m++) {
            x = x * 1664525u + 1013904223u;
            t[(x >> 8) % t.size()] = (char)(x >> 24);
        }
        spit(dir + "/ball.fast", t);
        try_waff(waff);
    }
    // headers that promise far more than the file holds (a count of 0x7fffffff would size a 32 GiB vector)
    for (int which = 0; which < 3; which++) {
        std::string t = fast;
        const int32_t huge = 0x7fffffff;
        if (which != 1) memcpy(&t[0], &huge, 4);
        if (which != 0) memcpy(&t[4], &huge, 4);
        spit(dir + "/ball.fast", t);
        wapt_waff* w = nullptr;
        CHECK(wapt_waff_load((dir + "/test.waff").c_str(), &w) == WAPT_E_IO);
        CHECK(w == nullptr);
        bad++;
    }
    spit(dir + "/ball.fast", fast);
    for (size_t cut = 0; cut <= obj.size(); cut += 11) {
        spit(dir + "/f.obj", obj.substr(0, cut));
        (void)wapt_convert_obj_to_fast((dir + "/f.obj").c_str(), (dir + "/f.fast").c_str());
    }
    for (int k = 0; k < 200; k++) {
        std::string t = obj;
        for (int m = 0; m < 3; m++) {
            x = x * 1664525u + 1013904223u;
            t[(x >> 8) % t.size()] = (char)(x >> 24);
        }
        spit(dir + "/f.obj", t);
        (void)wapt_convert_obj_to_fast((dir + "/f.obj").c_str(), (dir + "/f.fast").c_str());
    }
    printf("fuzz: %d accepted, %d refused\n", ok, bad);
    CHECK(ok > 0 && bad > 0);
    return 0;
}

static int run_oracle(const std::string& dir, int threads) {
    wapt_waff* w = nullptr;
    CHECK(wapt_waff_load((dir + "/test.waff").c_str(), &w) == WAPT_OK);
    wapt_scene_desc d;
    CHECK(wapt_waff_desc(w, &d) == WAPT_OK);
    d.camera.resI = 16;
    d.camera.resJ = 12;
    oracle_scene* s = oracle_scene_create(&d);
    CHECK(s != nullptr);
    wapt_render_params p;
    memset(&p, 0, sizeof(p));
    p.psamples = 2;
    p.hsamples = 2;
    p.lsamples = 1;
    p.max_depth = 3;
    p.me = 0.0f;
    p.mr = 0.01f;
    p.seed = 5;
    p.sample_begin = 0;
    p.sample_count = 2;
    std::vector<float> fb(16 * 12 * 3, 0.f);
    uint64_t rays = 0;
    CHECK(oracle_render(s, &p, fb.data(), threads, &rays) == 0);
    double sum = 0;
    for (float f : fb) sum += f;
    printf("oracle: %llu rays, mean %.6g\n", (unsigned long long)rays, sum / fb.size());
    CHECK(rays > 0 && sum == sum);
    oracle_scene_destroy(s);
    wapt_waff_free(w);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    const std::string mode = argv[1], dir = argv[2];
    if (mode == "host") return run_host(dir);
    if (mode == "fuzz") return run_fuzz(dir);
    if (mode == "oracle") return run_oracle(dir, 2);
    if (mode == "threads") return run_oracle(dir, 4);
    return 64;
}
