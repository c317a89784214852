// Host side kept from the reference program: .waff scene files, OBJ and
// .fast meshes, tone mapping and the BMP writer.  Written against the file
// grammar and the parser's observable behaviour (src/myRenderer.cpp:315-683,
// src/raytracer.cpp:80-165), without Boost or tinyobjloader.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "wapt_internal.h"

namespace {

thread_local std::string g_last_error;

// Character cursor over the whole file.  The reference alternates between
// getc() scans for a marker character and fscanf() reads of numbers and
// words; the methods below reproduce those two access patterns.
struct Cursor {
    std::string s;
    size_t i = 0;
    bool eof() const { return i >= s.size(); }
    int get() { return eof() ? EOF : (unsigned char)s[i++]; }
    void unget() {
        if (i > 0) i--;
    }
    void skip_ws() {
        while (!eof() && isspace((unsigned char)s[i])) i++;
    }
    void skip_line() {
        while (!eof() && s[i] != '\n') i++;
    }
    // fscanf(" word"): optional white space then the literal
    bool literal(const char* w) {
        size_t save = i;
        skip_ws();
        size_t n = strlen(w);
        if (s.compare(i, n, w) == 0) {
            i += n;
            return true;
        }
        i = save;
        return false;
    }
    bool number(double& v) {
        size_t save = i;
        skip_ws();
        if (eof()) {
            i = save;
            return false;
        }
        const char* b = s.c_str() + i;
        char* e = nullptr;
        v = strtod(b, &e);
        if (e == b) {
            i = save;
            return false;
        }
        i += (size_t)(e - b);
        return true;
    }
    bool f(float& v) {
        double d;
        if (!number(d)) return false;
        v = (float)d;
        return true;
    }
    bool integer(int& v) {
        size_t save = i;
        skip_ws();
        if (eof()) {
            i = save;
            return false;
        }
        const char* b = s.c_str() + i;
        char* e = nullptr;
        long l = strtol(b, &e, 10);
        if (e == b) {
            i = save;
            return false;
        }
        v = (int)l;
        i += (size_t)(e - b);
        return true;
    }
    bool word(std::string& w) {
        skip_ws();
        if (eof()) return false;
        size_t b = i;
        while (!eof() && !isspace((unsigned char)s[i])) i++;
        w = s.substr(b, i - b);
        return true;
    }
};

void set3(float* d, float a, float b, float c) {
    d[0] = a;
    d[1] = b;
    d[2] = c;
}

// wa_shader::wa_shader() + in-class initialisers (src/objects.cpp:5-17, src/objects.h:29-39)
wapt_material default_material() {
    wapt_material m;
    memset(&m, 0, sizeof(m));
    set3(m.kd, 0.5f, 0.5f, 0.5f);
    set3(m.ks, 0.5f, 0.5f, 0.5f);
    m.ior = 1.0f;
    m.flags = WAPT_MAT_OPAQUE | WAPT_MAT_CLEAR | WAPT_MAT_DIFFUSE;
    return m;
}

bool file_exists(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

std::string dir_of(const std::string& p) {
    size_t k = p.find_last_of("/\\");
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}

bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

int wapt_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

extern "C" const char* wapt_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* wapt_version(void) { return "wapt-mi355x 0.1 (gfx950)"; }

// ---------------------------------------------------------------- meshes

bool wapt_read_fast(const std::string& path, std::vector<float>& verts, std::vector<int32_t>& tris,
                    std::string& err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) {
        err = "cannot open " + path;
        return false;
    }
    int32_t hdr[2];
    in.read((char*)hdr, 8);
    if (!in || hdr[0] < 0 || hdr[1] < 0) {
        err = "bad .fast header in " + path;
        return false;
    }
    size_t nv = (size_t)hdr[0], nt = (size_t)hdr[1];
    // the counts come from the file: compare them with its size before anything is sized by them
    in.seekg(0, std::ios::end);
    const long long file_bytes = (long long)in.tellg();
    in.seekg(8, std::ios::beg);
    if (file_bytes < 0 || (unsigned long long)file_bytes < 8ull + 16ull * nv + 12ull * nt) {
        err = "truncated .fast file " + path;
        return false;
    }
    std::vector<float> v4(nv * 4);
    tris.resize(nt * 3);
    in.read((char*)v4.data(), (std::streamsize)(nv * 16));
    in.read((char*)tris.data(), (std::streamsize)(nt * 12));
    if (!in) {
        err = "truncated .fast file " + path;
        return false;
    }
    verts.resize(nv * 3);
    for (size_t i = 0; i < nv; i++)
        for (int a = 0; a < 3; a++) verts[3 * i + a] = v4[4 * i + a];
    for (size_t i = 0; i < nt * 3; i++)
        if (tris[i] < 0 || (size_t)tris[i] >= nv) {
            err = "vertex index out of range in " + path;
            return false;
        }
    return true;
}

// Positions and the faces of the first shape only, polygons fan-triangulated:
// what convertObjFileToEmbreeFormat keeps of tinyobj's result
// (src/raytracer.cpp:138-165).
bool wapt_read_obj(const std::string& path, std::vector<float>& verts, std::vector<int32_t>& tris,
                   std::string& err) {
    std::ifstream in(path);
    if (!in) {
        err = "cannot open " + path;
        return false;
    }
    verts.clear();
    tris.clear();
    std::string line;
    bool shape_closed = false;
    std::vector<int32_t> face;
    while (std::getline(in, line)) {
        size_t k = 0;
        while (k < line.size() && isspace((unsigned char)line[k])) k++;
        if (k >= line.size() || line[k] == '#') continue;
        if (line[k] == 'v' && k + 1 < line.size() && isspace((unsigned char)line[k + 1])) {
            const char* p = line.c_str() + k + 1;
            char* e;
            for (int a = 0; a < 3; a++) {
                verts.push_back(strtof(p, &e));
                p = e;
            }
        } else if (line[k] == 'f' && k + 1 < line.size() && isspace((unsigned char)line[k + 1])) {
            if (shape_closed) continue;
            face.clear();
            const char* p = line.c_str() + k + 1;
            for (;;) {
                while (*p && isspace((unsigned char)*p)) p++;
                if (!*p) break;
                char* e;
                long vi = strtol(p, &e, 10);
                if (e == p) break;
                p = e;
                while (*p && !isspace((unsigned char)*p)) p++;  // skip /vt/vn
                long nv = (long)(verts.size() / 3);
                long idx = vi > 0 ? vi - 1 : nv + vi;
                face.push_back((int32_t)idx);
            }
            for (size_t i = 2; i < face.size(); i++) {
                tris.push_back(face[0]);
                tris.push_back(face[i - 1]);
                tris.push_back(face[i]);
            }
        } else if ((line[k] == 'o' || line[k] == 'g') && (k + 1 >= line.size() || isspace((unsigned char)line[k + 1]))) {
            if (!tris.empty()) shape_closed = true;  // later shapes are ignored
        }
    }
    long nv = (long)(verts.size() / 3);
    for (int32_t t : tris)
        if (t < 0 || t >= nv) {
            err = "vertex index out of range in " + path;
            return false;
        }
    if (tris.empty()) {
        err = "no faces in " + path;
        return false;
    }
    return true;
}

static int convert_obj_to_fast(const char* obj_path, const char* fast_path) {
    if (!obj_path || !fast_path) return wapt_fail(WAPT_E_INVALID, "null path");
    std::vector<float> v;
    std::vector<int32_t> t;
    std::string err;
    if (!wapt_read_obj(obj_path, v, t, err)) return wapt_fail(WAPT_E_IO, err);
    std::ofstream of(fast_path, std::ios::binary);
    if (!of) return wapt_fail(WAPT_E_IO, std::string("cannot write ") + fast_path);
    int32_t hdr[2] = {(int32_t)(v.size() / 3), (int32_t)(t.size() / 3)};
    of.write((const char*)hdr, 8);
    for (size_t i = 0; i < v.size() / 3; i++) {
        float q[4] = {v[3 * i], v[3 * i + 1], v[3 * i + 2], 0.f};
        of.write((const char*)q, 16);
    }
    of.write((const char*)t.data(), (std::streamsize)(t.size() * 4));
    return of ? WAPT_OK : wapt_fail(WAPT_E_IO, std::string("write failed: ") + fast_path);
}

extern "C" int wapt_convert_obj_to_fast(const char* obj_path, const char* fast_path) {
    return wapt_guarded([&] { return convert_obj_to_fast(obj_path, fast_path); });
}

// ---------------------------------------------------------------- .waff

static int waff_load(const char* path, wapt_waff** out) {
    if (!path || !out) return wapt_fail(WAPT_E_INVALID, "null argument");
    *out = nullptr;
    std::ifstream in(path, std::ios::binary);
    if (!in) return wapt_fail(WAPT_E_IO, std::string("cannot open ") + path);
    std::stringstream ss;
    ss << in.rdbuf();
    Cursor c;
    c.s = ss.str();
    std::unique_ptr<wapt_waff> w(new wapt_waff());
    auto fail = [&](const std::string& m) { return wapt_fail(WAPT_E_IO, std::string(path) + ": " + m); };

    // camera block: everything up to the first 'v' is skipped (myRenderer.cpp:325-340)
    int ch;
    while ((ch = c.get()) != EOF && ch != 'v') {
    }
    if (ch == EOF) return fail("no camera ('v') block");
    float at[3];
    float hither;
    wapt_camera& cam = w->camera;
    memset(&cam, 0, sizeof(cam));
    bool ok = c.integer(cam.type);
    ok = ok && c.literal("from") && c.f(cam.P[0]) && c.f(cam.P[1]) && c.f(cam.P[2]);
    ok = ok && c.literal("at") && c.f(at[0]) && c.f(at[1]) && c.f(at[2]);
    ok = ok && c.literal("up") && c.f(cam.up[0]) && c.f(cam.up[1]) && c.f(cam.up[2]);
    ok = ok && c.literal("angle") && c.f(cam.fov_deg);
    ok = ok && c.literal("hither") && c.f(hither);
    ok = ok && c.literal("resolution") && c.integer(cam.resI) && c.integer(cam.resJ);
    ok = ok && c.literal("width") && c.f(cam.sensor_width);
    if (!ok) return fail("malformed camera block (need: v from at up angle hither resolution width)");
    {
        float d[3] = {at[0] - cam.P[0], at[1] - cam.P[1], at[2] - cam.P[2]};
        float m = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        set3(cam.D, d[0] / m, d[1] / m, d[2] / m);
    }
    if (cam.resI <= 0 || cam.resJ <= 0) return fail("resolution must be positive");

    // render settings: next 't' then "d <int>", next 'm' then "e <float>", next 'm' then "r <float>"
    // (myRenderer.cpp:350-365; the reference spins forever at end of file, we report it)
    while ((ch = c.get()) != EOF && ch != 't') {
    }
    if (ch == EOF || !c.literal("d") || !c.integer(w->td)) return fail("missing 'td <int>'");
    while ((ch = c.get()) != EOF && ch != 'm') {
    }
    if (ch == EOF || !c.literal("e") || !c.f(w->me)) return fail("missing 'me <float>'");
    while ((ch = c.get()) != EOF && ch != 'm') {
    }
    if (ch == EOF || !c.literal("r") || !c.f(w->mr)) return fail("missing 'mr <float>'");

    // global medium (myRenderer.cpp:370-422)
    wapt_material gs = default_material();
    int has_medium = 0;
    while ((ch = c.get()) != EOF) {
        if (ch == '#') {
            c.skip_line();
            continue;
        }
        if (ch == 'm') {
            c.integer(has_medium);
            if (has_medium != 0) {
                c.f(gs.ss[0]);
                c.f(gs.ss[1]);
                c.f(gs.ss[2]);
                c.f(gs.sa[0]);
                c.f(gs.sa[1]);
                c.f(gs.sa[2]);
            }
            break;
        }
        if (ch == 'l' || ch == 'g') {
            c.unget();
            break;
        }
    }
    if (has_medium == 0) {
        gs.flags |= WAPT_MAT_CLEAR;
    } else {
        gs.flags &= ~WAPT_MAT_CLEAR;
        for (int a = 0; a < 3; a++) gs.st[a] = gs.ss[a] + gs.sa[a];
    }
    w->global_medium = gs;

    // lights (myRenderer.cpp:427-497); each one also takes an object slot
    while ((ch = c.get()) != EOF) {
        if (ch == '#') {
            c.skip_line();
            continue;
        }
        if (ch == 'l') {
            int c1 = c.get();
            wapt_light L;
            memset(&L, 0, sizeof(L));
            bool have = false;
            if (c1 == 'a') {
                float v[12];
                bool good = true;
                for (int k = 0; k < 12; k++) good = good && c.f(v[k]);
                if (!good) return fail("area light needs 12 numbers");
                L.type = 0;
                set3(L.P, v[0], v[1], v[2]);
                set3(L.U, v[3], v[4], v[5]);
                set3(L.V, v[6], v[7], v[8]);
                set3(L.power, v[9], v[10], v[11]);
                have = true;
            } else if (c1 == 's') {
                float v[7];
                bool good = true;
                for (int k = 0; k < 7; k++) good = good && c.f(v[k]);
                if (!good) return fail("sphere light needs 7 numbers");
                L.type = 1;
                L.radius = v[0];
                set3(L.P, v[1], v[2], v[3]);
                set3(L.power, v[4], v[5], v[6]);
                have = true;
            } else if (c1 == ' ') {
                // point light = sphere of radius 1, power defaults to 10000
                float v[6] = {0, 0, 0, 10000.f, 10000.f, 10000.f};
                for (int k = 0; k < 6; k++)
                    if (!c.f(v[k])) break;
                L.type = 1;
                L.radius = 1.0f;
                set3(L.P, v[0], v[1], v[2]);
                set3(L.power, v[3], v[4], v[5]);
                have = true;
            }
            if (!have) continue;
            w->lights.push_back(L);
            wapt_material lm = default_material();  // diffuse 0.5 reflector that also emits
            lm.flags |= WAPT_MAT_LIGHT;
            w->materials.push_back(lm);
        }
        if (ch == 'g') {
            c.unget();
            break;
        }
    }

    // geometry (myRenderer.cpp:503-644).  The numeric locals below live across
    // 'g' lines exactly as in the reference, so a later line that does not set
    // one of them inherits the previous line's value.
    float tr = 0, tg = 0, tb = 0, ar = 0, ag = 0, ab = 0, er = 0, eg = 0, eb = 0, sr = 0, sg = 0, sb = 0;
    float ior = 1.0f;
    float ks_sticky[3] = {0.5f, 0.5f, 0.5f};  // initialize() does not reset ks
    while ((ch = c.get()) != EOF) {
        if (ch == '#') {
            c.skip_line();
            continue;
        }
        if (ch != 'g') continue;
        std::string type, file;
        if (!c.word(type)) return fail("geometry line without a type");
        wapt_material m;
        memset(&m, 0, sizeof(m));
        set3(m.kd, 0.5f, 0.5f, 0.5f);
        memcpy(m.ks, ks_sticky, sizeof(ks_sticky));
        m.flags = WAPT_MAT_OPAQUE | WAPT_MAT_CLEAR;
        int vtype = 1;
        if (type == "diffuse") {
            float d[3];
            if (!(c.f(d[0]) && c.f(d[1]) && c.f(d[2]) && c.word(file))) return fail("g diffuse r g b file");
            m.flags |= WAPT_MAT_DIFFUSE;
            set3(m.kd, d[0], d[1], d[2]);
            set3(m.ks, 0, 0, 0);
            tr = tg = tb = 0.f;
        } else if (type == "specular") {
            if (!(c.f(sr) && c.f(sg) && c.f(sb) && c.word(file))) return fail("g specular r g b file");
            m.flags |= WAPT_MAT_SPECULAR;
            set3(m.kd, 0, 0, 0);
            set3(m.ks, sr, sg, sb);
            tr = tg = tb = 0.f;
        } else if (type == "sss") {
            set3(m.kd, 0, 0, 0);
            m.flags = 0;  // neither opaque nor clear; the reference's parser never raises the sss flag either
            if (!c.integer(vtype)) return fail("g sss <type> ...");
            if (vtype == 0) {
                tr = tg = tb = 1.f;
            } else {
                bool good = c.f(tr) && c.f(tg) && c.f(tb) && c.f(ar) && c.f(ag) && c.f(ab) && c.f(er) && c.f(eg) &&
                            c.f(eb) && c.f(sr) && c.f(sg) && c.f(sb);
                if (!good) return fail("g sss 1 needs 12 numbers");
            }
            if (!c.word(file)) return fail("g sss: missing file");
        } else if (type == "dielectric") {
            set3(m.kd, 0, 0, 0);
            set3(m.ks, 1, 1, 1);
            m.flags = WAPT_MAT_CLEAR | WAPT_MAT_DIELECTRIC;
            tr = tg = tb = 1.f;
            if (!c.integer(vtype)) return fail("g dielectric <type> ...");
            if (vtype != 0) {
                if (!c.f(ior)) return fail("g dielectric 1 <ior> file");
            }
            if (!c.word(file)) return fail("g dielectric: missing file");
        } else if (type == "dielectricsss") {
            set3(m.kd, 0, 0, 0);
            set3(m.ks, 1, 1, 1);
            m.flags = WAPT_MAT_DIELECTRICSSS;
            if (!c.integer(vtype)) return fail("g dielectricsss <type> ...");
            if (vtype == 0) {
                tr = tg = tb = 1.f;
            } else {
                bool good = c.f(ior) && c.f(tr) && c.f(tg) && c.f(tb) && c.f(ar) && c.f(ag) && c.f(ab) && c.f(er) &&
                            c.f(eg) && c.f(eb) && c.f(sr) && c.f(sg) && c.f(sb);
                if (!good) return fail("g dielectricsss 1 needs 13 numbers");
            }
            if (!c.word(file)) return fail("g dielectricsss: missing file");
        } else {
            return fail("unknown geometry type '" + type + "'");
        }
        memcpy(ks_sticky, m.ks, sizeof(ks_sticky));
        m.ior = ior;
        set3(m.kt, tr, tg, tb);
        set3(m.sa, ar, ag, ab);
        set3(m.ss, sr, sg, sb);
        set3(m.st, ar + sr, ag + sg, ab + sb);
        (void)er;
        (void)eg;
        (void)eb;
        w->materials.push_back(m);
        w->mesh_files.push_back(file);
    }

    // flatten: light quads first, meshes after, object index = geomID (content.cpp:13-40)
    const int nl = (int)w->lights.size();
    for (int l = 0; l < nl; l++) {
        const wapt_light& L = w->lights[l];
        if (L.type != 0) continue;  // sphere lights own an object slot but no geometry
        float v0[3], v1[3], v2[3], v3[3];
        for (int a = 0; a < 3; a++) {
            v0[a] = L.P[a];
            v1[a] = L.P[a] + L.U[a];
            v2[a] = L.P[a] + L.U[a] + L.V[a];
            v3[a] = L.P[a] + L.V[a];
        }
        const float* t0[3] = {v0, v1, v3};
        const float* t1[3] = {v2, v3, v1};
        for (auto& tri : {t0, t1}) {
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) w->tri_verts.push_back(tri[k][a]);
            w->tri_obj.push_back(l);
            w->tri_prim.push_back(0);
        }
    }
    std::string base = dir_of(path);
    for (size_t g = 0; g < w->mesh_files.size(); g++) {
        std::string f = w->mesh_files[g];
        if (!file_exists(f)) {
            std::string alt = base + "/" + f;
            if (file_exists(alt)) f = alt;
        }
        std::vector<float> v;
        std::vector<int32_t> t;
        std::string err;
        bool good = ends_with(f, "fast") ? wapt_read_fast(f, v, t, err) : wapt_read_obj(f, v, t, err);
        if (!good) return fail(err);
        size_t nt = t.size() / 3;
        size_t b = w->tri_verts.size();
        w->tri_verts.resize(b + nt * 9);
        for (size_t i = 0; i < nt; i++) {
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) w->tri_verts[b + 9 * i + 3 * k + a] = v[3 * (size_t)t[3 * i + k] + a];
            w->tri_obj.push_back(nl + (int)g);
            w->tri_prim.push_back((int32_t)i);
        }
    }
    *out = w.release();
    return WAPT_OK;
}

extern "C" int wapt_waff_load(const char* path, wapt_waff** out) {
    return wapt_guarded([&] { return waff_load(path, out); });
}

extern "C" int wapt_waff_desc(const wapt_waff* w, wapt_scene_desc* d) {
    if (!w || !d) return wapt_fail(WAPT_E_INVALID, "null argument");
    memset(d, 0, sizeof(*d));
    d->camera = w->camera;
    d->global_medium = w->global_medium;
    d->n_lights = (int32_t)w->lights.size();
    d->lights = w->lights.data();
    d->n_objects = (int32_t)w->materials.size();
    d->materials = w->materials.data();
    d->n_tris = w->tri_obj.size();
    d->tri_verts = w->tri_verts.data();
    d->tri_obj = w->tri_obj.data();
    d->tri_prim = w->tri_prim.data();
    // worldBoundSphere_P / _rad, src/unidirectionalPathTracer.cpp:36-37
    d->world_center[0] = d->world_center[1] = d->world_center[2] = 275.0f;
    d->world_radius = 1100.0f;
    return WAPT_OK;
}

extern "C" int wapt_waff_settings(const wapt_waff* w, int32_t* td, float* me, float* mr) {
    if (!w) return wapt_fail(WAPT_E_INVALID, "null argument");
    if (td) *td = w->td;
    if (me) *me = w->me;
    if (mr) *mr = w->mr;
    return WAPT_OK;
}

extern "C" void wapt_waff_free(wapt_waff* w) { delete w; }

// ---------------------------------------------------------------- output

extern "C" int wapt_tonemap(float* fb, int32_t w, int32_t h) {
    if (!fb || w <= 0 || h <= 0) return wapt_fail(WAPT_E_INVALID, "bad framebuffer");
    size_t n = (size_t)w * h * 3;
    for (size_t i = 0; i < n; i++) {
        float v = (float)pow((double)fb[i], 0.45);
        if (v < 0.0f) v = 0.0f;  // NaN (negative input) fails both tests and stays, as in the reference
        if (v > 1.0f) v = 1.0f;
        fb[i] = v;
    }
    return WAPT_OK;
}

static int write_bmp(const char* path, int32_t w, int32_t h, const float* px) {
    if (!path || !px || w <= 0 || h <= 0) return wapt_fail(WAPT_E_INVALID, "bad image");
    size_t total = 138 + (size_t)w * h * 4;
    std::vector<unsigned char> d(total, 0);
    auto u16 = [&](size_t off, uint16_t v) { memcpy(&d[off], &v, 2); };
    auto u32 = [&](size_t off, uint32_t v) { memcpy(&d[off], &v, 4); };
    u16(0, 0x4D42);
    u32(2, (uint32_t)total);
    u32(10, 138);
    u32(14, 124);
    u32(18, (uint32_t)w);
    u32(22, (uint32_t)h);
    u16(26, 1);
    u16(28, 32);
    u32(30, 3);
    u32(54, 0x00ff0000u);
    u32(58, 0x0000ff00u);
    u32(62, 0x000000ffu);
    u32(66, 0xff000000u);
    u32(70, 0x73524742u);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        int r = (int)(px[3 * i + 0] * 255.0), g = (int)(px[3 * i + 1] * 255.0), b = (int)(px[3 * i + 2] * 255.0);
        d[138 + 4 * i + 0] = (unsigned char)b;
        d[138 + 4 * i + 1] = (unsigned char)g;
        d[138 + 4 * i + 2] = (unsigned char)r;
        d[138 + 4 * i + 3] = 255;
    }
    std::ofstream of(path, std::ios::binary);
    if (!of) return wapt_fail(WAPT_E_IO, std::string("cannot write ") + path);
    of.write((const char*)d.data(), (std::streamsize)total);
    return of ? WAPT_OK : wapt_fail(WAPT_E_IO, std::string("write failed: ") + path);
}

extern "C" int wapt_write_bmp(const char* path, int32_t w, int32_t h, const float* px) {
    return wapt_guarded([&] { return write_bmp(path, w, h, px); });
}
