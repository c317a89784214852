// CPU ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A scalar, recursive restatement of the reference's per-pixel radiance loop
// (willalexander/pathtracer, files cited function by function below), used by
// tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py
// to CHECK the GPU library.  Nothing under pathtracer_amd/ links, imports or
// calls it.
//
// What is and is not pinned:
//  * The estimator (camera, samplers, lights, Snell/Fresnel, shadow walk,
//    media) follows the reference sources line by line and is pinned against
//    the linear-radiance known answers of the reference's own code for the
//    bundled Cornell box and against masked regions of the published images
//    (tests/golden/).  The reference cannot be compiled in this image
//    (Boost, tinyobjloader and Embree 3 are absent), so no binary of it runs.
//  * Ray/triangle intersection: the reference calls Embree 3 (unpinned
//    version; src/content.cpp:97).  Embree is not available, the reference
//    holds no test vectors at that boundary: INTERSECTOR PARITY UNPINNED.
//    The stand-in is a double-precision Moeller-Trumbore closest hit with
//    Embree's interval rule (tnear, tfar], two-sided, either brute force or
//    through a binary BVH that returns the same answer.
//  * Random numbers: the reference shares one std::default_random_engine
//    between four threads without a lock (src/randomnumbers.cpp:11,30-41) and
//    cannot reproduce its own images.  The checker uses the same keyed hash as
//    the GPU library (restated below) so both draw identical samples.  A second
//    mode (OParams::profile bit 2) draws every number from a std::mt19937_64 of
//    the rendering thread instead, one after the other in call order - the
//    reference's own scheme without its race: a generator that shares nothing
//    with the keyed hash, to check the hash against.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

// ------------------------------------------------------------------ plain C structs (same layout as the library's ABI)

extern "C" {
struct OCamera {
    int32_t type;
    float P[3], D[3], up[3];
    float fov_deg;
    int32_t resI, resJ;
    float sensor_width;
};
struct OMaterial {
    float kd[3], ks[3], kt[3], sa[3], ss[3], st[3];
    float ior;
    uint32_t flags;
};
struct OLight {
    int32_t type;
    float P[3], U[3], V[3];
    float radius;
    float power[3];
};
struct OSceneDesc {
    OCamera camera;
    OMaterial global_medium;
    int32_t n_lights;
    const OLight* lights;
    int32_t n_objects;
    const OMaterial* materials;
    uint64_t n_tris;
    const float* tri_verts;
    const int32_t* tri_obj;
    const int32_t* tri_prim;
    float world_center[3];
    float world_radius;
};
struct OParams {
    int32_t psamples, hsamples, lsamples, max_depth;
    float me, mr;
    uint64_t seed;
    int32_t sample_begin, sample_count;
    int32_t profile, reserved;
};
struct ORay {
    float org[3], tnear, dir[3], tfar;
};
struct OHit {
    float t;
    int32_t prim, obj;
    float Ng[3];
};
}

enum {
    F_OPAQUE = 1u << 0, F_CLEAR = 1u << 1, F_LIGHT = 1u << 2, F_DIFFUSE = 1u << 3,
    F_SPECULAR = 1u << 4, F_SSS = 1u << 5, F_DIELECTRIC = 1u << 6, F_DIELECTRICSSS = 1u << 7
};

// ------------------------------------------------------------------ keyed random numbers (this build's design, not the reference's)

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
static inline uint32_t mix32(uint32_t h, uint32_t w) {
    w *= 0xcc9e2d51u; w = rotl32(w, 15); w *= 0x1b873593u;
    h ^= w; h = rotl32(h, 13);
    return h * 5u + 0xe6546b64u;
}
static inline uint32_t key_root(uint64_t seed, uint32_t sample) {
    return fmix32(mix32(mix32((uint32_t)seed, (uint32_t)(seed >> 32)), sample));
}
static inline uint32_t key_child(uint32_t key, uint32_t slot) { return fmix32(mix32(key, slot) ^ 0x9e3779b9u); }
static inline uint32_t stream_of(uint32_t key, uint32_t pixel) { return mix32(key, pixel); }
// sequential mode: the generator of the calling thread (null: keyed numbers)
static thread_local std::mt19937_64* tl_sequential = nullptr;
static inline float rnd(uint32_t stream, uint32_t dim) {
    if (tl_sequential) return (float)((*tl_sequential)() >> 40) * (1.0f / 16777216.0f);  // randFloat, randomnumbers.cpp:30-41
    return (float)(fmix32(mix32(stream, dim)) >> 8) * (1.0f / 16777216.0f);
}
enum { DIM_FIXED = 0u << 20, DIM_HEMI = 1u << 20, DIM_NEE = 2u << 20, DIM_MARCH = 3u << 20, DIM_VNEE = 4u << 20,
       DIM_VPROBE = 5u << 20, DIM_VSINGLE = 6u << 20, DIM_VCHILD = 7u << 20 };
enum { DIM_CAM0 = 0, DIM_CAM1 = 1, DIM_SURF = 2, DIM_VOL = 3, DIM_DIEL = 4 };
enum { SLOT_HEMI = 1u << 24, SLOT_SPEC = 2u << 24, SLOT_TRANS = 3u << 24, SLOT_VOL = 4u << 24 };

// ------------------------------------------------------------------ float3 / wa_colour (src/float3.*, src/fundamentals.cpp:29-193)

#define M_PI_REF 3.1415926535  // src/fundamentals.h:12

struct f3 {
    float x, y, z;
};
static inline f3 mk(float x, float y, float z) { return f3{x, y, z}; }
static inline f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 operator*(double s, f3 a) { return mk((float)(s * a.x), (float)(s * a.y), (float)(s * a.z)); }
static inline f3 operator*(f3 a, double s) { return s * a; }
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float mag(f3 a) { return sqrtf(dot(a, a)); }
static inline f3 normalized(f3 a) {
    float m = mag(a);
    return mk(a.x / m, a.y / m, a.z / m);
}
static inline bool is0(f3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
static inline f3 ld(const float* p) { return mk(p[0], p[1], p[2]); }

struct col {
    float r, g, b;
};
static inline col C(float v) { return col{v, v, v}; }
static inline col C(const float* p) { return col{p[0], p[1], p[2]}; }
static inline col operator*(col a, col b) { return col{a.r * b.r, a.g * b.g, a.b * b.b}; }
static inline col operator*(col a, float s) { return col{a.r * s, a.g * s, a.b * s}; }
static inline col operator*(float s, col a) { return a * s; }
static inline col operator+(col a, col b) { return col{a.r + b.r, a.g + b.g, a.b + b.b}; }
static inline void operator+=(col& a, col b) { a.r += b.r; a.g += b.g; a.b += b.b; }
static inline void operator*=(col& a, col b) { a.r *= b.r; a.g *= b.g; a.b *= b.b; }
static inline bool iszero(col a) { return a.r == 0.f && a.g == 0.f && a.b == 0.f; }
static inline col cexp(col a) { return col{(float)exp(a.r), (float)exp(a.g), (float)exp(a.b)}; }

// ------------------------------------------------------------------ scene

struct Tri {
    double v0[3], e1[3], e2[3];
    int32_t obj, prim;
    float ng[3];
};
struct BNode {
    double lo[3], hi[3];
    uint32_t left, right, first, count;
};
struct Light {
    int type;
    f3 P, U, V, N;
    float area, rad;
    col radiance, irradiance;
};
struct Shader {
    col kd, ks, kt, ss, st;
    float ior;
    uint32_t flags;
    int objID;
    bool isClear() const { return flags & F_CLEAR; }
    bool isOpaque() const { return flags & F_OPAQUE; }
};
struct Camera {
    int type, resI, resJ;
    f3 P, D, side, up, sensorOrigin;
    float pixelWidth, sensorWidth;
};

struct oracle_scene {
    std::vector<Tri> tris;
    std::vector<uint32_t> order;  // BVH leaf order -> triangle index
    std::vector<BNode> nodes;
    std::vector<Shader> shaders;  // per object
    Shader global;
    std::vector<Light> lights;
    Camera cam;
    f3 wc;
    float wr;
};

// wa_camera ctor + setResolution (src/fundamentals.cpp:227-263)
static Camera make_camera(const OCamera& c) {
    Camera k;
    k.type = c.type;
    k.resI = c.resI;
    k.resJ = c.resJ;
    k.P = ld(c.P);
    k.D = ld(c.D);
    k.side = normalized(cross(k.D, ld(c.up)));
    k.up = cross(k.side, k.D);
    k.sensorWidth = c.sensor_width;
    float sensorDist = (float)(0.5 * k.sensorWidth / tan(0.5 * c.fov_deg * (M_PI_REF / 180.0)));
    f3 t = k.side + (((float)k.resJ / (float)k.resI) * k.up);
    k.sensorOrigin = k.P + sensorDist * k.D - (0.5 * k.sensorWidth) * t;
    k.pixelWidth = k.sensorWidth / (float)k.resI;
    return k;
}

// wa_areaLight / wa_sphereLight ctors (src/lights.cpp:24-34, 136-143)
static Light make_light(const OLight& l) {
    Light L;
    L.type = l.type;
    L.P = ld(l.P);
    L.U = ld(l.U);
    L.V = ld(l.V);
    L.rad = l.radius;
    L.area = 0.f;
    L.N = mk(0, 0, 0);
    if (l.type == 0) {
        L.N = normalized(cross(L.U, L.V));
        L.area = (float)(sqrt((double)(l.U[0] * l.U[0] + l.U[1] * l.U[1] + l.U[2] * l.U[2])) *
                         sqrt((double)(l.V[0] * l.V[0] + l.V[1] * l.V[1] + l.V[2] * l.V[2])));
        for (int a = 0; a < 3; a++) {
            float irr = (float)(0.5 * l.power[a]) / L.area;
            (&L.irradiance.r)[a] = irr;
            (&L.radiance.r)[a] = (float)(irr / M_PI_REF);
        }
    } else {
        for (int a = 0; a < 3; a++) {
            float irr = (float)(l.power[a] / (4.0 * M_PI_REF * pow((double)l.radius, 2)));
            (&L.irradiance.r)[a] = irr;
            (&L.radiance.r)[a] = (float)(irr / M_PI_REF);
        }
    }
    return L;
}

static Shader make_shader(const OMaterial& m, int id) {
    Shader s;
    s.kd = C(m.kd);
    s.ks = C(m.ks);
    s.kt = C(m.kt);
    s.ss = C(m.ss);
    s.st = C(m.st);
    s.ior = m.ior;
    s.flags = m.flags;
    s.objID = id;
    return s;
}

static void build_bvh(oracle_scene* s) {
    size_t n = s->tris.size();
    s->order.resize(n);
    for (size_t i = 0; i < n; i++) s->order[i] = (uint32_t)i;
    std::vector<double> cent(3 * n), lo(3 * n), hi(3 * n);
    for (size_t i = 0; i < n; i++) {
        const Tri& t = s->tris[i];
        for (int a = 0; a < 3; a++) {
            double p0 = t.v0[a], p1 = t.v0[a] + t.e1[a], p2 = t.v0[a] + t.e2[a];
            lo[3 * i + a] = std::min(p0, std::min(p1, p2));
            hi[3 * i + a] = std::max(p0, std::max(p1, p2));
            cent[3 * i + a] = 0.5 * (lo[3 * i + a] + hi[3 * i + a]);
        }
    }
    s->nodes.clear();
    s->nodes.reserve(2 * n);
    struct Job {
        uint32_t node, b, e;
    };
    std::vector<Job> stack;
    s->nodes.push_back(BNode());
    stack.push_back({0u, 0u, (uint32_t)n});
    while (!stack.empty()) {
        Job j = stack.back();
        stack.pop_back();
        BNode nd;
        double clo[3], chi[3];
        for (int a = 0; a < 3; a++) {
            nd.lo[a] = clo[a] = DBL_MAX;
            nd.hi[a] = chi[a] = -DBL_MAX;
        }
        for (uint32_t i = j.b; i < j.e; i++) {
            uint32_t t = s->order[i];
            for (int a = 0; a < 3; a++) {
                nd.lo[a] = std::min(nd.lo[a], lo[3 * (size_t)t + a]);
                nd.hi[a] = std::max(nd.hi[a], hi[3 * (size_t)t + a]);
                clo[a] = std::min(clo[a], cent[3 * (size_t)t + a]);
                chi[a] = std::max(chi[a], cent[3 * (size_t)t + a]);
            }
        }
        uint32_t cnt = j.e - j.b;
        nd.left = nd.right = 0;
        nd.first = j.b;
        nd.count = cnt;
        if (cnt > 4) {
            int ax = 0;
            if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
            if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
            uint32_t mid = j.b + cnt / 2;
            std::nth_element(s->order.begin() + j.b, s->order.begin() + mid, s->order.begin() + j.e,
                             [&](uint32_t a, uint32_t b) { return cent[3 * (size_t)a + ax] < cent[3 * (size_t)b + ax]; });
            nd.count = 0;
            nd.left = (uint32_t)s->nodes.size();
            nd.right = nd.left + 1;
            s->nodes.push_back(BNode());
            s->nodes.push_back(BNode());
            stack.push_back({nd.left, j.b, mid});
            stack.push_back({nd.right, mid, j.e});
        }
        s->nodes[j.node] = nd;
    }
}

// ------------------------------------------------------------------ closest hit (stand-in for rtcIntersect1, src/content.cpp:97)

// Ray/triangle test in the form Embree 3's default triangle intersector uses (embree3, version not pinned
// by the reference; kernels/geometry/triangle_intersector_moeller.h: C = v0 - org, R = C x D,
// U = R.e2, V = R.e1 and T = Ng.C signed by den = Ng.D, valid when U >= 0, V >= 0, U+V <= |den| and
// |den|*tnear < T <= |den|*tfar with e1 = v0-v1, e2 = v2-v0, Ng = e2 x e1), evaluated in double.
static inline bool tri_hit(const Tri& T_, const double* o, const double* d, double tnear, double tfar, double* t_out) {
    // T_.e1 = v1-v0 and T_.e2 = v2-v0 are stored; Embree's e1 is the negated first edge
    double e1[3] = {-T_.e1[0], -T_.e1[1], -T_.e1[2]};
    const double* e2 = T_.e2;
    double Ng[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
    double C[3] = {T_.v0[0] - o[0], T_.v0[1] - o[1], T_.v0[2] - o[2]};
    double R[3] = {C[1] * d[2] - C[2] * d[1], C[2] * d[0] - C[0] * d[2], C[0] * d[1] - C[1] * d[0]};
    double den = Ng[0] * d[0] + Ng[1] * d[1] + Ng[2] * d[2];
    if (den == 0.0) return false;
    double absDen = fabs(den), sgn = den < 0.0 ? -1.0 : 1.0;
    double U = (R[0] * e2[0] + R[1] * e2[1] + R[2] * e2[2]) * sgn;
    double V = (R[0] * e1[0] + R[1] * e1[1] + R[2] * e1[2]) * sgn;
    if (!(U >= 0.0) || !(V >= 0.0) || !(U + V <= absDen)) return false;
    double T = (Ng[0] * C[0] + Ng[1] * C[1] + Ng[2] * C[2]) * sgn;
    if (!(absDen * tnear < T) || !(T <= absDen * tfar)) return false;
    *t_out = T / absDen;
    return true;
}

// optional event log of one path (oracle_debug_path)
static thread_local std::vector<float>* tl_log = nullptr;
static inline void log_event(float kind, float depth, f3 p, col c);
static thread_local uint64_t tl_rays = 0;
static thread_local uint64_t tl_shadow_rays = 0;
static std::atomic<uint64_t> g_shadow_rays{0};

// Returns triangle index or -1; *t receives the hit distance.  Ties go to the lower triangle index.
static int closest_hit(const oracle_scene* s, const float* org, const float* dir, float tnear, float tfar, double* t_hit,
                       bool brute) {
    tl_rays++;
    double o[3] = {org[0], org[1], org[2]}, d[3] = {dir[0], dir[1], dir[2]};
    double best = tfar;
    int hit = -1;
    if (brute) {
        for (size_t i = 0; i < s->tris.size(); i++) {
            double t;
            if (tri_hit(s->tris[i], o, d, tnear, best, &t))
                if (t < best || hit < 0) {
                    best = t;
                    hit = (int)i;
                }
        }
        *t_hit = best;
        return hit;
    }
    double id[3];
    for (int a = 0; a < 3; a++) id[a] = 1.0 / d[a];
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BNode& n = s->nodes[stack[--sp]];
        double t0 = tnear, t1 = best;
        bool miss = false;
        for (int a = 0; a < 3; a++) {
            if (d[a] == 0.0) {
                if (o[a] < n.lo[a] || o[a] > n.hi[a]) miss = true;
                continue;
            }
            double ta = (n.lo[a] - o[a]) * id[a], tb = (n.hi[a] - o[a]) * id[a];
            if (ta > tb) std::swap(ta, tb);
            // widen by a few ulps so that the box test never rejects a hit the triangle test accepts
            ta -= 1e-9 * fabs(ta) + 1e-12;
            tb += 1e-9 * fabs(tb) + 1e-12;
            t0 = std::max(t0, ta);
            t1 = std::min(t1, tb);
        }
        if (miss || t0 > t1) continue;
        if (n.count) {
            for (uint32_t k = 0; k < n.count; k++) {
                uint32_t ti = s->order[n.first + k];
                double t;
                if (tri_hit(s->tris[ti], o, d, tnear, best, &t))
                    if (t < best || hit < 0 || (t == best && (int)ti < hit)) {
                        best = t;
                        hit = (int)ti;
                    }
            }
        } else if (sp + 2 <= 128) {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    *t_hit = best;
    return hit;
}

static inline void log_event(float kind, float depth, f3 p, col c) {
    if (!tl_log) return;
    float r[8] = {kind, depth, p.x, p.y, p.z, c.r, c.g, c.b};
    tl_log->insert(tl_log->end(), r, r + 8);
}

// wa_primitive as filled by wa_content::traceFAST (src/content.cpp:91-123)
struct Intr {
    f3 P, N;
    int objID;
    bool isLight;
    col irradiance;
};

static bool traceFAST(const oracle_scene* s, f3 P, f3 D, Intr* intr) {
    float o[3] = {P.x, P.y, P.z}, d[3] = {D.x, D.y, D.z};
    double t;
    int h = closest_hit(s, o, d, 0.001f, 10000000.0f, &t, s->tris.size() <= 64);
    float tfar = (float)t;
    intr->P = mk(o[0] + tfar * d[0], o[1] + tfar * d[1], o[2] + tfar * d[2]);
    if (h < 0) return false;
    const Tri& T = s->tris[h];
    intr->N = normalized(ld(T.ng));
    intr->objID = T.obj;
    intr->isLight = T.obj < (int)s->lights.size();
    intr->irradiance = intr->isLight ? s->lights[T.obj].irradiance : C(0.f);
    return true;
}

// wa_primitive::Le (src/primitive.cpp:21-29)
static col Le(const Intr& i) {
    if (i.isLight) return col{(float)(i.irradiance.r / M_PI_REF), (float)(i.irradiance.g / M_PI_REF), (float)(i.irradiance.b / M_PI_REF)};
    return C(0.f);
}

// ------------------------------------------------------------------ samplers (src/fundamentals.cpp)

static void frame(f3 D, f3* ta, f3* tb) {  // :412-414
    f3 a = normalized(cross(D, mk(1, 0, 0)));
    if (D.y == 0.0f && D.z == 0.0f) a = normalized(cross(D, mk(0, 1, 0)));
    *ta = a;
    *tb = normalized(cross(a, D));
}
static f3 hemisphereSampleUniform(f3 D, float u1, float u2) {  // :408-425
    f3 ta, tb;
    frame(D, &ta, &tb);
    float cosTheta = (float)(1.0 - u1);
    float theta = acosf(cosTheta);
    float phi = (float)(u2 * 2.0 * M_PI_REF);
    return cosTheta * D + sin(theta) * (cos(phi) * ta + sin(phi) * tb);
}
static f3 hemisphereSampleUniformCone(f3 D, float maxAngle, float u1, float u2) {  // :429-444
    f3 ta, tb;
    frame(D, &ta, &tb);
    float theta = (float)acos(1.0 - u1 * (1.0 - cos(maxAngle)));
    float phi = (float)(u2 * 2.0 * M_PI_REF);
    return cos(theta) * D + sin(theta) * (cos(phi) * ta + sin(phi) * tb);
}
static f3 sphereSampleUniform(float u1, float u2) {  // :556-566
    float theta = (float)acos(1.0 - (2.0 * u1));
    float phi = (float)(u2 * 2.0 * M_PI_REF);
    return cos(theta) * mk(0, 1, 0) + sin(theta) * (cos(phi) * mk(1, 0, 0) + sin(phi) * mk(0, 0, 1));
}
static f3 sphereSampleBiased3(f3 D, float a, float u1, float u2, float* rpdf) {  // :501-523
    f3 ta, tb;
    frame(D, &ta, &tb);
    float phi = (float)(u1 * 2.0 * M_PI_REF);
    float theta = (float)(0.5 * M_PI_REF - pow(pow(0.5 * M_PI_REF, a + 1) * u2, 1.0 / (a + 1.0)));
    *rpdf = (float)(1.0 / (((a + 1) / (4.0 * pow(0.5 * M_PI_REF, a + 2))) * pow(0.5 * M_PI_REF - theta, a) / sin(theta)));
    return cos(theta) * D + sin(theta) * (cos(phi) * ta + sin(phi) * tb);
}
static bool solveQuadraticEquation(float a, float b, float c, float* s1, float* s2) {  // :569-578
    float disc = (float)(pow(b, 2) - (4 * a * c));
    if (disc < 0) return false;
    if (s1) *s1 = (float)(((-1 * b) + sqrt(disc)) / (2 * a));
    if (s2) *s2 = (float)(((-1 * b) - sqrt(disc)) / (2 * a));
    return true;
}
static float surfaceArea_to_solidAngle(f3 A, f3 B, f3 N) {  // :611-616
    f3 AtoB = B - A;
    return (float)(fabs(dot(normalized(AtoB), N)) / pow(mag(AtoB), 2));
}

// wa_camera::generateRayForPixel (src/fundamentals.cpp:306-327)
static void camera_ray(const Camera& c, int i, int j, float u1, float u2, f3* P, f3* D) {
    if (c.type == 0) {
        f3 d = c.sensorOrigin + c.pixelWidth * ((i + u1) * c.side + (j + u2) * c.up) - c.P;
        *P = c.P;
        *D = normalized(d);
    } else {
        *P = c.P + (c.pixelWidth * (i + u1) - 0.5 * c.sensorWidth) * c.side + (c.pixelWidth * (j + u2) - 0.5 * c.sensorWidth) * c.up;
        *D = c.D;
    }
}

// ------------------------------------------------------------------ wa_shader::refract (src/objects.cpp:66-107)

static f3 refract(const Shader& sh, f3 Ng, f3 D, float* fresnel) {
    float NdotD = dot(Ng, D);
    f3 tv = D - NdotD * Ng;
    float tm = mag(tv);
    // the reference normalises a zero vector at exactly normal incidence (NaN path); kept finite here and in the library
    f3 tangent = tm > 0.f ? mk(tv.x / tm, tv.y / tm, tv.z / tm) : mk(0, 0, 0);
    float cosTheta1 = NdotD;
    float sinTheta1 = (float)sqrt(std::max(0.0, 1.0 - pow(NdotD, 2)));
    float sinTheta2, cosTheta2;
    float ior = sh.ior;
    if (NdotD > 0.0) {
        sinTheta2 = sinTheta1 / ior;
        cosTheta2 = (float)sqrt(1.0 - pow(sinTheta2, 2));
        *fresnel = (float)(0.5 * (pow((ior * cosTheta1 - cosTheta2) / (ior * cosTheta1 + cosTheta2), 2) +
                                  pow((cosTheta1 - ior * cosTheta2) / (cosTheta1 + ior * cosTheta2), 2)));
        if (std::isnan(*fresnel)) *fresnel = 0.0;
        return sinTheta2 * tangent + cosTheta2 * Ng;
    }
    sinTheta2 = sinTheta1 * ior;
    if (sinTheta2 > 1.0) {
        *fresnel = 1.0;
        return mk(0, 0, 0);
    }
    cosTheta2 = (float)sqrt(1.0 - pow(sinTheta2, 2));
    cosTheta1 *= -1.0;
    *fresnel = (float)(0.5 * (pow((cosTheta1 - ior * cosTheta2) / (cosTheta1 + ior * cosTheta2), 2) +
                              pow((ior * -cosTheta2) / (ior * cosTheta1 + cosTheta2), 2)));  // sic, objects.cpp:101
    if (std::isnan(*fresnel)) *fresnel = 0.0;
    return sinTheta2 * tangent - 1.0 * cosTheta2 * Ng;
}

// ------------------------------------------------------------------ lights (src/lights.cpp)

static bool radianceSample(const Light& L, f3 SP, f3 SN, bool hasN, float u1, float u2, col* Li, f3* LP, float* rpdf,
                           float traceBias) {
    if (L.type == 0) {  // :84-101
        *LP = L.P + u1 * L.U + u2 * L.V;
        if (dot(SP - L.P, L.N) < traceBias) return false;
        *Li = L.radiance;
        *rpdf = L.area * surfaceArea_to_solidAngle(SP, *LP, L.N);
        return true;
    }
    // :217-246
    f3 SL = L.P - SP;
    float SLdist = mag(SL);
    f3 SLn = normalized(SL);
    float coneAngle = asinf(L.rad / SLdist);
    if (SLdist < (L.rad + traceBias)) return false;
    f3 D = hemisphereSampleUniformCone(SLn, coneAngle, u1, u2);
    if (hasN && dot(SN, D) <= 0.0) return false;
    float dist;
    if (!solveQuadraticEquation(1.0f, (float)(-2.0 * dot(SL, D)), (float)(pow(mag(SL), 2) - pow(L.rad, 2)), nullptr, &dist))
        dist = (float)(mag(SL) * cos(coneAngle));
    *LP = SP + dist * D;
    *Li = L.radiance;
    *rpdf = (float)(2.0 * M_PI_REF * (1.0 - cos(coneAngle)));
    return true;
}
static float light_pdf(const Light& L, f3 P, f3 SP) {
    if (L.type == 0) return (float)(1.0 / (L.area * surfaceArea_to_solidAngle(SP, P, L.N)));  // :106-109
    f3 SL = P - SP;  // :251-257 (the parameter shadows the light's centre)
    float coneAngle = asinf(L.rad / mag(SL));
    return (float)(1.0 / (2.0 * M_PI_REF * (1.0 - cos(coneAngle))));
}
static f3 light_centre(const Light& L) { return L.type == 0 ? L.P + 0.5 * (L.U + L.V) : L.P; }  // :36-39, :18-21

// ------------------------------------------------------------------ udp_renderer (src/unidirectionalPathTracer.cpp)

#define TRACE_BIAS 0.001
#define LOW_VISIBILITY_THRESHOLD 0.01
#define ONE_OVER_4_PI 0.07958
#define REFL_OR_TRANS_BLEND_FACTOR 0.75
enum { CAMERA_RAY = 0, DIFFUSE_RAY = 1, SPECULAR_RAY = 2 };

struct Renderer {
    const oracle_scene* sc;
    int hs, ls, td;
    float me, mr;
    uint32_t pixel;

    col volumeAttenuation(const Shader& medium, f3 A, f3 B) const {  // :407-427
        if (medium.isClear()) return C(1.0f);
        float l = mag(B - A);
        return cexp(col{(float)(-1.0 * medium.st.r * l), (float)(-1.0 * medium.st.g * l), (float)(-1.0 * medium.st.b * l)});
    }

    col visibility(f3 A, f3 B) const {  // :693-790
        float AB_l = mag(B - A);
        if (!(AB_l > 0.f)) return C(0.f);  // coincident end points: nothing to walk
        f3 ABn = normalized(B - A);
        float d = 0.0;
        col V = C(1.0f);
        f3 P0 = A, P1;
        const Shader* mediumCurrent = &sc->global;
        const Shader* shader = nullptr;
        bool enteringSurface = false;
        Intr intr;
        for (int it = 0; it < 4096; it++) {
            tl_shadow_rays++;
            bool hit = traceFAST(sc, P0, ABn, &intr);
            if (hit) d += mag(intr.P - P0);
            else d = (float)(2.0 * AB_l);
            if (d <= (AB_l + TRACE_BIAS)) {
                P1 = intr.P;
                shader = &sc->shaders[intr.objID];
                enteringSurface = dot(ABn, intr.N) < 0.0;
                if (enteringSurface) {
                    mediumCurrent = &sc->global;
                    if ((shader->flags & (F_DIELECTRIC | F_DIELECTRICSSS)) && shader->ior > 1.0) return C(0.f);
                } else
                    mediumCurrent = shader;
            } else {
                P1 = B;
                d = AB_l;
            }
            V *= volumeAttenuation(*mediumCurrent, P0, P1);
            if (d >= (AB_l - TRACE_BIAS)) return V;
            if (shader->isOpaque()) return C(0.f);
            if (V.r < LOW_VISIBILITY_THRESHOLD && V.g < LOW_VISIBILITY_THRESHOLD && V.b < LOW_VISIBILITY_THRESHOLD) return C(0.f);
            P0 = P1;
            mediumCurrent = enteringSurface ? shader : &sc->global;
        }
        return C(0.f);
    }

    col directLightHit(f3 P0, f3 D0, int lightID, const Shader& medium, f3* LP) const {  // :670-687
        Intr intr0;
        bool hit = traceFAST(sc, P0, -1.0 * D0, &intr0);
        if (!hit || intr0.objID != lightID) return C(0.f);
        *LP = intr0.P;
        return Le(intr0) * volumeAttenuation(medium, P0, intr0.P);
    }

    col directLightingSingleScatter(f3 P0, f3 D0, int objID, int lightID, const Shader& medium0) const {  // :631-665
        Intr intr0, intr1;
        bool hit = traceFAST(sc, P0, -1.0 * D0, &intr0);
        if (!hit || intr0.objID != objID) return C(0.f);
        float fresnel;
        f3 D1 = refract(medium0, intr0.N, D0, &fresnel);
        if (is0(D1)) return C(0.f);
        hit = traceFAST(sc, intr0.P, -1.0 * D1, &intr1);
        if (!hit || intr1.objID != lightID) return C(0.f);
        return Le(intr1) * volumeAttenuation(sc->global, intr0.P, intr1.P) * (float)(1.0 - fresnel) *
               volumeAttenuation(medium0, P0, intr0.P);
    }

    col directReflection(const Shader& shader, f3 P, f3 N, uint32_t stream) const {  // :345-381
        col Lo = C(0.f);
        int nl = (int)sc->lights.size();
        float rls = (float)(1.0 / ls);
        for (int l = 0; l < nl; l++)
            for (int i = 0; i < ls; i++) {
                uint32_t dim = DIM_NEE + 2u * (uint32_t)(l * ls + i);
                col Li;
                f3 LP;
                float rpdf;
                if (!radianceSample(sc->lights[l], P, N, true, rnd(stream, dim), rnd(stream, dim + 1), &Li, &LP, &rpdf, (float)TRACE_BIAS))
                    continue;
                f3 Wi = normalized(LP - P);
                float NdotWi = dot(Wi, N);
                if (NdotWi < 0.0) continue;
                col vis = visibility(P, LP);
                log_event(2.f, (float)l, LP, vis);
                if (iszero(vis)) continue;
                float scalar = (float)(rls * (float)(1.0 / M_PI_REF) * NdotWi * rpdf);
                Lo += shader.kd * vis * Li * scalar;
            }
        return Lo;
    }

    col indirectReflection(const Shader& shader, f3 P, f3 N, int depth, const Shader& medium, uint32_t key, uint32_t stream) const {  // :383-405
        col Lo = C(0.f);
        float rhs = (float)(1.0 / hs);
        for (int i = 0; i < hs; i++) {
            f3 Wi = hemisphereSampleUniform(N, rnd(stream, DIM_HEMI + 2u * i), rnd(stream, DIM_HEMI + 2u * i + 1u));
            float rpdf = (float)(2.0 * M_PI_REF);
            if (dot(Wi, N) < 0.0) continue;
            col Li = radiance(P, -1.0 * Wi, depth, false, DIFFUSE_RAY, medium, key_child(key, SLOT_HEMI | (uint32_t)i));
            float scalar = (float)(rhs * (float)(1.0 / M_PI_REF) * dot(N, Wi) * rpdf);
            Lo += shader.kd * Li * scalar;
        }
        return Lo;
    }

    col transmission(const Shader& surf, f3 P, f3 Ng, f3 D, int depth, bool includeEmission, const Shader& medium, uint32_t key) const {  // :296-311
        if (iszero(surf.kt)) return C(0.f);
        float fresnel;
        f3 D2 = refract(surf, Ng, D, &fresnel);
        if (is0(D2)) return C(0.f);
        return (float)(1.0 - fresnel) * radiance(P, D2, depth, includeEmission, CAMERA_RAY, medium, key_child(key, SLOT_TRANS));
    }

    col reflection(const Shader& shader, f3 P, f3 N, f3 Ng, f3 D, int depth, int rayType, bool includeEmission,
                   bool ignoreReflEvent, float reflEventProb, const Shader& medium, uint32_t key, uint32_t stream) const {  // :313-343
        col Lo_diffuse = C(0.f), Lo_specular = C(0.f);
        if (shader.flags & F_DIFFUSE) {
            Lo_diffuse = directReflection(shader, P, N, stream);
            if (!ignoreReflEvent && (depth + 1) <= td)
                Lo_diffuse += (float)(1.0 / reflEventProb) * indirectReflection(shader, P, N, depth + 1, medium, key, stream);
            return Lo_diffuse;  // a diffuse shader returns only this (:335)
        }
        if ((shader.flags & (F_SPECULAR | F_DIELECTRIC | F_DIELECTRICSSS)) && rayType != DIFFUSE_RAY) {
            if (!ignoreReflEvent && (depth + 1) <= td) {
                f3 R = -1.0 * D + (2.0 * dot(D, N)) * N;  // reflectVector :802-805
                Lo_specular += (float)(1.0 / reflEventProb) * (shader.ks * radiance(P, -1.0 * R, depth + 1, includeEmission, SPECULAR_RAY, medium, key_child(key, SLOT_SPEC)));
            }
        }
        if (shader.flags & F_SPECULAR) return Lo_specular;
        if (shader.flags & (F_DIELECTRIC | F_DIELECTRICSSS)) {
            float fresnel;
            refract(shader, Ng, D, &fresnel);
            return fresnel * Lo_specular;
        }
        return C(0.f);
    }

    col volumeScattering(int objID, const Shader& medium, f3 A, f3 B, int depth, bool doVolEvent, float volEventProb,
                         uint32_t key, uint32_t stream) const {  // :429-621
        col Lo = C(0.f);
        if (medium.isClear() || iszero(medium.ss)) return C(0.f);
        f3 IB = B - A;
        float IB_l = mag(IB);
        if (!(IB_l > 0.f)) return C(0.f);
        f3 IBn = normalized(IB);
        int RMS = (int)(IB_l * mr / ((float)depth + 1)) + 1;
        float rRMS = (float)(1.0 / (float)RMS);
        int nl = (int)sc->lights.size();
        float rls = (float)(1.0 / ls);
        const float phase = (float)(1.0 / (4.0 * M_PI_REF));  // wa_shader::phase, objects.cpp:61-64
        for (int s = 0; s < RMS; s++) {
            float random = rnd(stream, DIM_MARCH + (uint32_t)s);
            float RMD, rRMLpdf;
            if (me == 0.0) {  // sampling.cpp:75-81
                rRMLpdf = IB_l;
                RMD = random * IB_l;
            } else {  // sampling.cpp:91-100
                RMD = (float)((-1.0 / me) * log(1.0 - ((1.0 - exp(-1.0 * me * IB_l)) * random)));
                rRMLpdf = (float)(1.0 / ((me / (1.0 - exp(-1.0 * me * IB_l))) * exp(-1.0 * me * RMD)));
            }
            f3 S = A + RMD * IBn;
            col S_Lo = C(0.f);
            if (objID == -1) {
                for (int l = 0; l < nl; l++)
                    for (int i = 0; i < ls; i++) {
                        uint32_t dim = DIM_VNEE + 2u * (uint32_t)((s * nl + l) * ls + i);
                        col Li;
                        f3 LP;
                        float rpdf;
                        if (radianceSample(sc->lights[l], S, mk(0, 0, 0), false, rnd(stream, dim), rnd(stream, dim + 1), &Li, &LP, &rpdf, (float)TRACE_BIAS)) {
                            float Lpdf = (float)(1.0 / rpdf);
                            col vis = visibility(S, LP);
                            col c0 = vis * rpdf * Li;
                            float MIS = (float)(Lpdf / (Lpdf + ONE_OVER_4_PI));
                            if (rpdf == 0.0) MIS = 1.0;
                            c0 = c0 * MIS;
                            c0 *= medium.ss * phase;
                            c0 = c0 * rls;
                            S_Lo += c0;
                        }
                    }
                for (int l = 0; l < nl; l++) {
                    uint32_t dim = DIM_VPROBE + 2u * (uint32_t)(s * nl + l);
                    f3 Wi = -1.0 * sphereSampleUniform(rnd(stream, dim), rnd(stream, dim + 1));
                    float rpdf = (float)(4.0 * M_PI_REF);
                    f3 LP;
                    col Li = directLightHit(S, Wi, l, medium, &LP);
                    if (iszero(Li)) continue;
                    col c0 = Li * rpdf;
                    float altPdf = light_pdf(sc->lights[l], LP, S);
                    float MIS = (float)((ONE_OVER_4_PI) / (ONE_OVER_4_PI + altPdf));
                    if (std::isnan(MIS)) MIS = 1.0;
                    c0 = c0 * MIS;
                    c0 *= medium.ss * phase;
                    S_Lo += c0;
                }
            } else if (medium.objID >= 0) {
                // (the reference passes medium.objID here; for the global medium that field is never set and the
                //  probes below can only return black, so they are skipped)
                for (int l = 0; l < nl; l++)
                    for (int i = 0; i < 16; i++) {
                        uint32_t dim = DIM_VSINGLE + 2u * (uint32_t)((s * nl + l) * 16 + i);
                        f3 LPc = light_centre(sc->lights[l]);
                        float rpdf;
                        f3 Wi = -1.0 * sphereSampleBiased3(normalized(LPc - S), 1.5f, rnd(stream, dim), rnd(stream, dim + 1), &rpdf);
                        col Li = directLightingSingleScatter(S, Wi, medium.objID, l, medium);
                        if (iszero(Li)) continue;
                        col c0 = Li * (float)(1.0 / 16.0) * rpdf;
                        c0 *= medium.ss * phase;
                        S_Lo += c0;
                    }
            }
            if (doVolEvent && (depth + 1) <= td) {
                uint32_t dim = DIM_VCHILD + 2u * (uint32_t)s;
                f3 Wi = -1.0 * sphereSampleUniform(rnd(stream, dim), rnd(stream, dim + 1));
                col Li = radiance(S, Wi, depth + 1, false, CAMERA_RAY, medium, key_child(key, SLOT_VOL | (uint32_t)s));
                col c0 = Li * (float)(4.0 * M_PI_REF);
                c0 *= medium.ss * phase;
                S_Lo += (float)(1.0 / volEventProb) * c0;
            }
            S_Lo *= cexp(col{(float)(-1.0 * medium.st.r * RMD), (float)(-1.0 * medium.st.g * RMD), (float)(-1.0 * medium.st.b * RMD)});
            S_Lo = S_Lo * (rRMLpdf * rRMS);
            Lo += S_Lo;
        }
        return Lo;
    }

    col radiance(f3 P0, f3 D, int depth, bool includeEmission, int rayType, const Shader& medium0, uint32_t key) const {  // :198-293
        uint32_t stream = stream_of(key, pixel);
        f3 _D = -1.0 * D;
        if (!(dot(_D, _D) > 0.25f && dot(_D, _D) < 4.0f)) return C(0.f);  // NaN directions die here, as in the library
        Intr intr;
        bool hit = traceFAST(sc, P0, _D, &intr);
        if (!hit) {
            if (mag(P0 - sc->wc) >= sc->wr) return C(0.f);
            float lambda = 0.f;
            solveQuadraticEquation(1.0f, (float)(2.0 * dot(P0 - sc->wc, _D)),
                                   (float)(pow(mag(P0 - sc->wc), 2) - pow(sc->wr, 2)), &lambda, nullptr);
            f3 P1 = P0 + lambda * _D;
            return volumeScattering(-1, sc->global, P0, P1, depth, true, 1.0f, key, stream);
        }
        const Shader& shader = sc->shaders[intr.objID];
        log_event(1.f, (float)depth, intr.P, col{(float)intr.objID, intr.N.x + 10.f * intr.N.y + 100.f * intr.N.z, 0.f});
        f3 N1f;
        const Shader* medium1;
        int objID;
        if (dot(intr.N, D) > 0.0) {
            N1f = intr.N;
            medium1 = &shader;
            objID = -1;
        } else {
            N1f = -1.0 * intr.N;
            medium1 = &sc->global;
            objID = intr.objID;
        }
        float pS = (float)(1.0 / (float)(0.125 * depth + 1));
        bool doSurfaceEvent = rnd(stream, DIM_SURF) < pS;
        float pV = (float)(1.0 / (float)(0.04 * depth + 1));
        bool doVolumeEvent = rnd(stream, DIM_VOL) < pV;
        col Lo = C(0.f);
        if (doSurfaceEvent) {
            if (includeEmission) Lo += Le(intr);
            float fresnel;
            refract(shader, intr.N, D, &fresnel);
            bool reflOrTransEvent = false;
            float prob = 1.0;
            bool diel = shader.flags & (F_DIELECTRIC | F_DIELECTRICSSS);
            if (diel) prob = (float)((1.0 - REFL_OR_TRANS_BLEND_FACTOR) * 0.5 + REFL_OR_TRANS_BLEND_FACTOR * fresnel);
            if (diel && (rnd(stream, DIM_DIEL) < (1.0 - prob))) reflOrTransEvent = true;
            if (reflOrTransEvent && (depth + 1) <= td && !iszero(shader.kt))
                Lo += (float)(1.0 / (1.0 - prob)) * transmission(shader, intr.P, intr.N, D, depth + 1, includeEmission, *medium1, key);
            Lo += reflection(shader, intr.P, N1f, intr.N, D, depth, rayType, includeEmission, reflOrTransEvent, prob, medium0, key, stream);
            Lo *= volumeAttenuation(medium0, P0, intr.P);
            Lo = Lo * (float)(1.0 / pS);
        }
        Lo += volumeScattering(objID, medium0, P0, intr.P, depth, doVolumeEvent, pV, key, stream);
        return Lo;
    }
};

// ------------------------------------------------------------------ C interface for the tests

extern "C" {

oracle_scene* oracle_scene_create(const OSceneDesc* d) {
    if (!d || d->n_tris == 0) return nullptr;
    auto s = new oracle_scene();
    s->tris.resize(d->n_tris);
    for (uint64_t i = 0; i < d->n_tris; i++) {
        const float* v = d->tri_verts + 9 * i;
        Tri& T = s->tris[i];
        for (int a = 0; a < 3; a++) {
            T.v0[a] = v[a];
            T.e1[a] = (double)v[3 + a] - (double)v[a];
            T.e2[a] = (double)v[6 + a] - (double)v[a];
        }
        // Ng = (v1-v0) x (v2-v0) evaluated in float, as an intersector working on float vertices reports it
        f3 e1 = mk(v[3] - v[0], v[4] - v[1], v[5] - v[2]), e2 = mk(v[6] - v[0], v[7] - v[1], v[8] - v[2]);
        f3 n = cross(e1, e2);
        T.ng[0] = n.x;
        T.ng[1] = n.y;
        T.ng[2] = n.z;
        T.obj = d->tri_obj[i];
        T.prim = d->tri_prim ? d->tri_prim[i] : (int32_t)i;
    }
    build_bvh(s);
    for (int i = 0; i < d->n_objects; i++) s->shaders.push_back(make_shader(d->materials[i], i));
    s->global = make_shader(d->global_medium, -2);
    for (int i = 0; i < d->n_lights; i++) s->lights.push_back(make_light(d->lights[i]));
    s->cam = make_camera(d->camera);
    s->wc = ld(d->world_center);
    s->wr = d->world_radius;
    return s;
}

void oracle_scene_destroy(oracle_scene* s) { delete s; }

// mode 0: brute force over all triangles, 1: through the binary BVH
int oracle_intersect(const oracle_scene* s, const ORay* rays, uint32_t n, OHit* hits, int mode, int threads) {
    if (!s || !rays || !hits) return 1;
    threads = std::max(1, threads);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=]() {
            for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)threads) {
                double tt;
                int h = closest_hit(s, rays[i].org, rays[i].dir, rays[i].tnear, rays[i].tfar, &tt, mode == 0);
                OHit& H = hits[i];
                H.t = (float)tt;
                if (h < 0) {
                    H.prim = H.obj = -1;
                    H.Ng[0] = H.Ng[1] = H.Ng[2] = 0.f;
                } else {
                    H.prim = s->tris[h].prim;
                    H.obj = s->tris[h].obj;
                    memcpy(H.Ng, s->tris[h].ng, 12);
                }
            }
        });
    for (auto& x : th) x.join();
    return 0;
}

// udp_renderer::render / renderTile (src/unidirectionalPathTracer.cpp:78-176) with `threads` workers
// striding over columns like the reference's four.  fb receives mean radiance; *rays_out the number
// of closest-hit queries.
int oracle_render(const oracle_scene* s, const OParams* p, float* fb, int threads, uint64_t* rays_out) {
    if (!s || !p || !fb || p->psamples <= 0 || p->hsamples <= 0 || p->lsamples <= 0) return 1;
    threads = std::max(1, threads);
    int W = s->cam.resI, H = s->cam.resJ;
    std::atomic<uint64_t> rays{0};
    g_shadow_rays = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t]() {
            tl_rays = 0;
            tl_shadow_rays = 0;
            // profile bit 2: every number from this thread's own generator, in call order (see the header)
            std::mt19937_64 sequential(p->seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull * (uint64_t)(t + 1));
            tl_sequential = (p->profile & 4) ? &sequential : nullptr;
            Renderer R;
            R.sc = s;
            R.hs = p->hsamples;
            R.ls = p->lsamples;
            R.td = p->max_depth;
            R.me = p->me;
            R.mr = p->mr;
            for (int j = 0; j < H; j++)
                for (int i = t; i < W; i += threads) {
                    col pixelValue = C(0.f);
                    R.pixel = (uint32_t)(j * W + i);
                    for (int sm = 0; sm < p->sample_count; sm++) {
                        uint32_t key = key_root(p->seed, (uint32_t)(p->sample_begin + sm));
                        uint32_t stream = stream_of(key, R.pixel);
                        f3 P, D;
                        camera_ray(s->cam, i, j, rnd(stream, DIM_CAM0), rnd(stream, DIM_CAM1), &P, &D);
                        col v = R.radiance(P, -1.0 * D, 0, true, CAMERA_RAY, s->global, key) * (float)(1.0 / (float)p->psamples);
                        pixelValue += v;
                    }
                    fb[(j * W + i) * 3 + 0] = pixelValue.r;
                    fb[(j * W + i) * 3 + 1] = pixelValue.g;
                    fb[(j * W + i) * 3 + 2] = pixelValue.b;
                }
            rays += tl_rays;
            g_shadow_rays += tl_shadow_rays;
            tl_sequential = nullptr;
        });
    for (auto& x : th) x.join();
    if (rays_out) *rays_out = rays.load();
    return 0;
}

// Event log of the path of one pixel sample: records of 8 floats (kind, depth|light, x, y, z, r, g, b);
// kind 1 = surface hit (r = object), kind 2 = light sample after its shadow walk (rgb = visibility).
int oracle_debug_path(const oracle_scene* s, const OParams* p, int i, int j, int sample, float* out, int max_records) {
    std::vector<float> log;
    tl_log = &log;
    Renderer R;
    R.sc = s; R.hs = p->hsamples; R.ls = p->lsamples; R.td = p->max_depth; R.me = p->me; R.mr = p->mr;
    R.pixel = (uint32_t)(j * s->cam.resI + i);
    uint32_t key = key_root(p->seed, (uint32_t)sample);
    uint32_t stream = stream_of(key, R.pixel);
    f3 P, D;
    camera_ray(s->cam, i, j, rnd(stream, DIM_CAM0), rnd(stream, DIM_CAM1), &P, &D);
    col v = R.radiance(P, -1.0 * D, 0, true, CAMERA_RAY, s->global, key);
    tl_log = nullptr;
    log_event(0, 0, P, v);
    int n = (int)std::min<size_t>(log.size() / 8, (size_t)max_records);
    memcpy(out, log.data(), (size_t)n * 8 * sizeof(float));
    return n;
}

// closest-hit queries that the shadow walks of the last oracle_render issued
uint64_t oracle_last_shadow_rays(void) { return g_shadow_rays.load(); }

// ---- single functions, for known-answer tests

float oracle_rand(uint64_t seed, uint32_t sample, uint32_t pixel, uint32_t dim) {
    return rnd(stream_of(key_root(seed, sample), pixel), dim);
}
uint32_t oracle_key_child(uint32_t key, uint32_t slot) { return key_child(key, slot); }

void oracle_refract(const float* Ng, const float* D, float ior, float* out_dir, float* fresnel) {
    Shader sh;
    memset(&sh, 0, sizeof(sh));
    sh.ior = ior;
    f3 r = refract(sh, ld(Ng), ld(D), fresnel);
    out_dir[0] = r.x;
    out_dir[1] = r.y;
    out_dir[2] = r.z;
}

void oracle_camera_ray(const OCamera* c, int i, int j, float u1, float u2, float* P, float* D) {
    Camera k = make_camera(*c);
    f3 p, d;
    camera_ray(k, i, j, u1, u2, &p, &d);
    P[0] = p.x; P[1] = p.y; P[2] = p.z;
    D[0] = d.x; D[1] = d.y; D[2] = d.z;
}

int oracle_light_sample(const OLight* l, const float* SP, const float* SN, int hasN, float u1, float u2, float* Li,
                        float* LP, float* rpdf) {
    Light L = make_light(*l);
    col li = C(0.f);
    f3 lp = mk(0, 0, 0);
    bool ok = radianceSample(L, ld(SP), ld(SN), hasN != 0, u1, u2, &li, &lp, rpdf, (float)TRACE_BIAS);
    Li[0] = li.r; Li[1] = li.g; Li[2] = li.b;
    LP[0] = lp.x; LP[1] = lp.y; LP[2] = lp.z;
    return ok ? 1 : 0;
}

float oracle_light_pdf(const OLight* l, const float* P, const float* SP) { return light_pdf(make_light(*l), ld(P), ld(SP)); }

void oracle_sample_dirs(int which, const float* D, float a, float u1, float u2, float* out, float* rpdf) {
    f3 r;
    *rpdf = 0.f;
    switch (which) {
        case 0: r = hemisphereSampleUniform(ld(D), u1, u2); *rpdf = (float)(2.0 * M_PI_REF); break;
        case 1: r = hemisphereSampleUniformCone(ld(D), a, u1, u2); break;
        case 2: r = sphereSampleUniform(u1, u2); *rpdf = (float)(4.0 * M_PI_REF); break;
        default: r = sphereSampleBiased3(ld(D), a, u1, u2, rpdf); break;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

}  // extern "C"
