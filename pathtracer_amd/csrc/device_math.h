// Scalar building blocks of the radiance loop in fp32: vectors, the camera,
// direction samplers, Snell/Fresnel and the two light types.  Each function
// names the reference routine whose distribution or value it reproduces; the
// reference evaluates many of these in double because of its literals, here
// everything is single precision (the image tolerance is stated in the tests).
#pragma once
#include <math.h>
#include <stdint.h>

#include "rng.h"

#ifndef WAPT_HW_SINCOS
#define WAPT_HW_SINCOS 1
#endif

#define WAPT_PI 3.1415926535f          // M_PI as the reference defines it, src/fundamentals.h:12
#define WAPT_TRACE_BIAS 0.001f         // TRACE_BIAS, src/unidirectionalPathTracer.cpp:13
#define WAPT_TFAR 10000000.0f          // tfar of every query, src/content.cpp:95
#define WAPT_LOW_VIS 0.01f             // LOW_VISIBILITY_THRESHOLD, cpp:15
#ifndef WAPT_FAST_FACTORS
#define WAPT_FAST_FACTORS 1
#endif
#define WAPT_MIS_CONST 0.07958f        // ONE_OVER_4_PI as written, cpp:16
#define WAPT_BLEND 0.75f               // REFL_OR_TRANS_BLEND_FACTOR, cpp:17
#define WAPT_PHASE (1.0f / (4.0f * WAPT_PI))  // wa_shader::phase, src/objects.cpp:61-64

struct V3 {
    float x, y, z;
};
WAPT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
WAPT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
WAPT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
WAPT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
WAPT_HD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
WAPT_HD V3 operator*(V3 a, float s) { return V3{s * a.x, s * a.y, s * a.z}; }
WAPT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
WAPT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
WAPT_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Correctly rounded square root of a value that is zero or of ordinary magnitude (lengths, sines, discriminants of
// scene-scale quantities): the hardware's estimate moved by one ulp where its exact residual says so - the
// correction steps of the compiler's sqrtf sequence without its rescaling of denormal arguments (9 instructions
// instead of 16).  Zero, infinity and NaN come out as sqrtf gives them.
WAPT_HD float wsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const uint32_t b = __float_as_uint(s);
    const float dn = __uint_as_float(b - 1u), up = __uint_as_float(b + 1u);
    float r = (0.0f >= fmaf(-dn, s, x)) ? dn : s;
    r = (0.0f < fmaf(-up, s, x)) ? up : r;
    return r;
#else
    return sqrtf(x);
#endif
}
WAPT_HD float mag(V3 a) { return wsqrt(dot(a, a)); }
WAPT_HD V3 normalized_len(V3 a, float* len);
// Three divisions by the same length, as the reference has them, and rounded as a division rounds: one refined
// reciprocal for all three, then per component the steps of the hardware's division sequence (product, two
// corrections from the exact residual) without its range scaling - the divisor is a length of scene scale.
// 18 instructions instead of 30: the surface kernel is 4 % faster, the same-seed agreement with the CPU checker
// is unchanged scene by scene.  (One reciprocal and three plain products would be cheaper still and move the
// directions by an ulp: that alone lowered the agreement from 99.74 to 99.52 %.)
WAPT_HD V3 normalized(V3 a) {
    float m;
    return normalized_len(a, &m);
}
// normalized(a) and the length it divides by
WAPT_HD V3 normalized_len(V3 a, float* len) {
    float m = mag(a);
    *len = m;
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(m);
    r = fmaf(fmaf(-m, r, 1.0f), r, r);
    float qx = a.x * r, qy = a.y * r, qz = a.z * r;
    qx = fmaf(fmaf(-qx, m, a.x), r, qx);
    qy = fmaf(fmaf(-qy, m, a.y), r, qy);
    qz = fmaf(fmaf(-qz, m, a.z), r, qz);
    qx = fmaf(fmaf(-qx, m, a.x), r, qx);
    qy = fmaf(fmaf(-qy, m, a.y), r, qy);
    qz = fmaf(fmaf(-qz, m, a.z), r, qz);
    return V3{qx, qy, qz};
#else
    return V3{a.x / m, a.y / m, a.z / m};
#endif
}
WAPT_HD bool is_zero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
WAPT_HD V3 vexp(V3 a) { return V3{expf(a.x), expf(a.y), expf(a.z)}; }
WAPT_HD V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }

// ---- device-side scene records --------------------------------------------

// What the kernels read of a wa_shader (src/objects.h:13-41), in four 16-byte words; a surface event reads the first two.
struct __attribute__((aligned(16))) DMaterial {
    float kcol[3];        // the colour its surface event multiplies by: kd of a diffuse material, ks of any other
                          // (cpp:313-343: a diffuse shader never reads ks, a specular or dielectric one never kd)
    uint32_t flags;
    float ior;
    uint32_t kt_nonzero;  // kt != 0 (transmission(), cpp:298)
    uint32_t scatters;    // !isClear && ss != 0: volumeScattering does work (cpp:454)
    uint32_t pad0;
    float st[3];
    float pad1;
    float ss[3];
    float pad2;
};  // 64 B

struct DLight {
    int32_t type;
    float P[3], U[3], V[3], N[3];
    float area, radius;
    float Li[3];      // radiance: area 0.5*power/area/pi, sphere power/(4 pi r^2)/pi
    float centre[3];  // getCentre()
    // the light object's triangles in the BVH (filled on the device after the tree is built): an area light's quad
    // is two, a sphere light has none (src/content.cpp:42-50 gives only area lights a geometry)
    int32_t ntri, tri0, tri1;
    // area lights: no other object's surface lies in front of the quad closer than a shadow search can resolve
    // (set at scene creation, see light_front_is_clear): a walk that ends on the quad may stop searching short of it
    int32_t front_clear;
    // number of this light among those that a probe ray (directLightHit, cpp:670-687) can hit, -1 for a light without
    // geometry: the probes of a march sample lie light by light in their queue
    int32_t probe_slot;
};

struct DCamera {
    int32_t type, resI, resJ, pad;
    float P[3], D[3], side[3], up[3], sensorOrigin[3];
    float pixelWidth, sensorWidth;
};

// ---- camera: wa_camera ctor / setResolution / generateRayForPixel ----------
// src/fundamentals.cpp:227-263, 306-327

WAPT_HD void camera_ray(const DCamera& c, int i, int j, float u1, float u2, V3* org, V3* dir) {
    V3 side = ld3(c.side), up = ld3(c.up), P = ld3(c.P);
    if (c.type == 0) {
        V3 d = ld3(c.sensorOrigin) + c.pixelWidth * (((float)i + u1) * side + ((float)j + u2) * up) - P;
        *org = P;
        *dir = normalized(d);
    } else {
        *org = P + (c.pixelWidth * ((float)i + u1) - 0.5f * c.sensorWidth) * side +
               (c.pixelWidth * ((float)j + u2) - 0.5f * c.sensorWidth) * up;
        *dir = ld3(c.D);
    }
}

// ---- direction samplers ------------------------------------------------------

// Tangent frame of hemisphereSampleUniform & co (src/fundamentals.cpp:412-414).
WAPT_HD void frame(V3 D, V3* ta, V3* tb) {
    V3 a = cross(D, v3(1.f, 0.f, 0.f));
    if (D.y == 0.f && D.z == 0.f) a = cross(D, v3(0.f, 1.f, 0.f));
    a = normalized(a);
    *ta = a;
    *tb = normalized(cross(a, D));
}

// sine and cosine of 2 pi u, u in [0, 1): the hardware's sine and cosine take their argument in turns and are good
// to about 1e-6 here; the same-seed agreement with the CPU checker is the same with them as with sincosf, scene by
// scene, and the surface kernel is 3 % faster.
WAPT_HD void sincos_turns(float u, float* s, float* c) {
#if defined(__HIP_DEVICE_COMPILE__) && WAPT_HW_SINCOS
    *s = __builtin_amdgcn_sinf(u);
    *c = __builtin_amdgcn_cosf(u);
#else
    sincosf(u * 2.0f * WAPT_PI, s, c);
#endif
}

WAPT_HD V3 polar(V3 D, float cosT, float sinT, float turns) {
    V3 ta, tb;
    frame(D, &ta, &tb);
    float s, c;
    sincos_turns(turns, &s, &c);
    return cosT * D + sinT * (c * ta + s * tb);
}

// hemisphereSampleUniform(D, &rpdf): cos(theta) = 1 - u1, rpdf = 2 pi (fundamentals.cpp:408-425)
WAPT_HD V3 hemisphere_uniform(V3 D, float u1, float u2) {
    float cosT = 1.0f - u1;
    float sinT = wsqrt(fmaxf(0.f, 1.0f - cosT * cosT));
    return polar(D, cosT, sinT, u2);
}

// 1 - cos(asin(s)): the reference evaluates 1.0 - cos(coneAngle) in double (lights.cpp:244, 256); in single
// precision 1 - sqrt(1 - s^2) loses everything for a small light far away (a point light is a sphere of radius 1:
// s = 0.002, 1 - cos = 2e-6 against an ulp of 6e-8), so the difference is written without the cancellation.
WAPT_HD float one_minus_cos_of_asin(float s) {
    float c = wsqrt(fmaxf(0.f, 1.0f - s * s));
    return s * s / (1.0f + c);
}

// hemisphereSampleUniform(D, maxAngle, ..): uniform in the cone (fundamentals.cpp:429-444), given 1 - cos(maxAngle).
// cos(theta) = 1 - x with x = u1 (1 - cos(maxAngle)), and sin^2(theta) = x (2 - x) keeps a narrow cone's few ulps.
WAPT_HD V3 cone_uniform(V3 D, float omcMax, float u1, float u2) {
    float x = u1 * omcMax;
    float cosT = 1.0f - x;
    float sinT = wsqrt(fmaxf(0.f, x * (2.0f - x)));
    return polar(D, cosT, sinT, u2);
}

// sphereSampleUniform(): polar axis y (fundamentals.cpp:556-566)
WAPT_HD V3 sphere_uniform(float u1, float u2) {
    float cosT = 1.0f - 2.0f * u1;
    float sinT = wsqrt(fmaxf(0.f, 1.0f - cosT * cosT));
    float s, c;
    sincos_turns(u2, &s, &c);
    return v3(sinT * c, cosT, sinT * s);
}

// sphereSampleBiased3(D, a, &rpdf) (fundamentals.cpp:501-523); u1 drives phi, u2 theta.
WAPT_HD V3 sphere_biased3(V3 D, float a, float u1, float u2, float* rpdf) {
    const float hp = 0.5f * WAPT_PI;
    float theta = hp - powf(powf(hp, a + 1.0f) * u2, 1.0f / (a + 1.0f));
    float sinT = sinf(theta), cosT = cosf(theta);
    *rpdf = 1.0f / (((a + 1.0f) / (4.0f * powf(hp, a + 2.0f))) * powf(hp - theta, a) / sinT);
    return polar(D, cosT, sinT, u1);
}

// solveQuadraticEquation (fundamentals.cpp:569-578)
WAPT_HD bool solve_quadratic(float a, float b, float c, float* s1, float* s2) {
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.f) return false;
    float r = wsqrt(disc);
    *s1 = (-b + r) / (2.0f * a);
    *s2 = (-b - r) / (2.0f * a);
    return true;
}

// ---- wa_shader::refract (src/objects.cpp:66-107) ------------------------------
// D points back along the arriving ray.  Returns the refracted direction in
// the same convention (zero vector on total internal reflection) and the
// Fresnel reflectance, including the asymmetric second term on the way out.
WAPT_HD V3 refract_dir(V3 Ng, V3 D, float ior, float* fresnel) {
    float NdotD = dot(Ng, D);
    // At exactly normal incidence the reference divides 0 by 0 here and the
    // path dies as NaN; the guards keep the straight-through direction instead.
    V3 tv = D - NdotD * Ng;
    float tm = mag(tv);
    V3 tangent = tm > 0.f ? (1.0f / tm) * tv : v3(0.f, 0.f, 0.f);
    float c1 = NdotD;
    float s1 = wsqrt(fmaxf(0.f, 1.0f - NdotD * NdotD));
    float f;
    if (NdotD > 0.0f) {
        float s2 = s1 / ior;
        float c2 = wsqrt(1.0f - s2 * s2);
        float a = (ior * c1 - c2) / (ior * c1 + c2);
        float b = (c1 - ior * c2) / (c1 + ior * c2);
        f = 0.5f * (a * a + b * b);
        if (f != f) f = 0.0f;
        *fresnel = f;
        return s2 * tangent + c2 * Ng;
    }
    float s2 = s1 * ior;
    if (s2 > 1.0f) {
        *fresnel = 1.0f;
        return v3(0.f, 0.f, 0.f);
    }
    float c2 = wsqrt(1.0f - s2 * s2);
    c1 = -c1;
    float a = (c1 - ior * c2) / (c1 + ior * c2);
    float b = (ior * -c2) / (ior * c1 + c2);
    f = 0.5f * (a * a + b * b);
    if (f != f) f = 0.0f;
    *fresnel = f;
    return s2 * tangent - c2 * Ng;
}

// reflectVector (cpp:802-805)
WAPT_HD V3 reflect_dir(V3 V, V3 N) { return -V + 2.0f * dot(V, N) * N; }

// ---- lights ----------------------------------------------------------------

// surfaceArea_to_solidAngle (fundamentals.cpp:611-616)
WAPT_HD float area_to_solid_angle(V3 A, V3 B, V3 N) {
    V3 ab = B - A;
#if defined(__HIP_DEVICE_COMPILE__) && WAPT_FAST_FACTORS
    // |cos| / d^2 = |ab . N| / d^3; the result is only ever a factor of a contribution, so the two divisions (ten
    // instructions each) become one reciprocal
    const float m2 = dot(ab, ab);
    return fabsf(dot(ab, N)) * __builtin_amdgcn_rcpf(m2 * wsqrt(m2));
#else
    float m = mag(ab);
    return fabsf(dot(ab, N) / m) / (m * m);
#endif
}

// wa_areaLight::radianceSample (lights.cpp:84-101) / wa_sphereLight::radianceSample (217-246).
// has_normal=false is the reference's "SN == 0" call from inside a medium.
WAPT_HD bool light_sample(const DLight& L, V3 SP, V3 SN, bool has_normal, float u1, float u2, V3* LP, float* rpdf) {
    if (L.type == 0) {
        V3 P = ld3(L.P), N = ld3(L.N);
        *LP = P + u1 * ld3(L.U) + u2 * ld3(L.V);
        if (dot(SP - P, N) < WAPT_TRACE_BIAS) return false;
        *rpdf = L.area * area_to_solid_angle(SP, *LP, N);
        return true;
    }
    V3 SL = ld3(L.P) - SP;
    float dist = mag(SL);
    if (dist < L.radius + WAPT_TRACE_BIAS) return false;
    V3 SLn = (1.0f / dist) * SL;
    float sinC = L.radius / dist;
    float cosC = wsqrt(fmaxf(0.f, 1.0f - sinC * sinC));
    float omc = one_minus_cos_of_asin(sinC);
    V3 D = cone_uniform(SLn, omc, u1, u2);
    if (has_normal && dot(SN, D) <= 0.0f) return false;
    float s1, s2;
    float d;
    if (solve_quadratic(1.0f, -2.0f * dot(SL, D), dist * dist - L.radius * L.radius, &s1, &s2))
        d = s2;
    else
        d = dist * cosC;
    *LP = SP + d * D;
    *rpdf = 2.0f * WAPT_PI * omc;
    return true;
}

// wa_areaLight::pdf (lights.cpp:106-109) / wa_sphereLight::pdf (251-257)
WAPT_HD float light_pdf(const DLight& L, V3 P, V3 SP) {
    if (L.type == 0) return 1.0f / (L.area * area_to_solid_angle(SP, P, ld3(L.N)));
    float dist = mag(P - SP);
    return 1.0f / (2.0f * WAPT_PI * one_minus_cos_of_asin(L.radius / dist));
}
