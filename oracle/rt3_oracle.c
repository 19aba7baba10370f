/* rt3_oracle.c — CPU restatement of the reference's render path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (raytracer-3_amd/) never does.  Plain C, scalar, one function per reference function.
 *
 * Parity pin: the reference ships no golden vectors and cannot be built in this image
 * without a stand-in for its un-vendored logging dependency (CppDebugger), so oracle/_ref is not built.
 * Mode R of this file is pinned by the reference outputs recorded in SURVEY.md §6/§8c (PPM SHA-256 of the
 * built-in scene at 400x225 and 1920x1080, two known pixels) — tests/test_oracle_pin.py.  Mode X has no
 * reference implementation anywhere: "parity unpinned" by the reference; it follows the
 * design intent of raytracer_v4.glsl / random_v1.glsl and the semantics written down in DESIGN.md §4.
 *
 * Build: gcc -O2 -ffp-contract=off -mavx2 -mfma -fopenmp (oracle/Makefile).  -ffp-contract=off keeps
 * every a*b+c unfused unless written fmaf(); -mfma only makes the explicit fmaf() a single instruction.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/rt3.h"

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 v3p(const float* p) { return V(p[0], p[1], p[2]); }

/* #define dot3 of SequentialRenderer.cpp:32-33 and glm::dot (glm/detail/func_geometric.inl:48-55):
 * x*x' + y*y' + z*z', left to right, unfused. */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* glm::cross, glm/detail/func_geometric.inl:68-79. */
static inline v3 cross3(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt (func_geometric.inl:82-90,
 * func_exponential.inl:134-139). */
static inline v3 normalize3(v3 v) { float inv = 1.0f / sqrtf(dot3(v, v)); return V(v.x * inv, v.y * inv, v.z * inv); }

/* ====================================================================================================
 * Pre-render
 * ================================================================================================== */
static void set_face(rt3_gface* f, uint32_t a, uint32_t b, uint32_t c, v3 n, v3 col) {
    memset(f, 0, sizeof(*f));
    f->v1 = a; f->v2 = b; f->v3 = c;
    f->normal[0] = n.x; f->normal[1] = n.y; f->normal[2] = n.z;
    f->color[0] = col.x; f->color[1] = col.y; f->color[2] = col.z;
}
static void set_vert(float* v, uint32_t i, v3 p) { v[4 * i] = p.x; v[4 * i + 1] = p.y; v[4 * i + 2] = p.z; v[4 * i + 3] = 0.0f; }

/* create_triangle + cpu_pre_render_triangle, Triangle.cpp:28-76: normal = normalize(cross(p3-p1, p2-p1)),
 * colour is NOT headlight-shaded. */
void oracle_prerender_triangle(const float* p1, const float* p2, const float* p3, const float* color,
                               rt3_gface* faces, float* verts) {
    v3 a = v3p(p1), b = v3p(p2), c = v3p(p3);
    v3 n = normalize3(cross3(vsub(c, a), vsub(b, a)));
    set_vert(verts, 0, a); set_vert(verts, 1, b); set_vert(verts, 2, c);
    set_face(&faces[0], 0, 1, 2, n, v3p(color));
}

uint32_t oracle_sphere_face_count(uint32_t m, uint32_t p) { return m + 2 * ((p - 3) * m) + m; }   /* Sphere.cpp:101 */
uint32_t oracle_sphere_vertex_count(uint32_t m, uint32_t p) { return 2 + (p - 2) * m; }           /* Sphere.cpp:102 */

/* compute_point, Sphere.cpp:69-79: double trig on (float ratio promoted), rounded to float per component,
 * then center + radius * vec3 in float. */
static v3 sphere_point(float fx, float fy, v3 center, float radius, uint32_t m, uint32_t p) {
    double ty = M_PI * (double)(fy / (float)(p - 1));
    double tx = 2 * M_PI * (double)(fx / (float)m);
    v3 unit = V((float)(sin(ty) * cos(tx)), (float)cos(ty), (float)(sin(ty) * sin(tx)));
    return vadd(center, vscale(radius, unit));
}
/* headlight shading colour * |dot(n,(0,0,-1))|, Sphere.cpp:155 / Object.cpp:194; glm::dot order. */
static v3 headlight(v3 color, v3 n) {
    float d = fabsf(n.x * 0.0f + n.y * 0.0f + n.z * -1.0f);
    return V(color.x * d, color.y * d, color.z * d);
}

/* cpu_pre_render_sphere, Sphere.cpp:120-261. */
void oracle_prerender_sphere(const float* center, float radius, uint32_t m, uint32_t p, const float* color,
                             rt3_gface* faces, float* verts) {
    v3 C = v3p(center), col = v3p(color);
    for (uint32_t y = 1; y < p; y++) {
        for (uint32_t x = 0; x < m; x++) {
            uint32_t xm1 = x > 0 ? x - 1 : m - 1, ym1 = y - 1;
            if (y == 1) {
                uint32_t i1 = 0, i2 = 1 + xm1, i3 = 1 + x;
                v3 v1 = sphere_point(0.0f, 0.0f, C, radius, m, p);
                v3 v2 = sphere_point((float)xm1, (float)y, C, radius, m, p);
                v3 v3_ = sphere_point((float)x, (float)y, C, radius, m, p);
                v3 n = normalize3(cross3(vsub(v3_, v1), vsub(v2, v1)));
                set_face(&faces[x], i1, i2, i3, n, headlight(col, n));
                set_vert(verts, i1, v1); set_vert(verts, i2, v2); set_vert(verts, i3, v3_);
            } else if (y < p - 1) {
                uint32_t fi = m + 2 * (y - 2) * m;
                uint32_t i1 = 1 + (ym1 - 1) * m + xm1, i2 = 1 + (ym1 - 1) * m + x;
                uint32_t i3 = 1 + (y - 1) * m + xm1, i4 = 1 + (y - 1) * m + x;
                v3 v1 = sphere_point((float)xm1, (float)ym1, C, radius, m, p);
                v3 v2 = sphere_point((float)x, (float)ym1, C, radius, m, p);
                v3 v3_ = sphere_point((float)xm1, (float)y, C, radius, m, p);
                v3 v4 = sphere_point((float)x, (float)y, C, radius, m, p);
                v3 n1 = normalize3(cross3(vsub(v4, v1), vsub(v3_, v1)));
                v3 n2 = normalize3(cross3(vsub(v4, v1), vsub(v2, v1)));
                set_face(&faces[fi + 2 * x], i1, i3, i4, n1, headlight(col, n1));
                set_face(&faces[fi + 2 * x + 1], i1, i2, i4, n2, headlight(col, n2));
                set_vert(verts, i1, v1); set_vert(verts, i2, v2); set_vert(verts, i3, v3_); set_vert(verts, i4, v4);
            } else {
                uint32_t fi = m + 2 * (y - 2) * m;
                uint32_t i1 = 1 + (y - 1) * m, i2 = 1 + (ym1 - 1) * m + xm1, i3 = 1 + (ym1 - 1) * m + x;
                v3 v1 = sphere_point(0.0f, (float)y, C, radius, m, p);
                v3 v2 = sphere_point((float)xm1, (float)ym1, C, radius, m, p);
                v3 v3_ = sphere_point((float)x, (float)ym1, C, radius, m, p);
                v3 n = normalize3(cross3(vsub(v3_, v1), vsub(v2, v1)));
                set_face(&faces[fi + x], i1, i2, i3, n, headlight(col, n));
                set_vert(verts, i1, v1); set_vert(verts, i2, v2); set_vert(verts, i3, v3_);
            }
        }
    }
}

/* One line of an object file as operator>>(char, float, float, float) reads it (Object.cpp:101-106): skip
 * whitespace, one char, three floats.  Returns 0 on a line the reference would call fatal. */
static int parse_obj_line(const char* line, char* type, float* a, float* b, float* c) {
    const char* s = line;
    while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n' || *s == '\v' || *s == '\f') s++;
    if (!*s) return 0;
    *type = *s++;
    char* e;
    float* out[3] = { a, b, c };
    for (int i = 0; i < 3; i++) {
        *out[i] = strtof(s, &e);
        if (e == s) return 0;
        s = e;
    }
    return 1;
}

/* create_object's counting pass, Object.cpp:84-119. */
int oracle_object_count(const char* path, uint32_t* nf, uint32_t* nv) {
    FILE* h = fopen(path, "r");
    if (!h) return -1;
    char line[1024]; *nf = 0; *nv = 0;
    while (fgets(line, sizeof line, h)) {
        char t; float a, b, c;
        if (!parse_obj_line(line, &t, &a, &b, &c)) { fclose(h); return -2; }
        if (t == 'f') ++*nf; else if (t == 'v') ++*nv;
    }
    fclose(h);
    return 0;
}

/* cpu_pre_render_object, Object.cpp:131-199. */
int oracle_prerender_object(const char* path, const float* center, float scale, const float* color,
                            rt3_gface* faces, uint32_t nf, float* verts, uint32_t nv) {
    FILE* h = fopen(path, "r");
    if (!h) return -1;
    char line[1024]; uint32_t vi = 0, fi = 0;
    v3 C = v3p(center), col = v3p(color);
    while (fgets(line, sizeof line, h)) {
        char t; float a, b, c;
        if (!parse_obj_line(line, &t, &a, &b, &c)) { fclose(h); return -2; }
        if (t == 'v') {
            if (vi >= nv) { fclose(h); return -3; }
            set_vert(verts, vi++, vadd(C, vscale(scale, V(a, b, c))));
        } else if (t == 'f') {
            if (fi >= nf) { fclose(h); return -3; }
            set_face(&faces[fi++], (uint32_t)a, (uint32_t)b, (uint32_t)c, V(0, 0, 0), col);
        }
    }
    fclose(h);
    uint32_t off = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < fi; i++) {
        if (faces[i].v1 < off) off = faces[i].v1;
        if (faces[i].v2 < off) off = faces[i].v2;
        if (faces[i].v3 < off) off = faces[i].v3;
    }
    for (uint32_t i = 0; i < fi; i++) {
        faces[i].v1 -= off; faces[i].v2 -= off; faces[i].v3 -= off;
        v3 p1 = v3p(&verts[4 * faces[i].v1]), p2 = v3p(&verts[4 * faces[i].v2]), p3 = v3p(&verts[4 * faces[i].v3]);
        v3 n = normalize3(cross3(vsub(p3, p1), vsub(p2, p1)));
        v3 c2 = headlight(v3p(faces[i].color), n);
        faces[i].normal[0] = n.x; faces[i].normal[1] = n.y; faces[i].normal[2] = n.z;
        faces[i].color[0] = c2.x; faces[i].color[1] = c2.y; faces[i].color[2] = c2.z;
    }
    return 0;
}

/* SequentialRenderer::transfer_entity, SequentialRenderer.cpp:174-195. */
void oracle_transfer_entity(rt3_gface* dst_f, uint32_t* dst_nf, float* dst_v, uint32_t* dst_nv,
                            const rt3_gface* f, uint32_t nf, const float* v, uint32_t nv) {
    uint32_t off = *dst_nv;
    for (uint32_t i = 0; i < nf; i++) {
        rt3_gface g = f[i];
        g.v1 += off; g.v2 += off; g.v3 += off;
        dst_f[(*dst_nf)++] = g;
    }
    memcpy(dst_v + 4 * (size_t)off, v, 16 * (size_t)nv);
    *dst_nv += nv;
}

/* Camera::update, Camera.cpp:77-96. */
void oracle_camera_update(rt3_camera* cam, float focal, float vw, float vh) {
    v3 o = V(0, 0, 0), hor = V(vw, 0, 0), ver = V(0, vh, 0);
    v3 two = V(2.0f, 2.0f, 2.0f);
    v3 hh = V(hor.x / two.x, hor.y / two.y, hor.z / two.z), vv = V(ver.x / two.x, ver.y / two.y, ver.z / two.z);
    v3 llc = vsub(vsub(vsub(o, hh), vv), V(0, 0, focal));
    memcpy(cam->origin, &o, 12); memcpy(cam->horizontal, &hor, 12);
    memcpy(cam->vertical, &ver, 12); memcpy(cam->lower_left_corner, &llc, 12);
}

/* ====================================================================================================
 * Shared by both modes: sky (A.4), pack (A.5), PPM (a8)
 * ================================================================================================== */
/* SequentialRenderer.cpp:105-107.  0.5*(y+1.0) is double arithmetic in the reference; the float form below
 * gives identical bits, and is what the GLSL twin does (raytracer_v3.glsl:139-141). */
static v3 sky(v3 d) {
    float len = sqrtf(dot3(d, d));
    float uy = d.y / len;
    float t = (float)(0.5 * ((double)uy + 1.0));
    float a = 1.0f - t;
    return V(a * 1.0f + t * 0.5f, a * 1.0f + t * 0.7f, a * 1.0f + t * 1.0f);
}
/* glm::packUnorm4x8(vec4(1, b, g, r)), func_packing.inl:67-83 + SequentialRenderer.cpp:297. */
static uint32_t pack_channel(float c) {
    float m = c < 0.0f ? 0.0f : c;         /* glm::max(x, 0) = (x < 0) ? 0 : x */
    m = 1.0f < m ? 1.0f : m;               /* glm::min(x, 1) = (1 < x) ? 1 : x */
    return (uint32_t)(uint8_t)roundf(m * 255.0f);
}
static uint32_t pack_pixel(v3 c) {
    return 0xFFu | (pack_channel(c.z) << 8) | (pack_channel(c.y) << 16) | (pack_channel(c.x) << 24);
}

uint64_t oracle_frame_ppm_bytes(const uint32_t* px, uint32_t w, uint32_t h, uint8_t* out, uint64_t cap) {
    char hdr[128];
    int n = snprintf(hdr, sizeof hdr, "P6\n# Image rendered by the RayTracer-3\n%u %u\n255\n", w, h);
    uint64_t need = (uint64_t)n + 3ull * w * h;
    if (!out) return need;
    if (cap < need) return 0;
    memcpy(out, hdr, (size_t)n);
    uint8_t* p = out + n;
    for (uint64_t i = 0; i < (uint64_t)w * h; i++) { *p++ = (px[i] >> 24) & 0xFF; *p++ = (px[i] >> 16) & 0xFF; *p++ = (px[i] >> 8) & 0xFF; }
    return need;
}

/* ====================================================================================================
 * Mode R — SequentialRenderer::render + ray_color, SequentialRenderer.cpp:47-109, 269-308
 * ================================================================================================== */
static v3 mode_r_ray(const rt3_camera* cam, uint32_t w, uint32_t h, uint32_t x, uint32_t y) {
    /* :289-290 — float / (float - 1.0) evaluated in double, rounded to float. */
    float u = (float)((double)(float)x / ((double)(float)w - 1.0));
    float v = (float)((double)(float)(h - 1 - y) / ((double)(float)h - 1.0));
    v3 llc = v3p(cam->lower_left_corner), hor = v3p(cam->horizontal), ver = v3p(cam->vertical), org = v3p(cam->origin);
    return vsub(vadd(vadd(llc, vscale(u, hor)), vscale(v, ver)), org);   /* :293 */
}

static v3 mode_r_ray_color(const rt3_gface* faces, uint32_t nf, const float* verts, v3 origin, v3 direction) {
    uint32_t min_i = 0;
    float min_t = (float)1e99;                                       /* :52  -> +inf */
    for (uint32_t i = 0; i < nf; i++) {
        v3 normal = v3p(faces[i].normal);
        if (dot3(direction, normal) == 0) continue;                  /* :56 */
        v3 p1 = v3p(&verts[4 * faces[i].v1]), p2 = v3p(&verts[4 * faces[i].v2]), p3 = v3p(&verts[4 * faces[i].v3]);
        float plane_distance = dot3(normal, p1);                     /* :67 */
        float t = (dot3(normal, origin) + plane_distance) / dot3(normal, direction);   /* :70 (sic: plus) */
        if (t < 0 || t >= min_t) continue;                           /* :71 */
        v3 hit = vadd(origin, vscale(t, direction));                 /* :77 */
        v3 a = cross3(vsub(p2, p1), vsub(hit, p1));
        v3 b = cross3(vsub(p3, p2), vsub(hit, p2));
        v3 c = cross3(vsub(p1, p3), vsub(hit, p3));
        if (-dot3(normal, a) >= 0.0 && -dot3(normal, b) >= 0.0 && -dot3(normal, c) >= 0.0) { min_i = i; min_t = t; }
    }
    if (min_t < 1e99) return v3p(faces[min_i].color);                /* :101-103 */
    return sky(direction);
}

/* rows [y0, y1) of the frame.  The reference loop covers y = h-2 .. 0 (:286); callers choose. */
void oracle_render_mode_r(const rt3_gface* faces, uint32_t nf, const float* verts, uint32_t nv,
                          const rt3_camera* cam, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                          uint32_t* out, int threads) {
    (void)nv;
    v3 org = v3p(cam->origin);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = 0; x < w; x++)
            out[(size_t)y * w + x] = pack_pixel(mode_r_ray_color(faces, nf, verts, org, mode_r_ray(cam, w, h, x, y)));
}

/* ====================================================================================================
 * Mode X — DESIGN.md §4.  Design intent: raytracer_v4.glsl:157-178 (sphere quadratic), :190-214 (sample
 * -> ray), :220-283 (bounce loop, faces tested before spheres), random_v1.glsl:22-52 (RNG),
 * reduce_v1.glsl (per-sample storage then in-order average); scatter models: the book the README cites.
 * ================================================================================================== */
uint32_t oracle_hash_u32(uint32_t x) {          /* _random_hash(uint), random_v1.glsl:22-29 */
    x += x << 10; x ^= x >> 6; x += x << 3; x ^= x >> 11; x += x << 15;
    return x;
}
static inline uint32_t hash2(uint32_t a, uint32_t b) { return oracle_hash_u32(a ^ oracle_hash_u32(b)); }  /* _random_hash(uvec2), :31 */
float oracle_random_float(uint32_t m) {         /* _random_float_construct, :37-52 */
    uint32_t bits = (m & 0x007FFFFFu) | 0x3F800000u;
    float f; memcpy(&f, &bits, 4);
    return f - 1.0f;
}
static inline uint32_t rng_base(uint32_t pixel, uint32_t sample, uint32_t seed) { return hash2(pixel, hash2(sample, seed)); }
static inline float rnd(uint32_t base, uint32_t ctr) { return oracle_random_float(hash2(base, ctr)); }

static inline float dotf(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

/* (cos, sin) of 2*pi*u, u in [0,1): quadrant split + Taylor polynomials on [0, pi/2), Horner in fmaf. */
void oracle_sincos2pi(float u, float* c_out, float* s_out) {
    float a = u * 4.0f;
    int k = (int)a;
    float f = a - (float)k;
    float x = f * 1.57079637f;
    float x2 = x * x;
    float p = fmaf(x2, -2.50521084e-8f, 2.75573192e-6f);
    p = fmaf(x2, p, -1.98412698e-4f);
    p = fmaf(x2, p, 8.33333333e-3f);
    p = fmaf(x2, p, -1.66666667e-1f);
    float s = fmaf(x * x2, p, x);
    float q = fmaf(x2, 2.08767570e-9f, -2.75573192e-7f);
    q = fmaf(x2, q, 2.48015873e-5f);
    q = fmaf(x2, q, -1.38888889e-3f);
    q = fmaf(x2, q, 4.16666667e-2f);
    q = fmaf(x2, q, -0.5f);
    float c = fmaf(x2, q, 1.0f);
    switch (k & 3) {
        case 0: *c_out = c;  *s_out = s;  break;
        case 1: *c_out = -s; *s_out = c;  break;
        case 2: *c_out = -c; *s_out = -s; break;
        default: *c_out = s; *s_out = -c; break;
    }
}
static v3 unit_vector(float xi0, float xi1) {
    float z = fmaf(-2.0f, xi0, 1.0f);
    float rr = fmaf(-z, z, 1.0f);
    float r = sqrtf(rr > 0.0f ? rr : 0.0f);
    float c, s; oracle_sincos2pi(xi1, &c, &s);
    return V(r * c, r * s, z);
}
static inline v3 normalize_x(v3 v) { float inv = 1.0f / sqrtf(dotf(v, v)); return V(v.x * inv, v.y * inv, v.z * inv); }
static inline v3 reflect_x(v3 d, v3 n) { float k = 2.0f * dotf(d, n); return V(fmaf(-k, n.x, d.x), fmaf(-k, n.y, d.y), fmaf(-k, n.z, d.z)); }

typedef struct {
    const rt3_gface* faces; uint32_t nf; const float* verts; const rt3_material* fmats;
    const float* spheres; const rt3_material* smats; uint32_t ns;
} scene_t;

/* nearest hit; kind 0 none, 1 triangle, 2 sphere.  Triangles first, then spheres; strict '<' keeps the
 * earlier primitive on equal t (raytracer_v4.glsl:226-246). */
/* literal: the reference's own t = (n.o + n.p1) / (n.d), SequentialRenderer.cpp:70 (sic) — ray cast 0 under
 * RT3_FLAG_REFERENCE_PRIMARY. */
static int nearest(const scene_t* sc, v3 o, v3 d, float tmin, int literal, float* t_out, uint32_t* i_out) {
    float tbest = INFINITY; uint32_t ibest = 0; int kind = 0;
    for (uint32_t i = 0; i < sc->nf; i++) {                          /* hit_vertex, raytracer_v4.glsl:116-153, with the */
        const rt3_gface* f = &sc->faces[i];                          /* sign of n.o corrected for origin != 0           */
        v3 n = v3p(f->normal);
        float nd = dot3(d, n);
        if (nd == 0) continue;
        v3 p1 = v3p(&sc->verts[4 * f->v1]), p2 = v3p(&sc->verts[4 * f->v2]), p3 = v3p(&sc->verts[4 * f->v3]);
        float t = literal ? (dot3(n, o) + dot3(n, p1)) / nd : (dot3(n, p1) - dot3(n, o)) / nd;
        if (!(t >= tmin && t < tbest)) continue;
        v3 hit = vadd(o, vscale(t, d));
        v3 a = cross3(vsub(p2, p1), vsub(hit, p1));
        v3 b = cross3(vsub(p3, p2), vsub(hit, p2));
        v3 c = cross3(vsub(p1, p3), vsub(hit, p3));
        if (-dot3(n, a) >= 0.0f && -dot3(n, b) >= 0.0f && -dot3(n, c) >= 0.0f) { tbest = t; ibest = i; kind = 1; }
    }
    for (uint32_t i = 0; i < sc->ns; i++) {                          /* hit_sphere, raytracer_v4.glsl:157-178, unit d   */
        const float* s = &sc->spheres[4 * i];
        v3 oc = V(s[0] - o.x, s[1] - o.y, s[2] - o.z);
        float r2 = s[3] * s[3];
        float h = fmaf(oc.z, d.z, fmaf(oc.y, d.y, oc.x * d.x));
        float c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -r2)));
        float disc = fmaf(h, h, -c);
        if (!((c < 0.0f) || (disc > 0.0f && h > 0.0f))) continue;
        float sq = sqrtf(disc);
        float t = h - sq;
        if (!(t > tmin)) t = h + sq;
        if (t > tmin && t < tbest) { tbest = t; ibest = i; kind = 2; }
    }
    *t_out = tbest; *i_out = ibest;
    return kind;
}

/* radiance of one sample (pixel x,y of the full frame, sample s). */
static v3 sample_radiance(const scene_t* sc, const rt3_camera* cam, const rt3_params* P, uint32_t x, uint32_t y, uint32_t s,
                          uint64_t* casts) {
    const uint32_t W = P->width, H = P->height;
    uint32_t base = rng_base(y * W + x, s, P->seed);
    float jx = 0.0f, jy = 0.0f;
    if (P->spp > 1) {                                                /* raytracer_v4.glsl:197-206, offsets in PIXEL units */
        float xi0 = rnd(base, 1), xi1 = rnd(base, 2);
        uint32_t edge = (uint32_t)sqrtf((float)P->spp);
        while (edge * edge > P->spp) edge--;
        while ((edge + 1) * (edge + 1) <= P->spp) edge++;
        if (edge * edge == P->spp) {
            uint32_t sx = s % edge, sy = s / edge;
            jx = ((float)sx + xi0) / (float)edge - 0.5f;
            jy = ((float)sy + xi1) / (float)edge - 0.5f;
        } else { jx = xi0 - 0.5f; jy = xi1 - 0.5f; }
    }
    float u = ((float)x + jx) / ((float)W - 1.0f);
    float v = ((float)(H - 1 - y) + jy) / ((float)H - 1.0f);
    v3 llc = v3p(cam->lower_left_corner), hor = v3p(cam->horizontal), ver = v3p(cam->vertical), org = v3p(cam->origin);
    v3 dir = vsub(vadd(vadd(llc, vscale(u, hor)), vscale(v, ver)), org);
    v3 o = org;
    if (P->lens_radius > 0.0f) {
        float xi2 = rnd(base, 3), xi3 = rnd(base, 4);
        float r = P->lens_radius * sqrtf(xi2);
        float c, sn; oracle_sincos2pi(xi3, &c, &sn);
        float a = r * c, b = r * sn;
        float lh = sqrtf(dot3(hor, hor)), lv = sqrtf(dot3(ver, ver));
        v3 U = V(hor.x / lh, hor.y / lh, hor.z / lh), Vv = V(ver.x / lv, ver.y / lv, ver.z / lv);
        v3 off = V(a * U.x + b * Vv.x, a * U.y + b * Vv.y, a * U.z + b * Vv.z);
        o = vadd(org, off);
        dir = vsub(dir, off);
    }
    /* RT3_FLAG_REFERENCE_PRIMARY: ray cast 0 is the reference's — unnormalised direction (SequentialRenderer.cpp:293),
     * literal plane formula (:70), sky of that direction (:105-107), hit point o + t d with them (:77); the scatter
     * formulas then get the unit direction.  Otherwise the direction is normalised here. */
    const int ref = (P->flags & RT3_FLAG_REFERENCE_PRIMARY) != 0;
    float inv = 1.0f / sqrtf(dot3(dir, dir));
    v3 d = ref ? dir : V(dir.x * inv, dir.y * inv, dir.z * inv);

    v3 L = V(0, 0, 0), thr = V(1, 1, 1);
    for (uint32_t k = 0; k < P->max_depth; k++) {
        float t; uint32_t idx;
        ++*casts;
        int kind = nearest(sc, o, d, P->t_min, ref && k == 0, &t, &idx);
        if (!kind) {
            if (!(P->flags & RT3_FLAG_BLACK_BACKGROUND)) { v3 c = sky(d); L = V(fmaf(thr.x, c.x, L.x), fmaf(thr.y, c.y, L.y), fmaf(thr.z, c.z, L.z)); }
            break;
        }
        rt3_material m; v3 p, nout;
        if (kind == 1) {
            const rt3_gface* f = &sc->faces[idx];
            if (sc->fmats) m = sc->fmats[idx];
            else { m.kind = RT3_MAT_FLAT; m.param = 0; memcpy(m.rgb, f->color, 12); }
            p = vadd(o, vscale(t, d));
            nout = v3p(f->normal);
        } else {
            const float* sp = &sc->spheres[4 * idx];
            m = sc->smats[idx];
            p = V(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z));
            float invr = 1.0f / sp[3];
            nout = V((p.x - sp[0]) * invr, (p.y - sp[1]) * invr, (p.z - sp[2]) * invr);
        }
        v3 rgb = v3p(m.rgb);
        if (m.kind == RT3_MAT_FLAT) { L = V(fmaf(thr.x, rgb.x, L.x), fmaf(thr.y, rgb.y, L.y), fmaf(thr.z, rgb.z, L.z)); break; }
        if (k + 1 == P->max_depth) break;
        if (ref && k == 0) { float iv = 1.0f / sqrtf(dot3(d, d)); d = V(d.x * iv, d.y * iv, d.z * iv); }
        int front = dotf(d, nout) < 0.0f;
        v3 n = front ? nout : vneg(nout);
        uint32_t ctr = 1 + 8 * (k + 1);
        v3 nd;
        if (m.kind == RT3_MAT_LAMBERT) {
            v3 uv = unit_vector(rnd(base, ctr), rnd(base, ctr + 1));
            nd = vadd(n, uv);
            if (fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;
        } else if (m.kind == RT3_MAT_METAL) {
            v3 rn = normalize_x(reflect_x(d, n));
            nd = rn;
            if (m.param > 0.0f) {
                v3 uv = unit_vector(rnd(base, ctr), rnd(base, ctr + 1));
                nd = V(fmaf(m.param, uv.x, rn.x), fmaf(m.param, uv.y, rn.y), fmaf(m.param, uv.z, rn.z));
            }
            if (!(dotf(nd, n) > 0.0f)) break;                        /* absorbed */
        } else {                                                     /* RT3_MAT_DIELECTRIC */
            float ri = front ? 1.0f / m.param : m.param;
            float cosv = -dotf(d, n);
            if (cosv > 1.0f) cosv = 1.0f;
            float s2 = fmaf(-cosv, cosv, 1.0f);
            float sinv = sqrtf(s2 > 0.0f ? s2 : 0.0f);
            int cannot = ri * sinv > 1.0f;
            float r0 = (1.0f - ri) / (1.0f + ri); r0 = r0 * r0;
            float xx = 1.0f - cosv, x2 = xx * xx, x5 = x2 * x2 * xx;
            float R = fmaf(1.0f - r0, x5, r0);
            if (cannot || R > rnd(base, ctr + 2)) nd = reflect_x(d, n);
            else {
                v3 perp = V(fmaf(cosv, n.x, d.x) * ri, fmaf(cosv, n.y, d.y) * ri, fmaf(cosv, n.z, d.z) * ri);
                float par = -sqrtf(fabsf(1.0f - dotf(perp, perp)));
                nd = V(fmaf(par, n.x, perp.x), fmaf(par, n.y, perp.y), fmaf(par, n.z, perp.z));
            }
            rgb = V(1, 1, 1);
        }
        d = normalize_x(nd);
        o = p;
        thr = vmul(thr, rgb);
    }
    return L;
}

static int row_owned(const rt3_params* P, uint32_t y) {
    if (P->tile_count <= 1) return 1;
    return ((y / P->tile_rows) % P->tile_count) == P->tile_index;
}
uint32_t oracle_rows_owned(const rt3_params* P) {
    uint32_t n = 0;
    for (uint32_t y = 0; y < P->height; y++) n += (uint32_t)row_owned(P, y);
    return n;
}

/* Renders samples [s_begin, s_begin + s_count) of the rows this shard owns (compact buffer) and adds them, in sample
 * order, to the running per-pixel sums — the progressive accumulation of rt3_render_path_range (design intent:
 * reduce_v1.glsl:28-76).  sum_io / sumsq_io: 4 floats per pixel (r, g, b, 0), read when s_begin > 0, written back;
 * both optional (sumsq_io is only touched with RT3_FLAG_VARIANCE: sq = fma(L, L, sq)).  out = the frame resolved over
 * the s_begin + s_count samples so far.  Returns ray casts (UINT64_MAX: unsupported combination). */
uint64_t oracle_render_path_range(const rt3_gface* faces, uint32_t nf, const float* verts, const rt3_material* fmats,
                                  const float* spheres, const rt3_material* smats, uint32_t ns,
                                  const rt3_camera* cam, const rt3_params* P, uint32_t s_begin, uint32_t s_count,
                                  uint32_t* out, float* sum_io, float* sumsq_io, int threads) {
    if ((P->flags & RT3_FLAG_REFERENCE_PRIMARY) && ns != 0) return UINT64_MAX;      /* triangle-only, as the product */
    if (s_begin != 0 && !sum_io) return UINT64_MAX;
    scene_t sc = { faces, nf, verts, fmats, spheres, smats, ns };
    uint32_t* rows = (uint32_t*)malloc(sizeof(uint32_t) * (P->height ? P->height : 1));
    uint32_t nrows = 0;
    for (uint32_t y = 0; y < P->height; y++) if (row_owned(P, y)) rows[nrows++] = y;
    const int var = (P->flags & RT3_FLAG_VARIANCE) != 0 && sumsq_io != NULL;
    uint64_t casts = 0;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1) reduction(+:casts)
    for (uint32_t r = 0; r < nrows; r++) {
        uint32_t y = rows[r];
        for (uint32_t x = 0; x < P->width; x++) {
            size_t li = (size_t)r * P->width + x;
            v3 sum = V(0, 0, 0), sq = V(0, 0, 0);
            if (s_begin != 0) {
                sum = v3p(&sum_io[4 * li]);
                if (var) sq = v3p(&sumsq_io[4 * li]);
            }
            for (uint32_t s = s_begin; s < s_begin + s_count; s++) {  /* reduce_v1.glsl intent: samples summed in order */
                v3 L = sample_radiance(&sc, cam, P, x, y, s, &casts);
                sum = vadd(sum, L);
                if (var) sq = V(fmaf(L.x, L.x, sq.x), fmaf(L.y, L.y, sq.y), fmaf(L.z, L.z, sq.z));
            }
            if (sum_io) { sum_io[4 * li] = sum.x; sum_io[4 * li + 1] = sum.y; sum_io[4 * li + 2] = sum.z; sum_io[4 * li + 3] = 0.0f; }
            if (var) { sumsq_io[4 * li] = sq.x; sumsq_io[4 * li + 1] = sq.y; sumsq_io[4 * li + 2] = sq.z; sumsq_io[4 * li + 3] = 0.0f; }
            float n = (float)(s_begin + s_count);
            v3 c = V(sum.x / n, sum.y / n, sum.z / n);
            if (P->flags & RT3_FLAG_GAMMA2) c = V(c.x > 0 ? sqrtf(c.x) : 0.0f, c.y > 0 ? sqrtf(c.y) : 0.0f, c.z > 0 ? sqrtf(c.z) : 0.0f);
            out[li] = pack_pixel(c);
        }
    }
    free(rows);
    return casts;
}

/* The whole render: samples [0, spp).  out_sum (optional): the per-pixel sums, 3 floats per pixel. */
uint64_t oracle_render_path(const rt3_gface* faces, uint32_t nf, const float* verts, const rt3_material* fmats,
                            const float* spheres, const rt3_material* smats, uint32_t ns,
                            const rt3_camera* cam, const rt3_params* P, uint32_t* out, float* out_sum, int threads) {
    float* sum4 = NULL;
    size_t npix = (size_t)oracle_rows_owned(P) * P->width;
    if (out_sum) sum4 = (float*)malloc(sizeof(float) * 4 * (npix ? npix : 1));
    uint64_t casts = oracle_render_path_range(faces, nf, verts, fmats, spheres, smats, ns, cam, P, 0, P->spp, out, sum4, NULL, threads);
    if (out_sum) {
        for (size_t i = 0; i < npix; i++) { out_sum[3 * i] = sum4[4 * i]; out_sum[3 * i + 1] = sum4[4 * i + 1]; out_sum[3 * i + 2] = sum4[4 * i + 2]; }
        free(sum4);
    }
    return casts;
}

/* Element-wise float ops for the device arithmetic parity test (tests/test_gpu_arith.py). */
void oracle_arith(const float* a, const float* b, uint32_t n, float* div, float* sq, float* fm) {
    for (uint32_t i = 0; i < n; i++) { div[i] = a[i] / b[i]; sq[i] = sqrtf(fabsf(a[i])); fm[i] = fmaf(a[i], b[i], a[i]); }
}

/* ---- small probes for the known-answer tests (tests/test_oracle_kat.py) ---- */
void oracle_sky(const float* d, float* rgb) { v3 c = sky(v3p(d)); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z; }
uint32_t oracle_pack_pixel(float r, float g, float b) { return pack_pixel(V(r, g, b)); }
/* nearest hit of one ray; returns kind (0 none / 1 triangle / 2 sphere). */
int oracle_nearest(const rt3_gface* faces, uint32_t nf, const float* verts, const float* spheres, uint32_t ns,
                   const float* o, const float* d, float tmin, float* t, uint32_t* idx) {
    scene_t sc = { faces, nf, verts, NULL, spheres, NULL, ns };
    return nearest(&sc, v3p(o), v3p(d), tmin, 0, t, idx);
}
/* Mode-R colour of one ray (ray_color, SequentialRenderer.cpp:47-109). */
void oracle_ray_color(const rt3_gface* faces, uint32_t nf, const float* verts, const float* o, const float* d, float* rgb) {
    v3 c = mode_r_ray_color(faces, nf, verts, v3p(o), v3p(d));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
/* sky + pack applied element-wise, for the device arithmetic parity test. */
void oracle_arith2(const float* a, const float* b, uint32_t n, float* cs, float* sn, float* sk3, uint32_t* pk) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t bits; memcpy(&bits, &a[i], 4);
        float u = oracle_random_float(bits);
        oracle_sincos2pi(u, &cs[i], &sn[i]);
        v3 c = sky(V(a[i], b[i], -2.0f));
        sk3[3 * i] = c.x; sk3[3 * i + 1] = c.y; sk3[3 * i + 2] = c.z;
        pk[i] = pack_pixel(V(a[i], b[i], u));
    }
}
