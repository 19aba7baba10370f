// rt3_host.cpp — host half of librt3hip.so: the scene API that runs before the render path
// (entities -> GFace[] / vec4[]), camera set-up, PPM serialisation, benchmark scene builders.
//
// Everything here is ordinary CPU code in the reference too (src/lib/entities/*.cpp, src/lib/camera/*.cpp);
// the render path itself (rt3_render*, rt3_render_path*) lives in rt3_device.hip and has no CPU form.
// Compiled with -ffp-contract=off: the reference's flattening is plain IEEE binary32 with libm double trig.
#include "rt3.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit Vec3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, Vec3 b) { return Vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
// glm::dot / glm::cross / glm::normalize as the reference's vendored GLM 0.9.9.8 evaluates them
// (glm/detail/func_geometric.inl:48-55, :68-79, :82-90): sums left to right, normalize = v * (1/sqrt(v.v)).
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return Vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline Vec3 normalize(Vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

inline void put_vertex(float* vertices, uint32_t index, Vec3 p) {
    float* v = vertices + 4 * (size_t)index;
    v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = 0.0f;
}
inline void put_face(rt3_gface& f, uint32_t a, uint32_t b, uint32_t c, Vec3 n, Vec3 col) {
    std::memset(&f, 0, sizeof f);
    f.v1 = a; f.v2 = b; f.v3 = c;
    f.normal[0] = n.x; f.normal[1] = n.y; f.normal[2] = n.z;
    f.color[0] = col.x; f.color[1] = col.y; f.color[2] = col.z;
}
// colour * |n . (0,0,-1)|  — the baked "headlight" shading of Sphere.cpp:155 and Object.cpp:194.
inline Vec3 bake(Vec3 color, Vec3 n) { return color * std::fabs(dot(n, Vec3(0.0f, 0.0f, -1.0f))); }

// One point of the UV sphere, Sphere.cpp:69-79 (double-precision trig, float everything else).
struct UvSphere {
    Vec3 center; float radius; uint32_t meridians, parallels;
    Vec3 at(uint32_t ix, uint32_t iy) const {
        const float fx = (float)ix, fy = (float)iy;
        const double theta = M_PI * (fy / (float)(parallels - 1));
        const double phi = 2 * M_PI * (fx / (float)meridians);
        return center + radius * Vec3((float)(std::sin(theta) * std::cos(phi)), (float)std::cos(theta),
                                      (float)(std::sin(theta) * std::sin(phi)));
    }
    // vertex index of ring iy (1..parallels-2), column ix
    uint32_t ring(uint32_t iy, uint32_t ix) const { return 1 + (iy - 1) * meridians + ix; }
};

}  // namespace

extern "C" {

void rt3_prerender_triangle(const float p1[3], const float p2[3], const float p3[3], const float color[3],
                            rt3_gface* faces, float* vertices) {
    const Vec3 a(p1), b(p2), c(p3);
    put_vertex(vertices, 0, a);
    put_vertex(vertices, 1, b);
    put_vertex(vertices, 2, c);
    put_face(faces[0], 0, 1, 2, normalize(cross(c - a, b - a)), Vec3(color));   // Triangle.cpp:48, colour unshaded
}

uint32_t rt3_sphere_face_count(uint32_t m, uint32_t p) { return m + 2 * ((p - 3) * m) + m; }
uint32_t rt3_sphere_vertex_count(uint32_t m, uint32_t p) { return 2 + (p - 2) * m; }

// Sphere.cpp:120-261.  The reference walks (y, x) and rewrites shared vertices several times with identical
// values; here vertices are written once, then faces ring by ring — same arrays.
void rt3_prerender_sphere(const float center[3], float radius, uint32_t m, uint32_t p, const float color[3],
                          rt3_gface* faces, float* vertices) {
    const UvSphere s{ Vec3(center), radius, m, p };
    const Vec3 col(color);
    const uint32_t south = 1 + (p - 2) * m;

    put_vertex(vertices, 0, s.at(0, 0));
    for (uint32_t iy = 1; iy + 1 < p; iy++)
        for (uint32_t ix = 0; ix < m; ix++) put_vertex(vertices, s.ring(iy, ix), s.at(ix, iy));
    put_vertex(vertices, south, s.at(0, p - 1));

    for (uint32_t ix = 0; ix < m; ix++) {                       // north cap, faces [0, m)
        const uint32_t prev = ix > 0 ? ix - 1 : m - 1;
        const Vec3 v1 = s.at(0, 0), v2 = s.at(prev, 1), v3 = s.at(ix, 1);
        const Vec3 n = normalize(cross(v3 - v1, v2 - v1));
        put_face(faces[ix], 0, 1 + prev, 1 + ix, n, bake(col, n));
    }
    for (uint32_t iy = 2; iy + 1 < p; iy++) {                   // quads between ring iy-1 and ring iy
        const uint32_t base = m + 2 * (iy - 2) * m;
        for (uint32_t ix = 0; ix < m; ix++) {
            const uint32_t prev = ix > 0 ? ix - 1 : m - 1;
            const Vec3 v1 = s.at(prev, iy - 1), v2 = s.at(ix, iy - 1), v3 = s.at(prev, iy), v4 = s.at(ix, iy);
            const Vec3 n1 = normalize(cross(v4 - v1, v3 - v1)), n2 = normalize(cross(v4 - v1, v2 - v1));
            const uint32_t i1 = s.ring(iy - 1, prev), i2 = s.ring(iy - 1, ix), i3 = s.ring(iy, prev), i4 = s.ring(iy, ix);
            put_face(faces[base + 2 * ix], i1, i3, i4, n1, bake(col, n1));
            put_face(faces[base + 2 * ix + 1], i1, i2, i4, n2, bake(col, n2));
        }
    }
    {                                                           // south cap
        const uint32_t iy = p - 1, base = m + 2 * (iy - 2) * m;
        for (uint32_t ix = 0; ix < m; ix++) {
            const uint32_t prev = ix > 0 ? ix - 1 : m - 1;
            const Vec3 v1 = s.at(0, iy), v2 = s.at(prev, iy - 1), v3 = s.at(ix, iy - 1);
            const Vec3 n = normalize(cross(v3 - v1, v2 - v1));
            put_face(faces[base + ix], south, s.ring(iy - 1, prev), s.ring(iy - 1, ix), n, bake(col, n));
        }
    }
}

// Object.cpp:84-119 — every line must read as `char float float float`, otherwise the reference aborts.
int rt3_object_count(const char* path, uint32_t* n_faces, uint32_t* n_vertices) {
    std::ifstream in(path);
    if (!in.is_open()) return RT3_E_IO;
    uint32_t nf = 0, nv = 0;
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line);
        char kind; float a, b, c;
        if (!(ss >> kind >> a >> b >> c)) return RT3_E_IO;
        if (kind == 'f') ++nf; else if (kind == 'v') ++nv;
    }
    *n_faces = nf; *n_vertices = nv;
    return 0;
}

// Object.cpp:131-199.
int rt3_prerender_object(const char* path, const float center[3], float scale, const float color[3],
                         rt3_gface* faces, uint32_t n_faces, float* vertices, uint32_t n_vertices) {
    std::ifstream in(path);
    if (!in.is_open()) return RT3_E_IO;
    const Vec3 origin(center), col(color);
    uint32_t nf = 0, nv = 0;
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line);
        char kind; float a, b, c;
        if (!(ss >> kind >> a >> b >> c)) return RT3_E_IO;
        if (kind == 'v') {
            if (nv >= n_vertices) return RT3_E_ARG;
            put_vertex(vertices, nv++, origin + scale * Vec3(a, b, c));
        } else if (kind == 'f') {
            if (nf >= n_faces) return RT3_E_ARG;
            // indices arrive as floats (Object.cpp:157-170); the reference casts blindly, here anything that is not a
            // representable index is a malformed file (the cast of a negative, non-finite or >= 2^32 float is undefined)
            auto index_ok = [](float x) { return x >= 0.0f && x < 4294967296.0f; };     // false for NaN
            if (!index_ok(a) || !index_ok(b) || !index_ok(c)) return RT3_E_IO;
            put_face(faces[nf++], (uint32_t)a, (uint32_t)b, (uint32_t)c, Vec3(), col);
        }
    }
    uint32_t lowest = 0xFFFFFFFFu;                              // indices need not be zero-based (:181-186)
    for (uint32_t i = 0; i < nf; i++) {
        lowest = faces[i].v1 < lowest ? faces[i].v1 : lowest;
        lowest = faces[i].v2 < lowest ? faces[i].v2 : lowest;
        lowest = faces[i].v3 < lowest ? faces[i].v3 : lowest;
    }
    for (uint32_t i = 0; i < nf; i++) {
        rt3_gface& f = faces[i];
        f.v1 -= lowest; f.v2 -= lowest; f.v3 -= lowest;
        if (f.v1 >= nv || f.v2 >= nv || f.v3 >= nv) return RT3_E_IO;         // a face names a vertex the file does not hold
        const Vec3 a(vertices + 4 * (size_t)f.v1), b(vertices + 4 * (size_t)f.v2), c(vertices + 4 * (size_t)f.v3);
        const Vec3 n = normalize(cross(c - a, b - a));
        const Vec3 shaded = bake(Vec3(f.color), n);
        f.normal[0] = n.x; f.normal[1] = n.y; f.normal[2] = n.z;
        f.color[0] = shaded.x; f.color[1] = shaded.y; f.color[2] = shaded.z;
    }
    return 0;
}

// SequentialRenderer.cpp:174-195.
void rt3_transfer_entity(rt3_gface* dst_faces, uint32_t* dst_nf, float* dst_vertices, uint32_t* dst_nv,
                         const rt3_gface* faces, uint32_t nf, const float* vertices, uint32_t nv) {
    const uint32_t rebase = *dst_nv;
    rt3_gface* out = dst_faces + *dst_nf;
    for (uint32_t i = 0; i < nf; i++) {
        out[i] = faces[i];
        out[i].v1 += rebase; out[i].v2 += rebase; out[i].v3 += rebase;
    }
    std::memcpy(dst_vertices + 4 * (size_t)rebase, vertices, sizeof(float) * 4 * (size_t)nv);
    *dst_nf += nf;
    *dst_nv += nv;
}

// Camera.cpp:89-92.
void rt3_camera_update(rt3_camera* cam, float focal_length, float viewport_width, float viewport_height) {
    const Vec3 origin(0.0f, 0.0f, 0.0f), horizontal(viewport_width, 0.0f, 0.0f), vertical(0.0f, viewport_height, 0.0f);
    const Vec3 two(2.0f, 2.0f, 2.0f);
    const Vec3 llc = origin - horizontal / two - vertical / two - Vec3(0.0f, 0.0f, focal_length);
    std::memcpy(cam->origin, &origin, 12);
    std::memcpy(cam->horizontal, &horizontal, 12);
    std::memcpy(cam->vertical, &vertical, 12);
    std::memcpy(cam->lower_left_corner, &llc, 12);
}

// Book camera (look-from / look-at) expressed in the reference's four vectors.  Build-owned extension.
void rt3_camera_look_at(rt3_camera* cam, const float from[3], const float at[3], const float vup[3],
                        float vfov_deg, float aspect, float focus_dist) {
    const double theta = (double)vfov_deg * M_PI / 180.0;
    const float half_h = (float)std::tan(theta / 2.0);
    const float vh = 2.0f * half_h * focus_dist, vw = vh * aspect;
    const Vec3 f(from), a(at), up(vup);
    const Vec3 w = normalize(f - a), u = normalize(cross(up, w)), v = cross(w, u);
    const Vec3 horizontal = vw * u, vertical = vh * v;
    const Vec3 llc = f - 0.5f * horizontal - 0.5f * vertical - focus_dist * w;
    std::memcpy(cam->origin, &f, 12);
    std::memcpy(cam->horizontal, &horizontal, 12);
    std::memcpy(cam->vertical, &vertical, 12);
    std::memcpy(cam->lower_left_corner, &llc, 12);
}

// Frame.cpp:125-143.
uint64_t rt3_frame_ppm_bytes(const uint32_t* pixels, uint32_t width, uint32_t height, uint8_t* out, uint64_t cap) {
    const std::string header = "P6\n# Image rendered by the RayTracer-3\n" + std::to_string(width) + " " +
                               std::to_string(height) + "\n255\n";
    const uint64_t total = header.size() + 3ull * width * height;
    if (out == nullptr) return total;
    if (cap < total) return 0;
    std::memcpy(out, header.data(), header.size());
    uint8_t* p = out + header.size();
    for (uint64_t i = 0, n = (uint64_t)width * height; i < n; i++) {
        const uint32_t px = pixels[i];
        *p++ = (uint8_t)(px >> 24); *p++ = (uint8_t)(px >> 16); *p++ = (uint8_t)(px >> 8);
    }
    return total;
}

int rt3_frame_to_ppm(const uint32_t* pixels, uint32_t width, uint32_t height, const char* path) {
    std::vector<uint8_t> bytes(rt3_frame_ppm_bytes(pixels, width, height, nullptr, 0));
    rt3_frame_ppm_bytes(pixels, width, height, bytes.data(), bytes.size());
    FILE* f = std::fopen(path, "wb");
    if (!f) return RT3_E_IO;
    const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    return n == bytes.size() ? 0 : RT3_E_IO;
}

// random_v1.glsl:22-29 and :37-52.
uint32_t rt3_hash_u32(uint32_t x) {
    x += x << 10; x ^= x >> 6; x += x << 3; x ^= x >> 11; x += x << 15;
    return x;
}
float rt3_random_float(uint32_t m) {
    const uint32_t bits = (m & 0x007FFFFFu) | 0x3F800000u;
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f - 1.0f;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------
// Benchmark scenes.  xi(seed, slot, dim) = random(hash(uvec2(hash(uvec2(slot*8+dim, seed)))).
// ------------------------------------------------------------------------------------------------------
namespace {

inline float xi(uint32_t seed, uint32_t slot, uint32_t dim) {
    const uint32_t key = slot * 8u + dim + 1u;                  // never hash a zero key: hash(0) == 0
    return rt3_random_float(rt3_hash_u32(key ^ rt3_hash_u32(seed)));
}

struct SphereSink {
    float* cr; rt3_material* mats; uint32_t cap; uint32_t n;
    void add(float x, float y, float z, float r, uint32_t kind, float cr_, float cg, float cb, float param) {
        if (cr != nullptr && n < cap) {
            float* s = cr + 4 * (size_t)n;
            s[0] = x; s[1] = y; s[2] = z; s[3] = r;
            rt3_material& m = mats[n];
            m.rgb[0] = cr_; m.rgb[1] = cg; m.rgb[2] = cb; m.param = param; m.kind = kind;
        }
        n++;
    }
};

}  // namespace

extern "C" {

// BASELINE.json config 1: ground + two r=0.5 spheres, all Lambertian.
uint32_t rt3_scene_three_spheres(float* center_radius, rt3_material* materials, uint32_t cap) {
    SphereSink out{ center_radius, materials, cap, 0 };
    out.add(0.0f, -100.5f, -1.0f, 100.0f, RT3_MAT_LAMBERT, 0.8f, 0.8f, 0.0f, 0.0f);
    out.add(0.0f, 0.0f, -1.0f, 0.5f, RT3_MAT_LAMBERT, 0.7f, 0.3f, 0.3f, 0.0f);
    out.add(1.0f, 0.0f, -1.0f, 0.5f, RT3_MAT_LAMBERT, 0.8f, 0.6f, 0.2f, 0.0f);
    return out.n;
}

// BASELINE.json config 2/3: the book's final scene.  22x22 grid of r=0.2 spheres, jittered, minus those within
// 0.9 of (4, 0.2, 0); material by xi: <0.8 Lambertian (albedo xi*xi), <0.95 metal (albedo in [0.5,1), fuzz in
// [0,0.5)), else glass 1.5; plus ground r=1000 and three r=1 spheres.
uint32_t rt3_scene_weekend(uint32_t seed, float* center_radius, rt3_material* materials, uint32_t cap) {
    SphereSink out{ center_radius, materials, cap, 0 };
    out.add(0.0f, -1000.0f, 0.0f, 1000.0f, RT3_MAT_LAMBERT, 0.5f, 0.5f, 0.5f, 0.0f);
    uint32_t slot = 0;
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++, slot++) {
            const float choose = xi(seed, slot, 0);
            const float cx = (float)a + 0.9f * xi(seed, slot, 1), cy = 0.2f, cz = (float)b + 0.9f * xi(seed, slot, 2);
            const float dx = cx - 4.0f, dy = cy - 0.2f, dz = cz - 0.0f;
            if (!(std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9f)) continue;
            if (choose < 0.8f) {
                out.add(cx, cy, cz, 0.2f, RT3_MAT_LAMBERT, xi(seed, slot, 3) * xi(seed, slot, 4),
                        xi(seed, slot, 5) * xi(seed, slot, 6), xi(seed, slot, 7) * xi(seed, slot + 65536u, 0), 0.0f);
            } else if (choose < 0.95f) {
                out.add(cx, cy, cz, 0.2f, RT3_MAT_METAL, 0.5f + 0.5f * xi(seed, slot, 3), 0.5f + 0.5f * xi(seed, slot, 4),
                        0.5f + 0.5f * xi(seed, slot, 5), 0.5f * xi(seed, slot, 6));
            } else {
                out.add(cx, cy, cz, 0.2f, RT3_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
            }
        }
    }
    out.add(0.0f, 1.0f, 0.0f, 1.0f, RT3_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
    out.add(-4.0f, 1.0f, 0.0f, 1.0f, RT3_MAT_LAMBERT, 0.4f, 0.2f, 0.1f, 0.0f);
    out.add(4.0f, 1.0f, 0.0f, 1.0f, RT3_MAT_METAL, 0.7f, 0.6f, 0.5f, 0.0f);
    return out.n;
}

// BASELINE.json config 4: n Lambertian spheres, centres uniform in [-50,50]x[0.2,20]x[-100,0], r in [0.05,0.4].
uint32_t rt3_scene_stress(uint32_t n, uint32_t seed, float* center_radius, rt3_material* materials, uint32_t cap) {
    SphereSink out{ center_radius, materials, cap, 0 };
    for (uint32_t i = 0; i < n; i++) {
        out.add(-50.0f + 100.0f * xi(seed, i, 0), 0.2f + 19.8f * xi(seed, i, 1), -100.0f * xi(seed, i, 2),
                0.05f + 0.35f * xi(seed, i, 3), RT3_MAT_LAMBERT, 0.1f + 0.8f * xi(seed, i, 4), 0.1f + 0.8f * xi(seed, i, 5),
                0.1f + 0.8f * xi(seed, i, 6), 0.0f);
    }
    return out.n;
}

}  // extern "C"

// Cornell-style box (BASELINE.json config 5): five walls and two boxes, each rectangle split into grid x grid
// quads of two triangles, plus a two-triangle emissive quad under the ceiling.  Unindexed (3 vertices per face).
namespace {

struct MeshSink {
    rt3_gface* faces; float* vertices; rt3_material* mats; uint32_t cap; uint32_t n;
    void tri(Vec3 a, Vec3 b, Vec3 c, const rt3_material& m) {
        if (faces != nullptr && n < cap) {
            const float col[3] = { m.rgb[0], m.rgb[1], m.rgb[2] };
            const float pa[3] = { a.x, a.y, a.z }, pb[3] = { b.x, b.y, b.z }, pc[3] = { c.x, c.y, c.z };
            rt3_prerender_triangle(pa, pb, pc, col, faces + n, vertices + 12 * (size_t)n);
            faces[n].v1 = 3 * n; faces[n].v2 = 3 * n + 1; faces[n].v3 = 3 * n + 2;
            if (mats != nullptr) mats[n] = m;
        }
        n++;
    }
    // rectangle o + s*eu + t*ev, s,t in [0,1], tessellated g x g
    void rect(Vec3 o, Vec3 eu, Vec3 ev, uint32_t g, const rt3_material& m) {
        for (uint32_t j = 0; j < g; j++)
            for (uint32_t i = 0; i < g; i++) {
                const float s0 = (float)i / (float)g, s1 = (float)(i + 1) / (float)g;
                const float t0 = (float)j / (float)g, t1 = (float)(j + 1) / (float)g;
                const Vec3 p00 = o + s0 * eu + t0 * ev, p10 = o + s1 * eu + t0 * ev;
                const Vec3 p01 = o + s0 * eu + t1 * ev, p11 = o + s1 * eu + t1 * ev;
                tri(p00, p10, p11, m);
                tri(p00, p11, p01, m);
            }
    }
    void box(Vec3 lo, Vec3 hi, uint32_t g, const rt3_material& m) {
        const Vec3 dx(hi.x - lo.x, 0, 0), dy(0, hi.y - lo.y, 0), dz(0, 0, hi.z - lo.z);
        rect(lo, dx, dy, g, m); rect(lo + dz, dx, dy, g, m);
        rect(lo, dz, dy, g, m); rect(lo + dx, dz, dy, g, m);
        rect(lo, dx, dz, g, m); rect(lo + dy, dx, dz, g, m);
    }
};

inline rt3_material mat(uint32_t kind, float r, float g, float b, float param = 0.0f) {
    rt3_material m; m.rgb[0] = r; m.rgb[1] = g; m.rgb[2] = b; m.param = param; m.kind = kind; return m;
}

}  // namespace

extern "C" uint32_t rt3_scene_cornell(uint32_t grid, rt3_gface* faces, float* vertices, rt3_material* face_materials,
                                      uint32_t cap_faces) {
    MeshSink out{ faces, vertices, face_materials, cap_faces, 0 };
    const rt3_material white = mat(RT3_MAT_LAMBERT, 0.73f, 0.73f, 0.73f), red = mat(RT3_MAT_LAMBERT, 0.65f, 0.05f, 0.05f);
    const rt3_material green = mat(RT3_MAT_LAMBERT, 0.12f, 0.45f, 0.15f), light = mat(RT3_MAT_FLAT, 15.0f, 15.0f, 15.0f);
    // box spans x,y in [-1,1], z in [-4,-2]; the camera of Camera::update sits at the origin looking down -z
    out.rect(Vec3(-1, -1, -2), Vec3(0, 0, -2), Vec3(0, 2, 0), grid, red);       // left
    out.rect(Vec3(1, -1, -2), Vec3(0, 0, -2), Vec3(0, 2, 0), grid, green);      // right
    out.rect(Vec3(-1, -1, -2), Vec3(2, 0, 0), Vec3(0, 0, -2), grid, white);     // floor
    out.rect(Vec3(-1, 1, -2), Vec3(2, 0, 0), Vec3(0, 0, -2), grid, white);      // ceiling
    out.rect(Vec3(-1, -1, -4), Vec3(2, 0, 0), Vec3(0, 2, 0), grid, white);      // back
    const uint32_t gb = grid / 4 > 0 ? grid / 4 : 1;
    out.box(Vec3(-0.65f, -1.0f, -3.6f), Vec3(-0.05f, 0.2f, -3.0f), gb, white);  // tall box
    out.box(Vec3(0.1f, -1.0f, -3.0f), Vec3(0.7f, -0.4f, -2.4f), gb, white);     // short box
    out.rect(Vec3(-0.25f, 0.998f, -3.25f), Vec3(0.5f, 0, 0), Vec3(0, 0, 0.5f), 1, light);
    return out.n;
}
