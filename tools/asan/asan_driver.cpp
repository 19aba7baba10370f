// asan_driver.cpp — entry point of the sanitizer build of the HOST side (make -C raytracer-3_amd asan -> raytracer-3_amd/rt3_asan;
// SURVEY.md section 5, "Race detection / sanitizers": the reference's analogue is the Vulkan validation layer, src/lib/compute/Instance.cpp:29-60).
// g++ -fsanitize=address,undefined over csrc/rt3_host.cpp, host/HostApi.cpp, host/sceneparser/SceneParser.cpp, host/Main.cpp and (gcc, same
// flags) oracle/rt3_oracle.c; the device half of the C ABI is tools/asan/device_stubs.cpp.  tests/test_sanitizers.py drives it:
//
//   rt3_asan cli <args ...>      the rt3 command line (host/Main.cpp's main) with these arguments: parse_cli's error table, --dump-scene
//                                over every SceneLang input of tests/test_sceneparser.py
//   rt3_asan selftest <tmpdir>   the host scene API of include/rt3.h against the oracle's restatement of the same functions, byte for byte:
//                                OBJ loader (good, missing, unparsable and out-of-range files), tessellator, merge, cameras, PPM writer with
//                                exact and short buffers, the benchmark scene builders with exact, short and NULL outputs; then the oracle's
//                                Mode-R and Mode-X loops on small frames (shards, progressive accumulation with variance)
// Exit code 0 = every check held and no sanitizer report (a report aborts: -fno-sanitize-recover).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "rt3.h"

int rt3_main(int argc, const char** argv);                          // host/Main.cpp compiled with -Dmain=rt3_main

extern "C" {                                                        // oracle/rt3_oracle.c
void oracle_prerender_triangle(const float*, const float*, const float*, const float*, rt3_gface*, float*);
uint32_t oracle_sphere_face_count(uint32_t, uint32_t);
uint32_t oracle_sphere_vertex_count(uint32_t, uint32_t);
void oracle_prerender_sphere(const float*, float, uint32_t, uint32_t, const float*, rt3_gface*, float*);
int oracle_object_count(const char*, uint32_t*, uint32_t*);
int oracle_prerender_object(const char*, const float*, float, const float*, rt3_gface*, uint32_t, float*, uint32_t);
void oracle_transfer_entity(rt3_gface*, uint32_t*, float*, uint32_t*, const rt3_gface*, uint32_t, const float*, uint32_t);
void oracle_camera_update(rt3_camera*, float, float, float);
uint64_t oracle_frame_ppm_bytes(const uint32_t*, uint32_t, uint32_t, uint8_t*, uint64_t);
void oracle_render_mode_r(const rt3_gface*, uint32_t, const float*, uint32_t, const rt3_camera*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, int);
uint32_t oracle_rows_owned(const rt3_params*);
uint64_t oracle_render_path_range(const rt3_gface*, uint32_t, const float*, const rt3_material*, const float*, const rt3_material*, uint32_t,
                                  const rt3_camera*, const rt3_params*, uint32_t, uint32_t, uint32_t*, float*, float*, int);
uint64_t oracle_render_path(const rt3_gface*, uint32_t, const float*, const rt3_material*, const float*, const rt3_material*, uint32_t,
                            const rt3_camera*, const rt3_params*, uint32_t*, float*, int);
}

namespace {

int failures = 0;
#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed, %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

// Buffers of exactly the size asked for, straight from the heap: an overrun by one element is an ASan report.
template <class T> struct Exact {
    T* p; size_t n;
    explicit Exact(size_t n_) : p(n_ ? (T*)std::malloc(n_ * sizeof(T)) : nullptr), n(n_) { if (n) std::memset(p, 0xA5, n * sizeof(T)); }
    ~Exact() { std::free(p); }
    Exact(const Exact&) = delete;
    size_t bytes() const { return n * sizeof(T); }
};

void write_file(const std::string& path, const std::string& text) { std::ofstream(path, std::ios::binary) << text; }

// A closed fan of `cells` quads around an axis: 2 + 2 cells vertices, 4 cells faces; indices start at `base` (the reference rebases them).
std::string fan_obj(int cells, int base) {
    std::string s;
    char line[128];
    for (int i = 0; i < cells; i++) {
        const double a = 6.283185307179586 * i / cells;
        std::snprintf(line, sizeof line, "v %.6f %.6f %.6f\n", std::cos(a), -0.5, std::sin(a)); s += line;
        std::snprintf(line, sizeof line, "v %.6f %.6f %.6f\n", 0.6 * std::cos(a), 0.75, 0.6 * std::sin(a)); s += line;
    }
    s += "v 0 -0.5 0\nv 0 0.75 0\n";
    for (int i = 0; i < cells; i++) {
        const int a = base + 2 * i, b = base + 2 * ((i + 1) % cells), lo = base + 2 * cells, hi = lo + 1;
        std::snprintf(line, sizeof line, "f %d %d %d\nf %d %d %d\nf %d %d %d\nf %d %d %d\n", a, b, a + 1, b, b + 1, a + 1, lo, b, a, hi, a + 1, b + 1);
        s += line;
    }
    return s;
}

void objects(const std::string& tmp) {
    const float center[3] = { 0.5f, -0.25f, -6.0f }, color[3] = { 1.0f, 0.5f, 0.0f };
    for (int base = 0; base < 2; base++) {
        const std::string path = tmp + "/fan" + std::to_string(base) + ".obj";
        write_file(path, fan_obj(9, base));
        uint32_t nf = 0, nv = 0, onf = 0, onv = 0;
        CHECK(rt3_object_count(path.c_str(), &nf, &nv) == 0 && oracle_object_count(path.c_str(), &onf, &onv) == 0);
        CHECK(nf == 36 && nv == 20 && onf == nf && onv == nv);
        Exact<rt3_gface> f(nf), of(nf);
        Exact<float> v(4 * (size_t)nv), ov(4 * (size_t)nv);
        CHECK(rt3_prerender_object(path.c_str(), center, 0.37f, color, f.p, nf, v.p, nv) == 0);
        CHECK(oracle_prerender_object(path.c_str(), center, 0.37f, color, of.p, nf, ov.p, nv) == 0);
        CHECK(std::memcmp(f.p, of.p, f.bytes()) == 0 && std::memcmp(v.p, ov.p, v.bytes()) == 0);
        // buffers one entity too small: refused, not overrun
        Exact<rt3_gface> f_short(nf - 1);
        Exact<float> v_short(4 * (size_t)(nv - 1));
        CHECK(rt3_prerender_object(path.c_str(), center, 0.37f, color, f_short.p, nf - 1, v.p, nv) != 0);
        CHECK(rt3_prerender_object(path.c_str(), center, 0.37f, color, f.p, nf, v_short.p, nv - 1) != 0);
    }
    uint32_t nf = 0, nv = 0;
    CHECK(rt3_object_count((tmp + "/missing.obj").c_str(), &nf, &nv) != 0);
    // what the reference cannot read (Object.cpp:157-159) and what it would index out of bounds with (Object.cpp:188-194)
    const char* bad[] = { "v 1 2 3\n# a comment line\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -3\n",
                          "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 1e20\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 nan\n", "v 0 0\n", "f 1 2 3\n", "", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 4 5\n",
                          "x 0 0 0\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 4294967295 1 2\n" };
    for (size_t i = 0; i < sizeof bad / sizeof *bad; i++) {
        const std::string path = tmp + "/bad" + std::to_string(i) + ".obj";
        write_file(path, bad[i]);
        nf = nv = 0;
        if (rt3_object_count(path.c_str(), &nf, &nv) != 0) continue;                  // fatal at create_object time
        Exact<rt3_gface> f(nf);
        Exact<float> v(4 * (size_t)nv);
        const float c0[3] = { 0, 0, 0 }, white[3] = { 1, 1, 1 };
        const int rc = rt3_prerender_object(path.c_str(), c0, 1.0f, white, f.p, nf, v.p, nv);
        if (rc == 0)                                                                  // accepted: then every index is inside the vertex array
            for (uint32_t k = 0; k < nf; k++) CHECK(f.p[k].v1 < nv && f.p[k].v2 < nv && f.p[k].v3 < nv);
    }
}

void tessellation() {
    const float center[3] = { -2.0f, 0.25f, -5.0f }, color[3] = { 0.0f, 0.0f, 1.0f };
    const uint32_t grids[][2] = { { 3, 3 }, { 8, 8 }, { 16, 9 }, { 5, 4 }, { 64, 33 } };
    for (const auto& g : grids) {
        const uint32_t nf = rt3_sphere_face_count(g[0], g[1]), nv = rt3_sphere_vertex_count(g[0], g[1]);
        CHECK(nf == oracle_sphere_face_count(g[0], g[1]) && nv == oracle_sphere_vertex_count(g[0], g[1]));
        Exact<rt3_gface> f(nf), of(nf);
        Exact<float> v(4 * (size_t)nv), ov(4 * (size_t)nv);
        rt3_prerender_sphere(center, 1.25f, g[0], g[1], color, f.p, v.p);
        oracle_prerender_sphere(center, 1.25f, g[0], g[1], color, of.p, ov.p);
        CHECK(std::memcmp(f.p, of.p, f.bytes()) == 0 && std::memcmp(v.p, ov.p, v.bytes()) == 0);
        // merge two copies (transfer_entity, SequentialRenderer.cpp:174-195) into buffers of exactly the summed size
        Exact<rt3_gface> mf(2 * (size_t)nf), omf(2 * (size_t)nf);
        Exact<float> mv(8 * (size_t)nv), omv(8 * (size_t)nv);
        uint32_t a = 0, b = 0, oa = 0, ob = 0;
        for (int k = 0; k < 2; k++) {
            rt3_transfer_entity(mf.p, &a, mv.p, &b, f.p, nf, v.p, nv);
            oracle_transfer_entity(omf.p, &oa, omv.p, &ob, of.p, nf, ov.p, nv);
        }
        CHECK(a == 2 * nf && b == 2 * nv && oa == a && ob == b);
        CHECK(std::memcmp(mf.p, omf.p, mf.bytes()) == 0 && std::memcmp(mv.p, omv.p, mv.bytes()) == 0);
        CHECK(mf.p[nf].v1 == f.p[0].v1 + nv);
    }
    const float p1[3] = { 1, 0, -3 }, p2[3] = { -1, 0, -3 }, p3[3] = { 0, 1, -3 }, red[3] = { 1, 0, 0 };
    Exact<rt3_gface> f(1), of(1);
    Exact<float> v(12), ov(12);
    rt3_prerender_triangle(p1, p2, p3, red, f.p, v.p);
    oracle_prerender_triangle(p1, p2, p3, red, of.p, ov.p);
    CHECK(std::memcmp(f.p, of.p, f.bytes()) == 0 && std::memcmp(v.p, ov.p, v.bytes()) == 0);
}

void cameras_and_frames() {
    rt3_camera a, b;
    rt3_camera_update(&a, 2.0f, (400.0f / 225.0f) * 2.0f, 2.0f);
    oracle_camera_update(&b, 2.0f, (400.0f / 225.0f) * 2.0f, 2.0f);
    CHECK(std::memcmp(&a, &b, sizeof a) == 0);
    const float from[3] = { 13, 2, 3 }, at[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
    rt3_camera_look_at(&a, from, at, up, 20.0f, 16.0f / 9.0f, 10.0f);
    CHECK(a.origin[0] == 13.0f && std::isfinite(a.lower_left_corner[2]));
    const uint32_t w = 7, h = 5;
    Exact<uint32_t> px(w * h);
    for (uint32_t i = 0; i < w * h; i++) px.p[i] = 0x01020300u * (i + 1) | 0xFFu;
    const uint64_t need = rt3_frame_ppm_bytes(px.p, w, h, nullptr, 0);
    CHECK(need == oracle_frame_ppm_bytes(px.p, w, h, nullptr, 0) && need > 3 * w * h);
    Exact<uint8_t> out(need), oout(need), small(need - 1);
    CHECK(rt3_frame_ppm_bytes(px.p, w, h, out.p, need) == need && oracle_frame_ppm_bytes(px.p, w, h, oout.p, need) == need);
    CHECK(std::memcmp(out.p, oout.p, need) == 0);
    (void)rt3_frame_ppm_bytes(px.p, w, h, small.p, need - 1);                            // a short buffer must not be overrun, whatever it returns
}

void scenes() {
    {
        const uint32_t n = rt3_scene_three_spheres(nullptr, nullptr, 0);
        Exact<float> cr(4 * (size_t)n); Exact<rt3_material> m(n);
        CHECK(n == 3 && rt3_scene_three_spheres(cr.p, m.p, n) == n);
    }
    {
        const uint32_t n = rt3_scene_weekend(42, nullptr, nullptr, 0);
        Exact<float> cr(4 * (size_t)n), part(4 * 10); Exact<rt3_material> m(n), mpart(10);
        CHECK(n > 400 && n < 500 && rt3_scene_weekend(42, cr.p, m.p, n) == n);
        CHECK(rt3_scene_weekend(42, part.p, mpart.p, 10) == n);                          // a short buffer is filled up to cap, not beyond; the count is the one required
        CHECK(std::memcmp(part.p, cr.p, part.bytes()) == 0);
    }
    {
        Exact<float> cr(4 * 1000); Exact<rt3_material> m(1000);
        CHECK(rt3_scene_stress(1000, 43, nullptr, nullptr, 0) == 1000 && rt3_scene_stress(1000, 43, cr.p, m.p, 1000) == 1000);
        Exact<float> part(4 * 999); Exact<rt3_material> mpart(999);
        CHECK(rt3_scene_stress(1000, 43, part.p, mpart.p, 999) == 1000);
    }
    {
        const uint32_t nf = rt3_scene_cornell(4, nullptr, nullptr, nullptr, 0);
        Exact<rt3_gface> f(nf); Exact<float> v(12 * (size_t)nf); Exact<rt3_material> m(nf);
        CHECK(nf > 100 && rt3_scene_cornell(4, f.p, v.p, m.p, nf) == nf);
        for (uint32_t k = 0; k < nf; k++) CHECK(f.p[k].v1 < 3 * nf && f.p[k].v2 < 3 * nf && f.p[k].v3 < 3 * nf && m.p[k].kind <= RT3_MAT_DIELECTRIC);
        Exact<rt3_gface> f1(nf - 1); Exact<float> v1(12 * (size_t)(nf - 1)); Exact<rt3_material> m1(nf - 1);
        CHECK(rt3_scene_cornell(4, f1.p, v1.p, m1.p, nf - 1) == nf);
    }
    CHECK(rt3_hash_u32(1) == 0x124ea49du && rt3_hash_u32(0xFFFFFFFFu) == 0xae65a494u);  // SURVEY.md section 8c known answers
    CHECK(rt3_random_float(0x3F800000u) >= 0.0f && rt3_random_float(0xFFFFFFFFu) < 1.0f);
}

void oracle_loops() {
    // Mode R: a tessellated sphere + a triangle, every row (the caller of the reference loop chooses the rows)
    const float center[3] = { -0.5f, 0.0f, -4.0f }, blue[3] = { 0, 0, 1 }, red[3] = { 1, 0, 0 };
    const uint32_t nfs = rt3_sphere_face_count(8, 6), nvs = rt3_sphere_vertex_count(8, 6);
    Exact<rt3_gface> sf(nfs), tf(1), mf(nfs + 1);
    Exact<float> sv(4 * (size_t)nvs), tv(12), mv(4 * (size_t)(nvs + 3));
    rt3_prerender_sphere(center, 1.0f, 8, 6, blue, sf.p, sv.p);
    const float p1[3] = { 1.5f, -0.5f, -3 }, p2[3] = { 0.2f, -0.5f, -3 }, p3[3] = { 0.8f, 0.9f, -3.5f };
    rt3_prerender_triangle(p1, p2, p3, red, tf.p, tv.p);
    uint32_t nf = 0, nv = 0;
    rt3_transfer_entity(mf.p, &nf, mv.p, &nv, sf.p, nfs, sv.p, nvs);
    rt3_transfer_entity(mf.p, &nf, mv.p, &nv, tf.p, 1, tv.p, 3);
    const uint32_t W = 48, H = 27;
    rt3_camera cam;
    rt3_camera_update(&cam, 2.0f, ((float)W / (float)H) * 2.0f, 2.0f);
    Exact<uint32_t> img(W * H);
    oracle_render_mode_r(mf.p, nf, mv.p, nv, &cam, W, H, 0, H, img.p, 2);
    uint32_t hit = 0;
    for (uint32_t i = 0; i < W * H; i++) hit += (img.p[i] >> 24) == 0 || ((img.p[i] >> 8) & 0xFF) == 0 ? 1u : 0u;
    CHECK(hit > 20);                                                                      // both entities are in view
    // Mode X: the mesh with materials and the three-sphere scene together, a shard, then the progressive form with variance
    Exact<rt3_material> fm(nf);
    for (uint32_t k = 0; k < nf; k++) { fm.p[k].rgb[0] = 0.7f; fm.p[k].rgb[1] = 0.6f; fm.p[k].rgb[2] = 0.5f; fm.p[k].param = k % 3 == 2 ? 0.2f : 0.0f; fm.p[k].kind = 1 + k % 2; }
    Exact<float> cr(12); Exact<rt3_material> sm(3);
    rt3_scene_three_spheres(cr.p, sm.p, 3);
    sm.p[1].kind = RT3_MAT_DIELECTRIC; sm.p[1].param = 1.5f;
    rt3_params P = { W, H, 9, 8, 7, RT3_FLAG_GAMMA2 | RT3_FLAG_VARIANCE, 0.05f, 0.001f, 4, 1, 3 };
    const uint32_t rows = oracle_rows_owned(&P);
    CHECK(rows == rt3_rows_owned(&P) && rows > 0 && rows < H);
    Exact<uint32_t> whole(rows * W), parts(rows * W);
    Exact<float> sum3(3 * (size_t)rows * W), sum4(4 * (size_t)rows * W), sq4(4 * (size_t)rows * W);
    const uint64_t casts = oracle_render_path(mf.p, nf, mv.p, fm.p, cr.p, sm.p, 3, &cam, &P, whole.p, sum3.p, 2);
    uint64_t casts2 = 0;
    for (uint32_t s = 0; s < 9; s += 4) casts2 += oracle_render_path_range(mf.p, nf, mv.p, fm.p, cr.p, sm.p, 3, &cam, &P, s, s + 4 <= 9 ? 4 : 9 - s, parts.p, sum4.p, sq4.p, 2);
    CHECK(casts == casts2 && casts >= (uint64_t)rows * W * 9);
    CHECK(std::memcmp(whole.p, parts.p, whole.bytes()) == 0);
    for (size_t i = 0; i < (size_t)rows * W; i++) CHECK(sum3.p[3 * i] == sum4.p[4 * i] && sq4.p[4 * i] >= 0.0f);
    // the reference-primary corner: Mode X == Mode R byte for byte (consequence 1(i))
    rt3_params R = { W, H, 1, 1, 1, RT3_FLAG_REFERENCE_PRIMARY, 0.0f, 0.0f, 8, 0, 1 };
    Exact<uint32_t> x(W * H);
    oracle_render_path(mf.p, nf, mv.p, nullptr, nullptr, nullptr, 0, &cam, &R, x.p, nullptr, 2);
    CHECK(std::memcmp(x.p, img.p, x.bytes()) == 0);
}

}  // namespace

int main(int argc, const char** argv) {
    if (argc >= 2 && std::strcmp(argv[1], "cli") == 0) {
        std::vector<const char*> args;
        args.push_back(argv[0]);
        for (int i = 2; i < argc; i++) args.push_back(argv[i]);
        return rt3_main((int)args.size(), args.data());
    }
    if (argc == 3 && std::strcmp(argv[1], "selftest") == 0) {
        const std::string tmp = argv[2];
        objects(tmp);
        tessellation();
        cameras_and_frames();
        scenes();
        oracle_loops();
        std::printf("rt3_asan selftest: %s (%d failed checks)\n", failures ? "FAILED" : "ok", failures);
        return failures ? 1 : 0;
    }
    std::fprintf(stderr, "usage: %s cli <rt3 arguments ...> | selftest <tmpdir>\n", argv[0]);
    return 2;
}
