// HostApi.cpp — C++ mirror of the reference's host interface for the render path, implemented over the C ABI.
// (Camera.cpp, Frame.cpp, entities/*.cpp, the Renderer backend of the reference; see the headers for file:line.)
#include <cmath>
#include <cstring>
#include <fstream>
#include <thread>

#include "renderer/Renderer.hpp"

using namespace RayTracer;
using namespace RayTracer::ECS;

// ---------------------------------------------------------------------------------------------------- camera
void Camera::update(uint32_t width, uint32_t height, float focal_length, float viewport_width, float viewport_height) {
    frame = std::make_shared<Frame>(width, height);
    rt3_camera c;
    rt3_camera_update(&c, focal_length, viewport_width, viewport_height);
    origin = glm::vec3(c.origin[0], c.origin[1], c.origin[2]);
    horizontal = glm::vec3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    vertical = glm::vec3(c.vertical[0], c.vertical[1], c.vertical[2]);
    lower_left_corner = glm::vec3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
}

void Camera::look_at(uint32_t width, uint32_t height, const glm::vec3& from, const glm::vec3& at, const glm::vec3& vup,
                     float vfov_deg, float focus_dist) {
    frame = std::make_shared<Frame>(width, height);
    rt3_camera c;
    rt3_camera_look_at(&c, from.ptr(), at.ptr(), vup.ptr(), vfov_deg, (float)width / (float)height, focus_dist);
    origin = glm::vec3(c.origin[0], c.origin[1], c.origin[2]);
    horizontal = glm::vec3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    vertical = glm::vec3(c.vertical[0], c.vertical[1], c.vertical[2]);
    lower_left_corner = glm::vec3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
}

rt3_camera Camera::wire() const {
    rt3_camera c;
    std::memcpy(c.origin, origin.ptr(), 12);
    std::memcpy(c.horizontal, horizontal.ptr(), 12);
    std::memcpy(c.vertical, vertical.ptr(), 12);
    std::memcpy(c.lower_left_corner, lower_left_corner.ptr(), 12);
    return c;
}

// ---------------------------------------------------------------------------------------------------- frame
void Frame::to_ppm(const std::string& path) const {
    if (rt3_frame_to_ppm(d(), width, height, path.c_str()) != 0) throw Fatal("Could not open '" + path + "'");
}

namespace {
uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    if (!table[1])
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
void put_chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& body) {
    put_be32(out, (uint32_t)body.size());
    std::vector<uint8_t> t(tag, tag + 4);
    t.insert(t.end(), body.begin(), body.end());
    out.insert(out.end(), t.begin(), t.end());
    put_be32(out, crc32_of(t.data(), t.size()));
}
}  // namespace

// 8-bit RGBA PNG with stored (uncompressed) deflate blocks: same pixels as the reference's LodePNG output
// (Frame.cpp:88-96: r = >>24, g = >>16, b = >>8, a = 255), different compression.
void Frame::to_png(const std::string& path) const {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + 4 * (size_t)width));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        for (uint32_t x = 0; x < width; x++) {
            const uint32_t px = d()[(size_t)y * width + x];
            raw.push_back((uint8_t)(px >> 24)); raw.push_back((uint8_t)(px >> 16)); raw.push_back((uint8_t)(px >> 8)); raw.push_back(255);
        }
    }
    std::vector<uint8_t> z = { 0x78, 0x01 };
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521; b = (b + a) % 65521; }
        if (raw.empty()) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' }, ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.insert(ihdr.end(), { 8, 6, 0, 0, 0 });
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw Fatal("Could not write to PNG: cannot open '" + path + "'");
    f.write((const char*)out.data(), (std::streamsize)out.size());
}

// ---------------------------------------------------------------------------------------------------- entities
Triangle* ECS::create_triangle(const glm::vec3& p1, const glm::vec3& p2, const glm::vec3& p3, const glm::vec3& color) {
    Triangle* t = new Triangle;
    t->type = et_triangle; t->pre_render_mode = eprmf_cpu; t->pre_render_operation = epro_generate_triangle;
    t->pre_render_faces = 1; t->pre_render_vertices = 3;
    t->points[0] = p1; t->points[1] = p2; t->points[2] = p3; t->color = color;
    rt3_gface f; float v[12];
    rt3_prerender_triangle(p1.ptr(), p2.ptr(), p3.ptr(), color.ptr(), &f, v);      // the normal of Triangle.cpp:48
    t->normal = glm::vec3(f.normal[0], f.normal[1], f.normal[2]);
    return t;
}

Sphere* ECS::create_sphere(const glm::vec3& center, float radius, uint32_t n_meridians, uint32_t n_parallels, const glm::vec3& color) {
    Sphere* s = new Sphere;
    s->center = center; s->radius = radius; s->n_meridians = n_meridians; s->n_parallels = n_parallels; s->color = color;
    if (n_meridians == 0 && n_parallels == 0) { s->type = et_analytic_sphere; return s; }
    s->type = et_sphere; s->pre_render_mode = eprmf_cpu | eprmf_gpu; s->pre_render_operation = epro_generate_sphere;   // Sphere.cpp:94-98
    s->pre_render_faces = rt3_sphere_face_count(n_meridians, n_parallels);
    s->pre_render_vertices = rt3_sphere_vertex_count(n_meridians, n_parallels);
    return s;
}

Object* ECS::create_object(const std::string& file_path, const glm::vec3& center, float scale, const glm::vec3& color) {
    Object* o = new Object;
    o->type = et_object; o->pre_render_mode = eprmf_cpu; o->pre_render_operation = epro_load_object_file;
    o->file_path = file_path; o->center = center; o->scale = scale; o->color = color;
    if (rt3_object_count(file_path.c_str(), &o->pre_render_faces, &o->pre_render_vertices) != 0) {
        delete o;
        throw Fatal("Could not open file or encountered unreadable line: " + file_path);
    }
    return o;
}

static rt3_material make_material(uint32_t kind, const glm::vec3& rgb, float param) {
    rt3_material m; m.rgb[0] = rgb.x; m.rgb[1] = rgb.y; m.rgb[2] = rgb.z; m.param = param; m.kind = kind; return m;
}
rt3_material ECS::lambertian(const glm::vec3& albedo) { return make_material(RT3_MAT_LAMBERT, albedo, 0.0f); }
rt3_material ECS::metal(const glm::vec3& albedo, float fuzz) { return make_material(RT3_MAT_METAL, albedo, fuzz); }
rt3_material ECS::dielectric(float ior) { return make_material(RT3_MAT_DIELECTRIC, glm::vec3(1.0f), ior); }
rt3_material ECS::emissive(const glm::vec3& radiance) { return make_material(RT3_MAT_FLAT, radiance, 0.0f); }

// ---------------------------------------------------------------------------------------------------- backend
HipRenderer::HipRenderer(const std::vector<int>& devices) {
    for (int dev : devices) {
        rt3_ctx* c = rt3_create(dev);
        if (!c) {
            const std::string why = rt3_last_error(nullptr);
            for (rt3_ctx* made : ctx) rt3_destroy(made);
            throw Fatal(why);
        }
        ctx.push_back(c);
    }
    if (ctx.empty()) throw Fatal("HipRenderer needs at least one device");
}

HipRenderer::~HipRenderer() { for (rt3_ctx* c : ctx) rt3_destroy(c); }

void HipRenderer::prerender(const Tools::Array<RenderEntity*>& entities) {
    // VulkanRenderer::prerender's shape (VulkanRenderer.cpp:266-399): size the device buffers from the counts every entity
    // declares up front (RenderEntity.hpp:84-88), then fill them entity by entity at running offsets — tessellated on the
    // device when the entity allows it and gpu_prerender is on (:305-327), else pre-rendered here and transferred (:355-387).
    uint32_t total_f = 0, total_v = 0;
    bool any_material = false;
    for (size_t i = 0; i < entities.size(); i++) {
        if (entities[i]->type == et_analytic_sphere) continue;
        total_f += entities[i]->pre_render_faces; total_v += entities[i]->pre_render_vertices;
        any_material = any_material || entities[i]->has_material;
    }
    for (rt3_ctx* c : ctx) if (rt3_mesh_begin(c, total_f, total_v) != 0) throw Fatal(rt3_last_error(c));

    std::vector<rt3_gface> tmp_f;
    std::vector<float> tmp_v, sph;
    std::vector<rt3_material> face_mats, sph_mats;
    uint32_t fo = 0, vo = 0;
    for (size_t i = 0; i < entities.size(); i++) {
        RenderEntity* e = entities[i];
        if (e->type == et_analytic_sphere) {
            const Sphere* s = static_cast<const Sphere*>(e);
            sph.insert(sph.end(), { s->center.x, s->center.y, s->center.z, s->radius });
            sph_mats.push_back(e->has_material ? e->material : emissive(s->color));
            continue;
        }
        const bool on_device = gpu_prerender && (e->pre_render_mode & eprmf_gpu) && e->pre_render_operation == epro_generate_sphere &&
                               (!any_material || e->has_material);
        if (on_device) {
            const Sphere* s = static_cast<const Sphere*>(e);
            for (rt3_ctx* c : ctx)
                if (rt3_mesh_sphere(c, s->center.ptr(), s->radius, s->n_meridians, s->n_parallels, s->color.ptr(), fo, vo) != 0) throw Fatal(rt3_last_error(c));
        } else {
            if (!(e->pre_render_mode & eprmf_cpu))
                throw Fatal("Entity " + std::to_string(i) + " of type " + entity_type_names[e->type] + " cannot be pre-rendered by this back-end.");
            tmp_f.assign(e->pre_render_faces, rt3_gface{});
            tmp_v.assign(4 * (size_t)e->pre_render_vertices, 0.0f);
            switch (e->pre_render_operation) {
                case epro_generate_triangle: {
                    const Triangle* t = static_cast<const Triangle*>(e);
                    rt3_prerender_triangle(t->points[0].ptr(), t->points[1].ptr(), t->points[2].ptr(), t->color.ptr(), tmp_f.data(), tmp_v.data());
                    break;
                }
                case epro_generate_sphere: {
                    const Sphere* s = static_cast<const Sphere*>(e);
                    rt3_prerender_sphere(s->center.ptr(), s->radius, s->n_meridians, s->n_parallels, s->color.ptr(), tmp_f.data(), tmp_v.data());
                    break;
                }
                case epro_load_object_file: {
                    const Object* o = static_cast<const Object*>(e);
                    if (rt3_prerender_object(o->file_path.c_str(), o->center.ptr(), o->scale, o->color.ptr(), tmp_f.data(),
                                             e->pre_render_faces, tmp_v.data(), e->pre_render_vertices) != 0)
                        throw Fatal("Could not load object file '" + o->file_path + "'");
                    break;
                }
                default:
                    throw Fatal("Entity " + std::to_string(i) + " wants to be pre-rendered using unsupported operation '" +
                                entity_pre_render_operation_names[e->pre_render_operation] + "'.");
            }
            for (rt3_ctx* c : ctx)
                if (rt3_mesh_put(c, tmp_f.data(), e->pre_render_faces, tmp_v.data(), e->pre_render_vertices, fo, vo) != 0) throw Fatal(rt3_last_error(c));
        }
        if (any_material)
            for (uint32_t k = 0; k < e->pre_render_faces; k++)
                face_mats.push_back(e->has_material ? e->material
                                                    : emissive(glm::vec3(tmp_f[k].color[0], tmp_f[k].color[1], tmp_f[k].color[2])));
        fo += e->pre_render_faces;
        vo += e->pre_render_vertices;
    }
    for (rt3_ctx* c : ctx) {
        if (rt3_mesh_commit(c, any_material ? face_mats.data() : nullptr) != 0) throw Fatal(rt3_last_error(c));
        if (rt3_set_spheres(c, sph.data(), sph_mats.data(), (uint32_t)sph_mats.size()) != 0) throw Fatal(rt3_last_error(c));
    }
    n_faces = fo;
    n_spheres = sph_mats.size();
}

void HipRenderer::set_spheres(const std::vector<float>& center_radius, const std::vector<rt3_material>& materials) {
    for (rt3_ctx* c : ctx)
        if (rt3_set_spheres(c, center_radius.data(), materials.data(), (uint32_t)materials.size()) != 0) throw Fatal(rt3_last_error(c));
    n_spheres = materials.size();
}

void HipRenderer::set_mesh(const std::vector<rt3_gface>& faces, const std::vector<float>& vertices, const std::vector<rt3_material>& face_materials) {
    for (rt3_ctx* c : ctx)
        if (rt3_set_mesh(c, faces.data(), (uint32_t)faces.size(), vertices.data(), (uint32_t)(vertices.size() / 4),
                         face_materials.empty() ? nullptr : face_materials.data()) != 0) throw Fatal(rt3_last_error(c));
    n_faces = faces.size();
}

void HipRenderer::render(Camera& camera) const {
    const rt3_camera cam = camera.wire();
    const uint32_t w = camera.w(), h = camera.h();
    uint32_t* out = camera.get_frame().d();
    if (path.spp == 0) {                            // Mode R: the reference's own render, one GPU
        if (rt3_render(ctx[0], &cam, w, h, out) != 0) throw Fatal(rt3_last_error(ctx[0]));
        return;
    }
    // Mode X: interleaved row blocks over the GPUs (tiling intent: BlockInfo, raytracer_v4.glsl:70-79).  Every device renders its rows
    // into a tile in its OWN memory; the tiles then travel device to device into device 0's frame at their interleaved row offsets
    // (rt3_gather_rows: one strided peer copy per device over xGMI, queued on the rendering stream right behind the resolve kernel, so
    // a device that finishes early ships its rows while the others still trace) and the assembled frame leaves device 0 in one copy.
    // No pixel passes through host memory before that.  One host thread per device only issues the (asynchronous) work.
    const uint32_t n = (uint32_t)ctx.size();
    if (n == 1) {
        rt3_params p{ w, h, path.spp, path.max_depth, path.seed, path.flags, path.lens_radius, path.t_min, path.tile_rows, 0, 1 };
        if (rt3_render_path(ctx[0], &cam, &p, out) != 0) throw Fatal(rt3_last_error(ctx[0]));
        return;
    }
    struct DeviceBuffer {                                // freed on every way out
        rt3_ctx* c; void* p;
        ~DeviceBuffer() { rt3_device_free(c, p); }
    };
    DeviceBuffer frame{ ctx[0], rt3_device_alloc_words(ctx[0], (uint64_t)w * h) };
    if (!frame.p) throw Fatal(rt3_last_error(ctx[0]));
    std::vector<std::string> errors(n);
    std::vector<std::thread> workers;
    for (uint32_t i = 0; i < n; i++) {
        workers.emplace_back([&, i]() {
            rt3_params p{ w, h, path.spp, path.max_depth, path.seed, path.flags, path.lens_radius, path.t_min, path.tile_rows, i, n };
            const uint32_t rows = rt3_rows_owned(&p);
            if (rows == 0) return;
            DeviceBuffer tile{ ctx[i], rt3_device_alloc_words(ctx[i], (uint64_t)rows * w) };
            if (!tile.p) { errors[i] = rt3_last_error(ctx[i]); return; }
            if (rt3_render_path_device(ctx[i], &cam, &p, tile.p, rt3_stream(ctx[i])) != 0 ||
                rt3_gather_rows(ctx[0], frame.p, ctx[i], tile.p, &p, rt3_stream(ctx[i])) != 0 ||
                rt3_synchronize(ctx[i]) != 0)
                errors[i] = rt3_last_error(ctx[i]);
        });
    }
    for (std::thread& t : workers) t.join();
    for (const std::string& e : errors) if (!e.empty()) throw Fatal(e);
    if (rt3_device_read_words(ctx[0], frame.p, (uint64_t)w * h, out) != 0) throw Fatal(rt3_last_error(ctx[0]));
}

rt3_stats HipRenderer::stats() const {
    rt3_stats s;
    if (rt3_get_stats(ctx[0], &s) != 0) throw Fatal(rt3_last_error(ctx[0]));
    return s;
}

Renderer* RayTracer::initialize_renderer() { return new HipRenderer(); }
