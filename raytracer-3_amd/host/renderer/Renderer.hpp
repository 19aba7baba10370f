// renderer/Renderer.hpp — the backend boundary of the reference (src/lib/renderer/Renderer.hpp:34-63): an abstract
// Renderer with prerender() / render() and the link-time factory initialize_renderer().  HipRenderer is the
// MI355X backend; it drives librt3hip.so through the C ABI of include/rt3.h.
#ifndef RT3_HOST_RENDERER_HPP
#define RT3_HOST_RENDERER_HPP
#include <cstdint>
#include <vector>
#include "camera/Camera.hpp"
#include "entities/RenderEntity.hpp"
#include "tools/Array.hpp"
#include "Vertex.hpp"
#include "rt3.h"

namespace RayTracer {
class Renderer {
public:
    virtual ~Renderer() = default;
    virtual void prerender(const Tools::Array<ECS::RenderEntity*>& entities) = 0;
    virtual void render(Camera& camera) const = 0;
};

// Mode-X knobs the reference API has no place for (spp / depth / seed ...); spp == 0 means Mode R.
struct PathOptions {
    uint32_t spp = 0, max_depth = 50, seed = 1, flags = 0;
    float lens_radius = 0.0f, t_min = 0.001f;
    uint32_t tile_rows = 8;
};

class HipRenderer : public Renderer {
public:
    explicit HipRenderer(const std::vector<int>& devices = { 0 });
    ~HipRenderer() override;
    void prerender(const Tools::Array<ECS::RenderEntity*>& entities) override;   // flatten in entity order + upload
    void render(Camera& camera) const override;                                  // fills camera.get_frame().d()
    void configure(const PathOptions& options) { path = options; }
    void set_gpu_prerender(bool on) { gpu_prerender = on; }         // tessellate eprmf_gpu entities on the device (default: host)
    // scenes that are not entity lists (benchmark sphere fields)
    void set_spheres(const std::vector<float>& center_radius, const std::vector<rt3_material>& materials);
    void set_mesh(const std::vector<rt3_gface>& faces, const std::vector<float>& vertices_xyzw, const std::vector<rt3_material>& face_materials);
    rt3_stats stats() const;                                                     // of device 0's last render
    size_t faces() const { return n_faces; }
    size_t spheres() const { return n_spheres; }

private:
    std::vector<rt3_ctx*> ctx;                      // one device context per GPU; frame rows are sharded over them
    PathOptions path;
    bool gpu_prerender = false;
    size_t n_faces = 0, n_spheres = 0;
};

Renderer* initialize_renderer();                    // Renderer.hpp:63 — always the HIP backend here
}  // namespace RayTracer
#endif
