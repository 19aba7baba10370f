// entities/RenderEntity.hpp — the ECS-style entity records (reference: src/lib/entities/RenderEntity.hpp:24-89,
// Triangle.hpp:28-35, Sphere.hpp:34-45, Object.hpp:29-38) and their create_* factories.  Extension: an optional
// material (Mode X); without one an entity renders with the reference's flat baked colour.
#ifndef RT3_HOST_RENDER_ENTITY_HPP
#define RT3_HOST_RENDER_ENTITY_HPP
#include <cstdint>
#include <string>
#include "glm/glm.hpp"
#include "rt3.h"

namespace RayTracer::ECS {
enum EntityType { et_none = 0, et_triangle = 1, et_sphere = 2, et_object = 3, et_analytic_sphere = 4 /* extension */ };
enum EntityPreRenderModeFlags { eprmf_none = 0x0, eprmf_cpu = 0x1, eprmf_gpu = 0x2 };
enum EntityPreRenderOperation { epro_none = 0, epro_generate_triangle = 1, epro_generate_sphere = 2, epro_load_object_file = 3 };
static const std::string entity_type_names[] = { "none", "triangle", "sphere", "object", "analytic_sphere" };
static const std::string entity_pre_render_operation_names[] = { "none", "generate_triangle", "generate_sphere", "load_object_file" };

struct RenderEntity {
    EntityType type = et_none;
    unsigned int pre_render_mode = eprmf_none;
    EntityPreRenderOperation pre_render_operation = epro_none;
    uint32_t pre_render_faces = 0;          // known BEFORE pre-rendering: the backend sizes its buffers with them
    uint32_t pre_render_vertices = 0;
    bool has_material = false;              // extension
    rt3_material material{};
    virtual ~RenderEntity() = default;
};
struct Triangle : RenderEntity { glm::vec3 points[3]; glm::vec3 normal; glm::vec3 color; };
struct Sphere : RenderEntity { glm::vec3 center; float radius; uint32_t n_meridians, n_parallels; glm::vec3 color; };
struct Object : RenderEntity { std::string file_path; glm::vec3 center; float scale; glm::vec3 color; };

Triangle* create_triangle(const glm::vec3& p1, const glm::vec3& p2, const glm::vec3& p3, const glm::vec3& color);
// n_meridians == n_parallels == 0 asks for an analytic sphere (extension) instead of a tessellation
Sphere* create_sphere(const glm::vec3& center, float radius, uint32_t n_meridians, uint32_t n_parallels, const glm::vec3& color);
Object* create_object(const std::string& file_path, const glm::vec3& center, float scale, const glm::vec3& color);

rt3_material lambertian(const glm::vec3& albedo);
rt3_material metal(const glm::vec3& albedo, float fuzz);
rt3_material dielectric(float ior);
rt3_material emissive(const glm::vec3& radiance);
template <class E> E* with_material(E* e, const rt3_material& m) { e->has_material = true; e->material = m; return e; }
}  // namespace RayTracer::ECS
#endif
