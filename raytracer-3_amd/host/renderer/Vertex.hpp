// renderer/Vertex.hpp — RayTracer::GFace, the indexed face record handed to the kernels
// (reference: src/lib/renderer/Vertex.hpp:39-51).  Same 48-byte layout as rt3_gface in include/rt3.h.
#ifndef RT3_HOST_VERTEX_HPP
#define RT3_HOST_VERTEX_HPP
#include <cstdint>
#include "glm/glm.hpp"
#include "rt3.h"

namespace RayTracer {
struct GFace {
    alignas(4) uint32_t v1, v2, v3;
    alignas(16) glm::vec3 normal;
    alignas(16) glm::vec3 color;
};
static_assert(sizeof(GFace) == sizeof(rt3_gface) && sizeof(GFace) == 48, "GFace must match the C ABI");
}  // namespace RayTracer
#endif
