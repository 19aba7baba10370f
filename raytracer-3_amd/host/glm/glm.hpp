// glm/glm.hpp — the sliver of GLM the reference's public headers expose (glm::vec3 / glm::vec4 in entity,
// camera and GFace signatures).  Build-owned; the reference vendors GLM 0.9.9.8 under src/lib/glm, which is not
// copied.  Only storage and brace construction are needed here: all arithmetic on these types happens behind the
// C ABI (librt3hip.so), with the evaluation order documented in DESIGN.md §3.
#ifndef RT3_HOST_GLM_HPP
#define RT3_HOST_GLM_HPP
namespace glm {
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    vec3(double x_, double y_, double z_) : x((float)x_), y((float)y_), z((float)z_) {}
    const float* ptr() const { return &x; }
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(const vec3& v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    operator vec3() const { return vec3(x, y, z); }
};
}  // namespace glm
#endif
