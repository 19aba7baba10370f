// camera/Camera.hpp — RayTracer::Camera: four public vectors + an owned Frame (reference: src/lib/camera/Camera.hpp:25-68).
#ifndef RT3_HOST_CAMERA_HPP
#define RT3_HOST_CAMERA_HPP
#include <memory>
#include "glm/glm.hpp"
#include "Frame.hpp"
#include "rt3.h"

namespace RayTracer {
class Camera {
public:
    glm::vec3 origin, horizontal, vertical, lower_left_corner;

    // Camera::update (Camera.cpp:77-96): origin 0, axis-aligned viewport, fresh Frame(width, height)
    void update(uint32_t width, uint32_t height, float focal_length, float viewport_width, float viewport_height);
    // extension: book-style look-from / look-at (the reference camera cannot move)
    void look_at(uint32_t width, uint32_t height, const glm::vec3& from, const glm::vec3& at, const glm::vec3& vup,
                 float vfov_deg, float focus_dist);

    uint32_t w() const { return frame->w(); }
    uint32_t h() const { return frame->h(); }
    const Frame& get_frame() const { return *frame; }
    rt3_camera wire() const;                        // the four vectors as the C ABI takes them

private:
    std::shared_ptr<Frame> frame;
};
}  // namespace RayTracer
#endif
