// camera/Frame.hpp — RayTracer::Frame: the uint32 RGBA8 frame a render fills (reference: src/lib/camera/Frame.hpp:41-82).
#ifndef RT3_HOST_FRAME_HPP
#define RT3_HOST_FRAME_HPP
#include <cstdint>
#include <string>
#include <vector>

namespace RayTracer {
struct Fatal : public std::runtime_error {          // stands where the reference throws CppDebugger::Fatal
    explicit Fatal(const std::string& what) : std::runtime_error(what) {}
};

class Frame {
    mutable std::vector<uint32_t> pixels;           // d() hands out a mutable pointer from a const Frame (Frame.hpp:70)
    uint32_t width, height;

public:
    Frame(uint32_t w_, uint32_t h_) : pixels((size_t)w_ * h_), width(w_), height(h_) {}
    uint32_t w() const { return width; }
    uint32_t h() const { return height; }
    uint32_t* d() const { return pixels.data(); }
    void to_ppm(const std::string& path) const;     // Frame.cpp:109-148
    void to_png(const std::string& path) const;     // Frame.cpp:82-106 (stored-deflate PNG, no LodePNG)
};
}  // namespace RayTracer
#endif
