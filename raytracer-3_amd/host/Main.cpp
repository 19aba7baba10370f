// Main.cpp — the rt3 command line: same flags, defaults, messages and exit codes as the reference's entry point
// (src/Main.cpp:62-81 options, :89-239 parsing, :246-315 main), driving the HIP backend through the reference's own
// call sequence: initialize_renderer() -> Camera::update -> create_* -> prerender -> render -> Frame::to_ppm/to_png.
//
// Kept: -f/--format png|ppm (default png), -W/--width (800), -H/--height (600), -h/--help, first positional =
// output path (later ones ignored), value forms `-W 400`, `-W400`, `--width 400`; exit code 0 after help, -1 on a
// usage error, -1 on a fatal backend error.  Fixed: `--key=value`, which the reference mis-parses (Main.cpp:110).
// Added (defaults reproduce the reference render): --scene, --spp, --depth, --seed, --gpus.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "renderer/Renderer.hpp"
#include "sceneparser/SceneParser.hpp"

using namespace RayTracer;

namespace {

struct Options {
    std::string output_path;
    bool png = true;
    uint32_t width = 800, height = 600;
    std::string scene = "builtin";
    uint32_t spp = 0, depth = 50, seed = 1, gpus = 1;
    bool gpu_prerender = false, dump_scene = false;
};

void print_usage(const char* exe) {
    std::cout << "Usage: " << exe << " [<options>] <output_path>\n\n"
              << "Options:\n"
              << "\t-f,--format\tThe format of the resulting frame. Supported formats are: 'png' and 'ppm' (default: png).\n"
              << "\t-W,--width\tThe width of the resulting image, in pixels (default: 800).\n"
              << "\t-H,--height\tThe height of th resulting image, in pixels (default: 600).\n"
              << "\t   --scene\tbuiltin | three | weekend | stress100k | cornell | <file>.scene (default: builtin = src/Main.cpp's teddy + sphere).\n"
              << "\t   --spp\tSamples per pixel; enables the path tracer (default: off = the reference's 1-ray render).\n"
              << "\t   --depth\tMaximum ray casts per path (default: 50).\n"
              << "\t   --seed\tRender seed (default: 1).\n"
              << "\t   --gpus\tNumber of GPUs to shard the frame over (default: 1).\n"
              << "\t   --gpu-prerender\tTessellate spheres on the GPU instead of the host (same arrays).\n"
              << "\t   --dump-scene\tParse the --scene file, print its entities and exit.\n"
              << "\n\t-h,--help\tShows this help menu, then exits.\n\n";
}

// Parses an unsigned option value; prints the reference's message and returns false on failure.
bool parse_u32(const std::string& text, const char* what_lower, const char* what_cap, uint32_t* out) {
    try {
        size_t used = 0;
        const unsigned long v = std::stoul(text, &used);
        if (v > std::numeric_limits<uint32_t>::max()) { std::cerr << what_cap << " too large '" + text + "'"; return false; }
        *out = (uint32_t)v;
        return true;
    } catch (std::invalid_argument&) {
        std::cerr << "Invalid " << what_lower << " '" + text + "'";
    } catch (std::out_of_range&) {
        std::cerr << what_cap << " too large '" + text + "'";
    }
    return false;
}

// 1 = go on, 0 = help was shown, -1 = usage error (the three return values of the reference's parse_cli)
int parse_cli(Options& opt, int argc, const char** argv) {
    bool have_path = false;
    for (int i = 1; i < argc; i++) {
        const std::string arg = argv[i];
        if (arg.empty() || arg[0] != '-') {
            if (!have_path) { opt.output_path = arg; have_path = true; }
            continue;                                               // extra positionals are ignored (Main.cpp:218-228)
        }
        // split "-Wvalue" / "--key=value" / "--key value"
        std::string key = arg, value;
        if (arg.size() > 1 && arg[1] != '-') { key = arg.substr(0, 2); value = arg.substr(2); }
        else if (arg.find('=') != std::string::npos) { key = arg.substr(0, arg.find('=')); value = arg.substr(arg.find('=') + 1); }

        if (key == "-h" || key == "--help") { print_usage(argv[0]); return 0; }
        if (key == "--gpu-prerender") { opt.gpu_prerender = true; continue; }
        if (key == "--dump-scene") { opt.dump_scene = true; continue; }
        const bool known = key == "-f" || key == "--format" || key == "-W" || key == "--width" || key == "-H" || key == "--height" ||
                           key == "--scene" || key == "--spp" || key == "--depth" || key == "--seed" || key == "--gpus";
        if (!known) {
            std::cerr << "Unknown option '" << argv[i] << "'\n\n" << "Run '" << argv[0] << " -h' to see a list of valid options.\n\n";
            return -1;
        }
        if (value.empty()) {
            if (i == argc - 1 || argv[i + 1][0] == '-') { std::cerr << key << " has no value." << std::endl; return -1; }
            value = argv[++i];
        }
        if (key == "-f" || key == "--format") {
            if (value == "png") opt.png = true;
            else if (value == "ppm") opt.png = false;
            else { std::cerr << "Unknown output format '" << value << "'" << std::endl; return -1; }
        } else if (key == "-W" || key == "--width") { if (!parse_u32(value, "width", "Width", &opt.width)) return -1; }
        else if (key == "-H" || key == "--height") { if (!parse_u32(value, "height", "Height", &opt.height)) return -1; }
        else if (key == "--spp") { if (!parse_u32(value, "spp", "Spp", &opt.spp)) return -1; }
        else if (key == "--depth") { if (!parse_u32(value, "depth", "Depth", &opt.depth)) return -1; }
        else if (key == "--seed") { if (!parse_u32(value, "seed", "Seed", &opt.seed)) return -1; }
        else if (key == "--gpus") { if (!parse_u32(value, "gpus", "Gpus", &opt.gpus)) return -1; }
        else opt.scene = value;
    }
    if (opt.output_path.empty() && !opt.dump_scene) { std::cerr << "No output path given." << std::endl; return -1; }
    return 1;
}

template <class Fn>
void sphere_scene(HipRenderer& r, Fn generate) {
    const uint32_t n = generate(nullptr, nullptr, 0);
    std::vector<float> cr(4 * (size_t)n);
    std::vector<rt3_material> mats(n);
    generate(cr.data(), mats.data(), n);
    r.prerender(Tools::Array<ECS::RenderEntity*>());
    r.set_spheres(cr, mats);
}

}  // namespace

int main(int argc, const char** argv) {
    Options opt;
    const int parsed = parse_cli(opt, argc, argv);
    if (parsed <= 0) return parsed;

    if (opt.dump_scene) {                                            // parse a .scene file and print its entities (no GPU needed)
        try {
            Tools::Array<ECS::RenderEntity*> entities = SceneParser::parse_file(opt.scene);
            for (size_t i = 0; i < entities.size(); i++) {
                const ECS::RenderEntity* e = entities[i];
                std::cout << ECS::entity_type_names[e->type] << " faces=" << e->pre_render_faces << " vertices=" << e->pre_render_vertices;
                if (e->type == ECS::et_triangle) { const auto* t = static_cast<const ECS::Triangle*>(e); for (int k = 0; k < 3; k++) std::cout << " p" << k + 1 << "=(" << t->points[k].x << "," << t->points[k].y << "," << t->points[k].z << ")"; std::cout << " color=(" << t->color.x << "," << t->color.y << "," << t->color.z << ")"; }
                if (e->type == ECS::et_sphere || e->type == ECS::et_analytic_sphere) { const auto* s = static_cast<const ECS::Sphere*>(e); std::cout << " center=(" << s->center.x << "," << s->center.y << "," << s->center.z << ") radius=" << s->radius << " grid=" << s->n_meridians << "x" << s->n_parallels << " color=(" << s->color.x << "," << s->color.y << "," << s->color.z << ")"; }
                if (e->type == ECS::et_object) { const auto* o = static_cast<const ECS::Object*>(e); std::cout << " center=(" << o->center.x << "," << o->center.y << "," << o->center.z << ") scale=" << o->scale << " color=(" << o->color.x << "," << o->color.y << "," << o->color.z << ")"; }
                if (e->has_material) std::cout << " material=" << e->material.kind << " param=" << e->material.param;
                std::cout << "\n";
                delete e;
            }
            return 0;
        } catch (Fatal& e) { std::cerr << "fatal: " << e.what() << std::endl; return -1; }
    }

    try {
        std::vector<int> devices;
        for (uint32_t i = 0; i < (opt.gpus ? opt.gpus : 1); i++) devices.push_back((int)i);
        // RT3_DEVICE_LIST="0,0,0" (testing aid): the device behind each of the --gpus shards, so that the multi-device path —
        // device tiles, device-to-device gather into shard 0's frame — can be exercised on a box with a single GPU
        if (const char* list = std::getenv("RT3_DEVICE_LIST")) {
            devices.clear();
            std::stringstream ss(list);
            std::string item;
            while (std::getline(ss, item, ',')) if (!item.empty()) devices.push_back(std::atoi(item.c_str()));
            if (devices.empty()) devices.push_back(0);
        }
        HipRenderer renderer(devices);
        renderer.set_gpu_prerender(opt.gpu_prerender);
        Camera cam;
        PathOptions path;
        path.spp = opt.spp; path.max_depth = opt.depth ? opt.depth : 1; path.seed = opt.seed;
        const float aspect = (float)opt.width / (float)opt.height;

        if (opt.scene == "builtin") {                               // Main.cpp:272, :280-283
            cam.update(opt.width, opt.height, 2.0f, aspect * 2.0f, 2.0f);
            Tools::Array<ECS::RenderEntity*> entities({
                ECS::create_object("bin/objects/teddy.obj", { 0.0f, 0.0f, -3.0f }, 1.0f / 17.0f, { 1.0f, 0.0f, 0.0f }),
                ECS::create_sphere({ -2.0f, 0.0f, -5.0f }, 1.0f, 8, 8, { 0.0f, 0.0f, 1.0f }) });
            renderer.prerender(entities);
            for (size_t i = 0; i < entities.size(); i++) delete entities[i];
        } else if (opt.scene == "three") {
            cam.update(opt.width, opt.height, 1.0f, aspect * 2.0f, 2.0f);
            sphere_scene(renderer, [](float* c, rt3_material* m, uint32_t cap) { return rt3_scene_three_spheres(c, m, cap); });
            path.flags = RT3_FLAG_GAMMA2;
        } else if (opt.scene == "weekend") {
            cam.look_at(opt.width, opt.height, { 13.0f, 2.0f, 3.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 1.0f, 0.0f }, 20.0f, 10.0f);
            sphere_scene(renderer, [](float* c, rt3_material* m, uint32_t cap) { return rt3_scene_weekend(42, c, m, cap); });
            path.flags = RT3_FLAG_GAMMA2; path.lens_radius = 0.05f;
        } else if (opt.scene == "stress100k") {
            cam.look_at(opt.width, opt.height, { 0.0f, 8.0f, 12.0f }, { 0.0f, 6.0f, -50.0f }, { 0.0f, 1.0f, 0.0f }, 45.0f, 1.0f);
            sphere_scene(renderer, [](float* c, rt3_material* m, uint32_t cap) { return rt3_scene_stress(100000, 43, c, m, cap); });
            path.flags = RT3_FLAG_GAMMA2;
        } else if (opt.scene == "cornell") {
            cam.update(opt.width, opt.height, 2.0f, aspect * 2.0f, 2.0f);
            const uint32_t n = rt3_scene_cornell(64, nullptr, nullptr, nullptr, 0);
            std::vector<rt3_gface> faces(n); std::vector<float> verts(12 * (size_t)n); std::vector<rt3_material> mats(n);
            rt3_scene_cornell(64, faces.data(), verts.data(), mats.data(), n);
            renderer.prerender(Tools::Array<ECS::RenderEntity*>());
            renderer.set_mesh(faces, verts, mats);
            path.flags = RT3_FLAG_GAMMA2 | RT3_FLAG_BLACK_BACKGROUND;
        } else if (opt.scene.size() > 6 && opt.scene.compare(opt.scene.size() - 6, 6, ".scene") == 0) {   // a SceneLang file
            cam.update(opt.width, opt.height, 2.0f, aspect * 2.0f, 2.0f);
            Tools::Array<ECS::RenderEntity*> entities = SceneParser::parse_file(opt.scene);
            renderer.prerender(entities);
            for (size_t i = 0; i < entities.size(); i++) delete entities[i];
            path.flags = opt.spp ? RT3_FLAG_GAMMA2 : 0;
        } else {
            std::cerr << "Unknown scene '" << opt.scene << "'" << std::endl;
            return -1;
        }
        const bool from_file = opt.scene.size() > 6 && opt.scene.compare(opt.scene.size() - 6, 6, ".scene") == 0;
        if (opt.scene != "builtin" && !from_file && path.spp == 0) path.spp = 16;   // the analytic scenes only exist in Mode X
        renderer.configure(path);
        renderer.render(cam);

        const rt3_stats st = renderer.stats();
        std::cerr << "rendered " << opt.width << "x" << opt.height << (path.spp ? " x " + std::to_string(path.spp) + " spp" : " (mode R)")
                  << " in " << st.total_ms << " ms on device 0: " << st.ray_casts << " rays, " << st.prim_tests << " ray-primitive tests\n";
        if (opt.png) cam.get_frame().to_png(opt.output_path);
        else cam.get_frame().to_ppm(opt.output_path);
    } catch (Fatal& e) {
        std::cerr << "fatal: " << e.what() << std::endl;            // the reference logs and returns -1 (Main.cpp:305-308)
        return -1;
    }
    return 0;
}
