// sceneparser/SceneParser.hpp — a parser for the reference's SceneLang (spec: src/lib/sceneparser/SceneLang.md; the
// reference's own SceneParser class is an empty stub that is not even part of its build).
//
// Implemented: the `data`, `entities` and `global` sections (§2), inline `.obj` blocks and `extern .obj` files (§3 data),
// triangle / sphere / object entities with typed or untyped parameters (the grammar of §3 requires a type keyword, the
// reference's sample tests/test.scene omits it: both are accepted), C-style constant expressions over bool / int / uint /
// float / vec3 with + - * / % and unary -, parentheses, casts and `<entity>.<field>` / `global.<field>` references,
// `@warning/@error/@ignore/@suppress` statements (errors abort, the rest are skipped), `#include "file"` (§4), comments.
// Reserved entity keywords (Appendix B is empty upstream; taken from tests/test.scene and the create_* signatures):
//   triangle: p1 p2 p3 color            sphere: center radius n_meridians n_parallels color            object: center scale data color
// Extension for Mode X (optional on every entity): material (lambertian | metal | dielectric | emissive), fuzz, ior;
// a sphere with n_meridians = n_parallels = 0 (or neither given) is analytic.
#ifndef RT3_HOST_SCENE_PARSER_HPP
#define RT3_HOST_SCENE_PARSER_HPP
#include <string>
#include <vector>
#include "entities/RenderEntity.hpp"
#include "tools/Array.hpp"

namespace RayTracer {
class SceneParser {
public:
    // Parses `path` (and its includes) and returns the entities in file order; the caller owns them (as with ECS::create_*).
    // Throws Fatal with "<file>:<line>: message" on the first error.
    static Tools::Array<ECS::RenderEntity*> parse_file(const std::string& path);
    static Tools::Array<ECS::RenderEntity*> parse_string(const std::string& text, const std::string& name = "<string>");
};
}  // namespace RayTracer
#endif
