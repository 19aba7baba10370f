// sceneparser/SceneParser.cpp — recursive-descent implementation of SceneParser.hpp.
#include "sceneparser/SceneParser.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <unistd.h>

#include "camera/Frame.hpp"          // RayTracer::Fatal

using namespace RayTracer;
using namespace RayTracer::ECS;

namespace {

enum class T { End, Id, Int, Float, String, Punct, Dot };            // Punct: one or two characters, in text

struct Token { T kind = T::End; std::string text; int line = 0; bool space_before = false, tight_after = false; };

struct Value {                                                         // 32-bit constants of SceneLang §3
    enum Kind { Bool, Int, Uint, Float, Vec3 } kind = Float;
    double s = 0.0;                                                    // scalar payload
    double v[3] = { 0, 0, 0 };
    static Value scalar(Kind k, double x) { Value r; r.kind = k; r.s = x; return r; }
    static Value vec(double x, double y, double z) { Value r; r.kind = Vec3; r.v[0] = x; r.v[1] = y; r.v[2] = z; return r; }
    double num() const { return s; }
};

struct DataBlock { std::string path; };                                // every .obj ends up as a file for create_object

class Parser {
public:
    Parser(const std::string& text, const std::string& name, const std::string& dir, std::map<std::string, DataBlock>& data,
           std::map<std::string, std::map<std::string, Value>>& fields, std::vector<RenderEntity*>& out, int depth = 0)
        : src(text), file(name), base_dir(dir), include_depth(depth), data_blocks(data), known(fields), entities(out) { advance(); }

    void parse_toplevel() {
        while (tok.kind != T::End) {
            skip_error_statements();
            if (tok.kind == T::End) break;
            const std::string section = expect_id("section name");
            expect_punct("{");
            if (section == "data") while (!is_punct("}")) data_statement();
            else if (section == "entities") while (!is_punct("}")) entity_statement();
            else if (section == "global") while (!is_punct("}")) parameter("global", known["global"], nullptr);
            else fail("unknown section '" + section + "' (expected data, entities or global)");
            expect_punct("}");
        }
    }

private:
    std::string src, file, base_dir;
    size_t pos = 0;
    int line = 1;
    int include_depth = 0;                                            // files above this one in the #include chain
    int nesting = 0;                                                  // recursion depth of the expression parser
    static constexpr int kMaxIncludeDepth = 32, kMaxNesting = 200;    // found by the sanitizer build: a file that includes itself, 5000 '(' in a row
    struct Nest {                                                     // every recursive descent step holds one
        Parser& p;
        explicit Nest(Parser& q) : p(q) { if (++p.nesting > kMaxNesting) p.fail("expression nested deeper than " + std::to_string(kMaxNesting) + " levels"); }
        ~Nest() { --p.nesting; }
    };
    Token tok;
    std::map<std::string, DataBlock>& data_blocks;
    std::map<std::string, std::map<std::string, Value>>& known;       // entity id (or "global") -> field -> value
    std::vector<RenderEntity*>& entities;

    [[noreturn]] void fail(const std::string& msg) const { throw Fatal(file + ":" + std::to_string(tok.line ? tok.line : line) + ": " + msg); }

    // ------------------------------------------------------------------ lexer
    void skip_blank() {
        for (;;) {
            while (pos < src.size() && (src[pos] == ' ' || src[pos] == '\t' || src[pos] == '\r' || src[pos] == '\n')) { if (src[pos] == '\n') line++; pos++; }
            if (src.compare(pos, 2, "/*") == 0) {
                const size_t end = src.find("*/", pos + 2);
                if (end == std::string::npos) fail("unterminated comment");
                for (size_t i = pos; i < end; i++) if (src[i] == '\n') line++;
                pos = end + 2;
            } else if (src.compare(pos, 2, "//") == 0) {
                while (pos < src.size() && src[pos] != '\n') pos++;
            } else if (src[pos] == '#' && src.compare(pos, 8, "#include") == 0) {
                pos += 8;
                skip_blank();
                if (pos >= src.size() || (src[pos] != '"' && src[pos] != '\'')) fail("#include needs a string");
                const char q = src[pos];
                const size_t end = src.find(q, pos + 1);
                if (end == std::string::npos) fail("unterminated string");
                std::string inc = src.substr(pos + 1, end - pos - 1);
                pos = end + 1;
                if (inc.empty() || inc[0] != '/') inc = base_dir + inc;             // relative to the current source file (§4)
                std::ifstream in(inc);
                if (!in.is_open()) fail("cannot open included file '" + inc + "'");
                std::stringstream ss; ss << in.rdbuf();
                const size_t slash = inc.find_last_of('/');
                if (include_depth >= kMaxIncludeDepth) fail("#include nested deeper than " + std::to_string(kMaxIncludeDepth) + " files (does '" + inc + "' include itself?)");
                Parser sub(ss.str(), inc, slash == std::string::npos ? "" : inc.substr(0, slash + 1), data_blocks, known, entities, include_depth + 1);
                sub.parse_toplevel();                                             // conceptually pasted in place
            } else return;
        }
    }

    void advance() {
        const size_t before = pos;
        skip_blank();
        tok = Token();
        tok.line = line;
        tok.space_before = pos > before || pos == 0;
        if (pos >= src.size()) return;
        const char c = src[pos];
        auto is_id0 = [](char ch) { return (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_'; };
        auto is_digit = [](char ch) { return ch >= '0' && ch <= '9'; };
        if (is_id0(c) || c == '@') {
            size_t e = pos + 1;
            while (e < src.size() && (is_id0(src[e]) || is_digit(src[e]) || (src[e] == '-' && e + 1 < src.size() && (is_id0(src[e + 1]) || is_digit(src[e + 1]))))) e++;
            tok.kind = T::Id; tok.text = src.substr(pos, e - pos); pos = e;
        } else if (is_digit(c) || (c == '.' && pos + 1 < src.size() && is_digit(src[pos + 1]))) {
            char* end = nullptr;
            std::strtod(src.c_str() + pos, &end);
            const size_t e = (size_t)(end - src.c_str());
            tok.text = src.substr(pos, e - pos);
            tok.kind = tok.text.find_first_of(".eE") == std::string::npos ? T::Int : T::Float;
            pos = e;
        } else if (c == '"' || c == '\'') {
            std::string s; size_t e = pos + 1;
            while (e < src.size() && src[e] != c) {
                if (src[e] == '\\' && e + 1 < src.size()) { const char n = src[++e]; s += n == 'n' ? '\n' : n == 't' ? '\t' : n == 'r' ? '\r' : n; }
                else { if (src[e] == '\n') line++; s += src[e]; }
                e++;
            }
            if (e >= src.size()) fail("unterminated string");
            tok.kind = T::String; tok.text = s; pos = e + 1;
        } else if (c == '.') { tok.kind = T::Dot; tok.text = "."; pos++; }
        else {
            static const char* two[] = { ">>", "<<", "==", "!=", ">=", "<=", "&&", "||" };
            tok.kind = T::Punct; tok.text = std::string(1, c);
            for (const char* t : two) if (src.compare(pos, 2, t) == 0) tok.text = t;
            pos += tok.text.size();
            tok.tight_after = pos < src.size() && (is_digit(src[pos]) || is_id0(src[pos]) || src[pos] == '.' || src[pos] == '(');
        }
    }

    bool is_punct(const char* p) const { return tok.kind == T::Punct && tok.text == p; }
    void expect_punct(const char* p) { if (!is_punct(p)) fail(std::string("expected '") + p + "', got '" + tok.text + "'"); advance(); }
    std::string expect_id(const char* what) { if (tok.kind != T::Id || tok.text[0] == '@') fail(std::string("expected ") + what + ", got '" + tok.text + "'"); std::string s = tok.text; advance(); return s; }

    void skip_error_statements() {                                     // §2: @warning/@error/@ignore <id>|<string>; the sample also uses @suppress
        while (tok.kind == T::Id && tok.text[0] == '@') {
            const std::string which = tok.text;
            advance();
            if (tok.kind != T::Id && tok.kind != T::String) fail(which + " needs an identifier or a string");
            if (which == "@error") fail("@error: " + tok.text);
            if (which == "@warning") std::fprintf(stderr, "%s:%d: warning: %s\n", file.c_str(), tok.line, tok.text.c_str());
            advance();
        }
    }

    // ------------------------------------------------------------------ data section
    void data_statement() {
        skip_error_statements();
        bool external = false;
        if (tok.kind == T::Id && tok.text == "extern") { external = true; advance(); }
        if (tok.kind != T::Dot) fail("expected a data format ('.obj')");
        advance();
        if (expect_id("data format") != "obj") fail("only the .obj format is supported");
        const std::string id = expect_id("data identifier");
        if (data_blocks.count(id)) fail("duplicate data identifier '" + id + "'");
        if (external) {
            expect_punct(":");
            if (tok.kind != T::String) fail("extern data needs a path string");
            data_blocks[id].path = tok.text;
            advance();
            expect_punct(";");
        } else {                                                       // inline block: raw text up to the matching brace
            if (!is_punct("{")) fail("expected '{' after the data identifier");
            size_t depth = 1, e = pos;
            while (e < src.size() && depth) { if (src[e] == '{') depth++; else if (src[e] == '}') depth--; if (src[e] == '\n') line++; e++; }
            if (depth) fail("unterminated data block");
            std::string body = src.substr(pos, e - 1 - pos), cleaned;
            std::stringstream ss(body); std::string ln;                // object files may not contain blank lines (Object.cpp:157)
            while (std::getline(ss, ln)) if (ln.find_first_not_of(" \t\r") != std::string::npos) cleaned += ln + "\n";
            char tmpl[] = "/tmp/rt3_scene_XXXXXX";
            const int fd = mkstemp(tmpl);
            if (fd < 0 || write(fd, cleaned.data(), cleaned.size()) != (ssize_t)cleaned.size()) fail("cannot write a temporary object file");
            close(fd);
            data_blocks[id].path = tmpl;
            pos = e;
            advance();
        }
    }

    // ------------------------------------------------------------------ expressions (C precedence: unary, * / %, + -)
    Value primary() {
        const Nest guard(*this);
        if (is_punct("(")) {
            advance();
            if (tok.kind == T::Id && (tok.text == "bool" || tok.text == "int" || tok.text == "uint" || tok.text == "float" || tok.text == "vec3")) {
                const std::string ty = tok.text;
                advance();
                expect_punct(")");
                return cast(unary(), ty);
            }
            const Value v = expr();
            expect_punct(")");
            return v;
        }
        if (tok.kind == T::Int || tok.kind == T::Float) {
            const Value v = Value::scalar(tok.kind == T::Int ? Value::Int : Value::Float, std::strtod(tok.text.c_str(), nullptr));
            advance();
            return v;
        }
        if (tok.kind == T::Id) {
            const std::string id = tok.text;
            advance();
            if (id == "true" || id == "false") return Value::scalar(Value::Bool, id == "true");
            if (tok.kind == T::Dot) {                                  // <ref> := <id>.<id>
                advance();
                const std::string field = expect_id("field name");
                const auto e = known.find(id);
                if (e == known.end() || !e->second.count(field)) fail("unknown reference '" + id + "." + field + "'");
                return e->second.at(field);
            }
            if (is_punct("(")) {                                       // build-in functions (Appendix C is empty upstream)
                advance();
                std::vector<Value> args;
                while (!is_punct(")")) { args.push_back(expr()); if (is_punct(",")) advance(); }
                advance();
                if (id == "sqrt" && args.size() == 1) return Value::scalar(Value::Float, std::sqrt(args[0].num()));
                if (id == "sin" && args.size() == 1) return Value::scalar(Value::Float, std::sin(args[0].num()));
                if (id == "cos" && args.size() == 1) return Value::scalar(Value::Float, std::cos(args[0].num()));
                if (id == "vec3" && args.size() == 3) return Value::vec(args[0].num(), args[1].num(), args[2].num());
                fail("unknown function '" + id + "'");
            }
            fail("unexpected identifier '" + id + "' in an expression");
        }
        fail("expected an expression, got '" + tok.text + "'");
    }
    Value unary() {
        const Nest guard(*this);
        if (is_punct("-")) { advance(); Value v = unary(); if (v.kind == Value::Vec3) return Value::vec(-v.v[0], -v.v[1], -v.v[2]); v.s = -v.s; if (v.kind == Value::Uint) v.kind = Value::Int; return v; }
        if (is_punct("+")) { advance(); return unary(); }
        if (is_punct("!")) { advance(); return Value::scalar(Value::Bool, unary().num() == 0.0); }
        return primary();
    }
    static Value arith(const Value& a, const Value& b, char op) {
        auto f = [op](double x, double y) { return op == '+' ? x + y : op == '-' ? x - y : op == '*' ? x * y : op == '/' ? x / y : std::fmod(x, y); };
        if (a.kind == Value::Vec3 || b.kind == Value::Vec3) {
            Value r; r.kind = Value::Vec3;
            for (int i = 0; i < 3; i++) r.v[i] = f(a.kind == Value::Vec3 ? a.v[i] : a.s, b.kind == Value::Vec3 ? b.v[i] : b.s);
            return r;
        }
        const bool fl = a.kind == Value::Float || b.kind == Value::Float;
        double r = f(a.s, b.s);
        if (!fl) r = std::trunc(r);                                    // integer arithmetic truncates like C
        return Value::scalar(fl ? Value::Float : Value::Int, r);
    }
    Value term() { Value v = unary(); while (is_punct("*") || is_punct("/") || is_punct("%")) { const char op = tok.text[0]; advance(); v = arith(v, unary(), op); } return v; }
    // `a - b` and `a-b` subtract; `a -b` starts the next component of a vec3 written as three numbers (the grammar's <float> has
    // no sign, tests/test.scene writes `p1: -1.0 0.0 0.0;`, so signed components need this whitespace rule)
    bool is_new_component() const { return (is_punct("-") || is_punct("+")) && tok.space_before && tok.tight_after; }
    Value expr() { Value v = term(); while ((is_punct("+") || is_punct("-")) && !is_new_component()) { const char op = tok.text[0]; advance(); v = arith(v, term(), op); } return v; }

    Value cast(Value v, const std::string& ty) const {
        if (ty == "vec3") return v.kind == Value::Vec3 ? v : Value::vec(v.s, v.s, v.s);
        const double x = v.kind == Value::Vec3 ? v.v[0] : v.s;
        if (ty == "float") return Value::scalar(Value::Float, x);
        if (ty == "bool") return Value::scalar(Value::Bool, x != 0.0);
        return Value::scalar(ty == "uint" ? Value::Uint : Value::Int, std::trunc(x));
    }
    // a parameter value: an expression, or three expressions in a row for a vec3 (`color: 1.0 0.0 0.0;`, tests/test.scene)
    Value value() {
        Value a = expr();
        if (is_punct(";")) return a;
        const Value b = expr(), c = expr();
        return Value::vec(a.num(), b.num(), c.num());
    }

    // ------------------------------------------------------------------ parameters and entities
    // parses `[<data_type>] <id>: <value>;` or `data <id>: .obj <id>;` (the sample writes `data: .obj id;`)
    void parameter(const std::string& owner, std::map<std::string, Value>& fields, std::string* data_ref) {
        skip_error_statements();
        std::string ty, key = expect_id("parameter name");
        if (key == "data") {
            if (tok.kind == T::Id) advance();                         // `data <id>: <format> <id>;` — the parameter's own id is unused
            expect_punct(":");
            if (tok.kind != T::Dot) fail("expected a data format ('.obj')");
            advance();
            if (expect_id("data format") != "obj") fail("only the .obj format is supported");
            const std::string ref = expect_id("data identifier");
            if (!data_blocks.count(ref)) fail("unknown data '" + ref + "'");
            if (!data_ref) fail("'" + owner + "' cannot take a data parameter");
            *data_ref = ref;
            expect_punct(";");
            return;
        }
        if (key == "bool" || key == "int" || key == "uint" || key == "float" || key == "vec3") { ty = key; key = expect_id("parameter name"); }
        expect_punct(":");
        Value v;
        if (tok.kind == T::Id && tok.kind != T::Dot && !ty.size() && (tok.text == "lambertian" || tok.text == "metal" || tok.text == "dielectric" || tok.text == "emissive")) {
            v = Value::scalar(Value::Uint, tok.text == "lambertian" ? RT3_MAT_LAMBERT : tok.text == "metal" ? RT3_MAT_METAL : tok.text == "dielectric" ? RT3_MAT_DIELECTRIC : RT3_MAT_FLAT);
            advance();
        } else v = value();
        if (!ty.empty()) v = cast(v, ty);
        expect_punct(";");
        fields[key] = v;
    }

    static glm::vec3 vec_of(const std::map<std::string, Value>& f, const std::string& key, glm::vec3 fallback) {
        const auto it = f.find(key);
        if (it == f.end()) return fallback;
        const Value& v = it->second;
        return v.kind == Value::Vec3 ? glm::vec3((float)v.v[0], (float)v.v[1], (float)v.v[2]) : glm::vec3((float)v.s);
    }
    static double num_of(const std::map<std::string, Value>& f, const std::string& key, double fallback) {
        const auto it = f.find(key);
        return it == f.end() ? fallback : (it->second.kind == Value::Vec3 ? it->second.v[0] : it->second.s);
    }

    void entity_statement() {
        skip_error_statements();
        const std::string type = expect_id("entity type");
        if (type != "triangle" && type != "sphere" && type != "object") fail("unknown entity type '" + type + "'");
        const std::string id = expect_id("entity identifier");
        if (id == "global") fail("an entity cannot be called 'global'");
        if (known.count(id)) fail("duplicate entity identifier '" + id + "'");
        std::map<std::string, Value>& f = known[id];
        std::string data_ref;
        expect_punct("{");
        while (!is_punct("}")) parameter(id, f, type == "object" ? &data_ref : nullptr);
        expect_punct("}");

        const glm::vec3 color = vec_of(f, "color", glm::vec3(1.0f));
        RenderEntity* e = nullptr;
        if (type == "triangle") {
            for (const char* k : { "p1", "p2", "p3" }) if (!f.count(k)) fail("triangle '" + id + "' lacks '" + k + "'");
            e = create_triangle(vec_of(f, "p1", {}), vec_of(f, "p2", {}), vec_of(f, "p3", {}), color);
        } else if (type == "sphere") {
            if (!f.count("center") || !f.count("radius")) fail("sphere '" + id + "' needs 'center' and 'radius'");
            e = create_sphere(vec_of(f, "center", {}), (float)num_of(f, "radius", 1.0), (uint32_t)num_of(f, "n_meridians", 0), (uint32_t)num_of(f, "n_parallels", 0), color);
        } else {
            if (data_ref.empty()) fail("object '" + id + "' needs a 'data' parameter");
            std::string path = data_blocks.at(data_ref).path;
            e = create_object(path, vec_of(f, "center", glm::vec3(0.0f)), (float)num_of(f, "scale", 1.0), color);
        }
        if (f.count("material")) {
            const uint32_t kind = (uint32_t)num_of(f, "material", 0);
            rt3_material m = kind == RT3_MAT_LAMBERT ? lambertian(color) : kind == RT3_MAT_METAL ? metal(color, (float)num_of(f, "fuzz", 0.0))
                           : kind == RT3_MAT_DIELECTRIC ? dielectric((float)num_of(f, "ior", 1.5)) : emissive(color);
            e->has_material = true;
            e->material = m;
        }
        entities.push_back(e);
    }
};

Tools::Array<RenderEntity*> run(const std::string& text, const std::string& name, const std::string& dir) {
    std::map<std::string, DataBlock> data;
    std::map<std::string, std::map<std::string, Value>> fields;
    std::vector<RenderEntity*> out;
    try {
        Parser p(text, name, dir, data, fields, out);
        p.parse_toplevel();
    } catch (...) {
        for (RenderEntity* e : out) delete e;
        for (auto& d : data) if (d.second.path.compare(0, 15, "/tmp/rt3_scene_") == 0) std::remove(d.second.path.c_str());
        throw;
    }
    // inline data was materialised as temporary files; create_object has already counted them, pre-rendering reads them again,
    // so they stay until the process ends (they are tiny) — extern paths are the user's own files
    return Tools::Array<RenderEntity*>(out);
}

}  // namespace

Tools::Array<RenderEntity*> SceneParser::parse_file(const std::string& path) {
    std::ifstream in(path);
    if (!in.is_open()) throw Fatal("Could not open scene file '" + path + "'");
    std::stringstream ss;
    ss << in.rdbuf();
    const size_t slash = path.find_last_of('/');
    return run(ss.str(), path, slash == std::string::npos ? "" : path.substr(0, slash + 1));
}

Tools::Array<RenderEntity*> SceneParser::parse_string(const std::string& text, const std::string& name) { return run(text, name, ""); }
