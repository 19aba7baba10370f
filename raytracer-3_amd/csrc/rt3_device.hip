// rt3_device.hip — the render path on gfx950 (CDNA4): HIP kernels + the device half of the C ABI (include/rt3.h).
//
//   rt3_kernel_common.hpp   constants, arithmetic helpers, Mode-X launch arguments, start_path()
//   rt3_path.hpp            refill (ballot + prefix count, ray stock) and shade_lane()
//   rt3_valu_scan.hpp       k_mode_r      Mode R: SequentialRenderer::render + ray_color (src/lib/renderer/SequentialRenderer.cpp:47-109,
//                                         269-308; GLSL twin src/lib/shaders/raytracer_v3.glsl:91-143,187-203), one thread per pixel,
//                                         IEEE arithmetic in the reference's order
//                           k_mode_r_fast the same through a conservative bounding-sphere scan on the vector ALU
//                           k_trace       Mode X: the recursion of the (unfinished) raytracer_v4.glsl:187-290 flattened into an
//                                         iterative per-wavefront loop; every lane carries one path, lanes whose path ended are
//                                         refilled by ballot + prefix count, ray state never leaves the registers
//   rt3_matrix_filter.hpp   k_trace_mfma / k_trace_mfma_tiled / k_mode_r_mfma: the same loops with the candidate search on the
//                           matrix cores (the default; RT3_NO_MFMA=1 selects the VALU scans)
//   rt3_reduce.hpp          per-sample radiance (SampleStorage of raytracer_v4.glsl:107-111) summed in sample order and resolved
//                           (the reduce pass reduce_v1.glsl never got) — the image is bitwise independent of scheduling and GPU count
//   rt3_scene_kernels.hpp   HIP equivalents of the pre-render shaders and of the merge
//   below                   the device context and the extern "C" entry points
//
// Compiled with -ffp-contract=off: a*b+c is two roundings unless written __builtin_fmaf.  Division and sqrt are
// the correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), f32 denormals are kept.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "rt3.h"

#include "rt3_kernel_common.hpp"
#include "rt3_path.hpp"
#include "rt3_valu_scan.hpp"
#include "rt3_matrix_filter.hpp"
#include "rt3_level_filter.hpp"
#include "rt3_reduce.hpp"
#include "rt3_scene_kernels.hpp"

// ======================================================================================================
// Host side of the device context
// ======================================================================================================
struct rt3_ctx {
    int device = 0;
    int num_cu = 0;
    hipStream_t stream = nullptr;                                   // used by the synchronous entry points
    std::string err;

    // mesh
    uint32_t n_faces = 0;
    float4* d_tri = nullptr; float4* d_tri_mat = nullptr; uint32_t* d_tri_kind = nullptr; float4* d_tri_bound = nullptr; u32x4* d_tri_frag = nullptr;
    // merged entity buffers on the device (GFace[] / vec4[] as the reference keeps them), filled by rt3_mesh_*
    rt3_gface* d_gfaces = nullptr; float4* d_verts = nullptr; uint32_t cap_gfaces = 0, cap_verts = 0;
    rt3_material* d_face_mats_in = nullptr; uint32_t* d_error = nullptr;
    // spheres
    uint32_t n_sph = 0;
    float4* d_sph = nullptr; uint32_t* d_sph_frag = nullptr; uint32_t* d_sph_frag32 = nullptr; float sph_centre[3] = { 0.0f, 0.0f, 0.0f }; uint32_t n_direct = 0; uint32_t direct[4] = { 0, 0, 0, 0 }; float tri_centre[3] = { 0.0f, 0.0f, 0.0f }; uint32_t* d_box = nullptr; uint32_t* d_tri_frag_r = nullptr; float* d_sph_invr = nullptr; float4* d_sph_mat = nullptr; uint32_t* d_sph_kind = nullptr;

    // group rows of the two-level filter: faces in face order, spheres in the order of a spatial median split
    u32x4* d_tri_gfrag = nullptr; float4* d_tri_grp = nullptr; uint32_t* d_tri_perm = nullptr; uint32_t n_tri_groups = 0;
    u32x4* d_sph_gfrag = nullptr; float4* d_sph_grp = nullptr; uint32_t* d_sph_perm = nullptr; uint32_t n_sph_groups = 0;
    float4* d_tri_leaf = nullptr; float4* d_sph_leaf = nullptr; uint32_t n_tri_leaves = 0, n_sph_leaves = 0;      // three-level filter: the leaf groups' bounds
    float4* d_tri_rowb = nullptr; float4* d_sph_rowb = nullptr;    // ... and the rows' own bounds in f32 (rows behind a ray are dropped before their leaves are tested)
    float4* d_tri_rec = nullptr;                                   // the faces' records in group order (the exact test of the multi-level filter reads these)
    u32x4* d_tri_sfrag = nullptr; u32x4* d_sph_sfrag = nullptr; float4* d_tri_srowb = nullptr; float4* d_sph_srowb = nullptr;   // four levels: super-rows of kSuper rows
    uint32_t n_tri_super = 0, n_sph_super = 0;
    uint32_t* d_strips = nullptr; size_t strip_entries = 0;          // deferred member tests: kStripPairs pairs per wave of the grid

    // work buffers
    Rgb* d_rad = nullptr; size_t rad_entries = 0;
    float4* d_accum = nullptr; size_t accum_entries = 0;
    float4* d_accum_sq = nullptr; size_t accum_sq_entries = 0;      // RT3_FLAG_VARIANCE: per-pixel sums of squares
    uint32_t* d_out = nullptr; size_t out_entries = 0;
    uint32_t* d_work = nullptr;                                     // [0] work counter
    unsigned long long* d_casts = nullptr;
    uint64_t rad_cap_bytes = 16ull << 30;
    bool force_plain_mode_r = false;                                // tests: compare the two Mode-R kernels
    bool force_brute = false;                                       // tests / fuzzers: unfiltered Mode-X kernel
    bool force_flat = false;                                        // tests / A-B: one filter row per primitive (no groups)
    uint64_t last_filter_rows = 0;                                  // rows the matrix filter scanned per ray cast in the last Mode-X render
    // the accumulation a progressive render continues (rt3_render_path_range): what it belongs to and how far it got
    bool acc_valid = false; rt3_params acc_params{}; rt3_camera acc_cam{}; uint32_t acc_done = 0; uint32_t acc_npix = 0;
    // launch configuration per (kernel, dynamic LDS): max dynamic LDS attribute set, workgroups per CU
    std::map<std::pair<const void*, size_t>, int> occupancy;
    std::set<int> peers_enabled;                                    // devices this context's device may already write to

    // stats of the last render
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;              // per dominant-kernel launch
    uint32_t ev_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_acc = nullptr; bool ev_acc_recorded = false;      // end of the last Mode-X render: what the next user of d_rad / d_accum waits for
    hipStream_t last_stream = nullptr;
    uint64_t last_samples = 0;
    bool last_was_path = false;
    bool last_mfma16 = false;                                       // the last trace kernel was a tiled one (16x16x32 MFMAs)
    bool rendered = false;
    // what the render in flight will report once its last launch has been issued (committed only then)
};

namespace {

thread_local std::string g_create_error;

#define RT3_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                               \
            return RT3_E_DEVICE;                                                                        \
        }                                                                                               \
    } while (0)

int fail(rt3_ctx* ctx, int code, const std::string& msg) { ctx->err = msg; return code; }

template <typename T>
int upload(rt3_ctx* ctx, T** dst, const std::vector<T>& src) {
    if (*dst) { RT3_HIP(hipFree(*dst)); *dst = nullptr; }
    if (src.empty()) return 0;
    RT3_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
    RT3_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
template <typename T>
int ensure(rt3_ctx* ctx, T** buf, size_t* have, size_t want) {
    if (*have >= want && *buf) return 0;
    if (*buf) { RT3_HIP(hipFree(*buf)); *buf = nullptr; *have = 0; }
    RT3_HIP(hipMalloc((void**)buf, want * sizeof(T)));
    *have = want;
    return 0;
}

CamDev cam_dev(const rt3_camera* c) {
    return CamDev{ c->origin[0], c->origin[1], c->origin[2], c->horizontal[0], c->horizontal[1], c->horizontal[2],
                   c->vertical[0], c->vertical[1], c->vertical[2],
                   c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2] };
}

int take_event_pair(rt3_ctx* ctx, hipEvent_t* a, hipEvent_t* b) {
    if (ctx->ev_used == ctx->ev.size()) {
        hipEvent_t x, y;
        RT3_HIP(hipEventCreate(&x));
        RT3_HIP(hipEventCreate(&y));
        ctx->ev.emplace_back(x, y);
    }
    *a = ctx->ev[ctx->ev_used].first;
    *b = ctx->ev[ctx->ev_used].second;
    ctx->ev_used++;
    return 0;
}

FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{ 0u, 0xFFFFFFFFu };
    if (d <= 1) return f;
    const uint32_t p = 31u - (uint32_t)__builtin_clz(d);
    if ((d & (d - 1)) == 0) { f.magic = 0; f.shift = p - 1; return f; }
    const uint64_t num = 1ull << (32 + p);
    uint64_t m = num / d;
    const uint64_t rem = num % d;
    m += m;
    const uint64_t twice = rem + rem;
    if (twice >= d) m += 1;
    f.magic = (uint32_t)(m + 1);
    f.shift = p;
    return f;
}
uint32_t fastdiv_host(uint32_t n, FastDiv f) {
    if (f.shift == 0xFFFFFFFFu) return n;
    const uint32_t q = (uint32_t)(((uint64_t)f.magic * n) >> 32);
    return (((n - q) >> 1) + q) >> f.shift;
}
// exhaustive near multiples + a pseudo-random sweep; a wrong magic would silently shift pixels
bool fastdiv_ok(uint32_t d, uint32_t n_max) {
    if (d == 0) return false;
    const FastDiv f = make_fastdiv(d);
    uint32_t x = 0x9E3779B9u;
    for (uint32_t i = 0; i < 4096; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint32_t n = x % (n_max + 1u);
        if (fastdiv_host(n, f) != n / d) return false;
    }
    for (uint64_t k = 0; k <= 64 && k * d <= n_max; k++) {
        const uint64_t m = (uint64_t)(n_max / d - k) * d;
        for (int o = -1; o <= 1; o++) {
            const int64_t n = (int64_t)m + o;
            if (n >= 0 && n <= (int64_t)n_max && fastdiv_host((uint32_t)n, f) != (uint32_t)n / d) return false;
        }
    }
    return true;
}

// Device form of a material: (rgb, param), except for a dielectric, whose attenuation is 1 and whose per-hit constants are
// precomputed here with the float operations of DESIGN.md §4.5: (1/ior, r0 for ri = 1/ior, r0 for ri = ior, ior), r0 = ((1-ri)/(1+ri))^2.
float4 pack_material(const rt3_material& m) {
    if (m.kind != RT3_MAT_DIELECTRIC) return make_float4(m.rgb[0], m.rgb[1], m.rgb[2], m.param);
    const float ri_f = 1.0f / m.param, ri_b = m.param;
    float r0f = (1.0f - ri_f) / (1.0f + ri_f), r0b = (1.0f - ri_b) / (1.0f + ri_b);
    r0f = r0f * r0f; r0b = r0b * r0b;
    return make_float4(ri_f, r0f, r0b, m.param);
}

// Sphere-side operand fragments of the matrix filter: [row block of 32 spheres][4 MFMA operands][64 lanes] x 8 bf16.
// Lane l holds, for operand row (l & 31) — sphere b of the block sits in row frag_row_of(b) — K elements 8 (l >> 5) .. +7;
// padding rows can never be candidates.  Coordinates relative to `centre` (sphere_filter_centre: it keeps |C|^2 + |o|^2, and with it the
// filter's margin, small for a scene that is not built around the world origin).
// Rows of `direct` spheres (sphere_direct_list below) can never be candidates: the kernels test those spheres for every ray anyway.
bool is_direct(uint32_t j, const uint32_t* direct, uint32_t n_direct) {
    for (uint32_t i = 0; i < n_direct; i++) if (direct[i] == j) return true;
    return false;
}
std::vector<uint32_t> build_sphere_frags(const float* center_radius, uint32_t n, const float centre[3], const uint32_t* direct, uint32_t n_direct) {
    const uint32_t blocks = (n + 31u) / 32u;
    std::vector<uint32_t> out((size_t)blocks * 4 * 64 * 4, 0u);
    for (uint32_t j = 0; j < blocks * 32; j++) {
        uint32_t fr[4][2][4];
        if (j < n && !is_direct(j, direct, n_direct)) {
            const float* s = center_radius + 4 * (size_t)j;
            const float cx = (float)((double)s[0] - centre[0]), cy = (float)((double)s[1] - centre[1]), cz = (float)((double)s[2] - centre[2]);
            const double c2 = (double)cx * cx + (double)cy * cy + (double)cz * cz, r2 = (double)s[3] * s[3];
            bound_frag_row(cx, cy, cz, filter_kj(c2, r2), fr);
        } else bound_frag_row(0.0f, 0.0f, 0.0f, kNeverCandidate, fr);
        for (int q = 0; q < 4; q++)
            for (int hh = 0; hh < 2; hh++)
                std::memcpy(&out[((((size_t)(j / 32) * 4 + q) * 64) + hh * 32 + frag_row_of(j % 32)) * 4], fr[q][hh], 16);
    }
    return out;
}

// The same spheres as fragments of the K = 32 form of the tiled kernels (rt3_matrix_filter.hpp): [row block][h][lane 16 g + c] x 8 bf16,
// coordinates relative to the same centre.
std::vector<uint32_t> build_sphere_frags32(const float* center_radius, uint32_t n, const float centre[3], const uint32_t* direct, uint32_t n_direct) {
    const uint32_t blocks = (n + 31u) / 32u;
    std::vector<uint32_t> out((size_t)blocks * 2 * 64 * 4, 0u);
    for (uint32_t j = 0; j < blocks * 32; j++) {
        uint32_t fr[4][4];
        if (j < n && !is_direct(j, direct, n_direct)) {
            const float* s = center_radius + 4 * (size_t)j;
            const float cx = (float)((double)s[0] - centre[0]), cy = (float)((double)s[1] - centre[1]), cz = (float)((double)s[2] - centre[2]);
            const double c2 = (double)cx * cx + (double)cy * cy + (double)cz * cz, r2 = (double)s[3] * s[3];
            bound_frag32_row(cx, cy, cz, filter_kj32(c2, r2), fr);
        } else bound_frag32_row(0.0f, 0.0f, 0.0f, kNeverCandidate, fr);
        for (uint32_t g = 0; g < 4; g++) std::memcpy(&out[frag32_index(j / 32, j % 32, g) * 4], fr[g], 16);
    }
    return out;
}

// The centre of the spheres' filter coordinates: the component-wise median of the centres — the middle of where the spheres are,
// whatever a few far or huge ones do (the book scene's ground sphere, centre y = -1000, moves the mean by two units and the median not at
// all; weights of 1 / r^2, the minimiser of the sum of margin / r^2, follow the few smallest spheres instead: 9 % more exact tests on the
// 100 000-sphere scene).  Non-finite coordinates are skipped; (0, 0, 0) without any.
void sphere_filter_centre(const float* center_radius, uint32_t n, float out[3]) {
    std::vector<float> v;
    v.reserve(n);
    for (int a = 0; a < 3; a++) {
        v.clear();
        for (uint32_t i = 0; i < n; i++) { const float c = center_radius[4 * (size_t)i + a]; if (std::isfinite(c)) v.push_back(c); }
        out[a] = 0.0f;
        if (v.empty()) continue;
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        out[a] = v[v.size() / 2];
    }
}

// Spheres that (nearly) every ray is a candidate for: the line of a ray that starts somewhere in the scene meets a sphere whose radius is
// comparable to its distance from there — the book scene's ground (r = 1000, its centre 1000 away).  The filter cannot reject such a sphere
// and it costs the pair list one entry per ray, so the matrix-filter kernels test it directly instead (TraceArgs::direct).  Any choice is
// correct; this one takes the (at most four) spheres with the largest r / max(|centre - c0|, R) above 1/2, R = the median distance of the
// centres from c0, i.e. the scene's own size: a unit sphere in the middle of the book scene (candidate for a few per cent of the rays) stays
// in the filter — a direct test costs every ray ~30 instructions.
uint32_t sphere_direct_list(const float* center_radius, uint32_t n, const float c0[3], uint32_t out[4]) {
    std::vector<double> dist(n);
    for (uint32_t i = 0; i < n; i++) {
        const float* s = center_radius + 4 * (size_t)i;
        const double dx = (double)s[0] - c0[0], dy = (double)s[1] - c0[1], dz = (double)s[2] - c0[2];
        dist[i] = std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    double scene = 0.0;
    if (n) {
        std::vector<double> d(dist);
        for (double& v : d) if (!std::isfinite(v)) v = 0.0;
        std::nth_element(d.begin(), d.begin() + d.size() / 2, d.end());
        scene = d[d.size() / 2];
    }
    float best[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    uint32_t count = 0;
    for (uint32_t i = 0; i < n; i++) {
        const float ratio = (float)((double)center_radius[4 * (size_t)i + 3] / std::max(std::max(dist[i], scene), 1e-30));
        if (!(ratio >= 0.5f)) continue;                             // (NaN: not chosen)
        uint32_t k = count < 4 ? count++ : 4;
        if (k == 4) {                                               // replace the weakest if this one is stronger
            uint32_t w = 0;
            for (uint32_t q = 1; q < 4; q++) if (best[q] < best[w]) w = q;
            if (!(ratio > best[w])) continue;
            k = w;
        }
        best[k] = ratio; out[k] = i;
    }
    return count;
}

// Group order for the two-level filter: a median split of the centres along the longest axis of their box, repeated until a
// part holds at most `group` primitives; the left part of every split is a multiple of `group`, so that only the very last group is short.
// Consecutive runs of `group` entries of the result are the groups; 0xFFFFFFFF pads the last one.  Spheres on the `direct` list (tested for
// every ray anyway) and spheres whose record is not finite (no exact test can accept them) stay out.
// Any order is correct — the nearest-hit key carries the sphere's own index, and the minimum over the keys does not depend on the order the
// pairs are tested in — a compact one keeps the groups' bounding spheres small.
void median_split_order(std::vector<uint32_t>& ids, const float* xyz_stride4, uint32_t group, uint32_t super = 1) {
    struct Part { size_t begin, end; };
    std::vector<Part> stack;
    stack.push_back({ 0, ids.size() });
    while (!stack.empty()) {
        const Part part = stack.back();
        stack.pop_back();
        const size_t count = part.end - part.begin;
        if (count <= group) continue;
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t k = part.begin; k < part.end; k++)
            for (int a = 0; a < 3; a++) { const float c = xyz_stride4[4 * (size_t)ids[k] + a]; lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); }
        int axis = 0;
        for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
        // parts larger than a super group (group x super primitives: one row of the three-level filter) split at multiples of it, so that the
        // `super` leaf groups of a row are one part of the split
        const size_t unit = count > (size_t)group * super ? (size_t)group * super : group;
        size_t half = (count / 2 + unit - 1) / unit * unit;
        if (half >= count) half = count - unit;                     // (count > unit here)
        std::nth_element(ids.begin() + part.begin, ids.begin() + part.begin + half, ids.begin() + part.end,
                         [&](uint32_t x, uint32_t y) { return xyz_stride4[4 * (size_t)x + axis] < xyz_stride4[4 * (size_t)y + axis]; });
        stack.push_back({ part.begin + half, part.end });
        stack.push_back({ part.begin, part.begin + half });
    }
}
std::vector<uint32_t> sphere_group_order(const float* center_radius, uint32_t n, const uint32_t* direct, uint32_t n_direct, uint32_t group, uint32_t super) {
    std::vector<uint32_t> ids;
    ids.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        const float* s = center_radius + 4 * (size_t)i;
        if (is_direct(i, direct, n_direct) || !std::isfinite(s[0]) || !std::isfinite(s[1]) || !std::isfinite(s[2]) || !std::isfinite(s[3] * s[3])) continue;
        ids.push_back(i);
    }
    median_split_order(ids, center_radius, group, super);
    ids.resize((ids.size() + group * super - 1) / (group * super) * (group * super), 0xFFFFFFFFu);
    return ids;
}
// The same for faces, from their bounding spheres (cx, cy, cz, r^2 as k_commit_mesh wrote them): a mesh may list its faces in any order (the
// reference's teddy.obj does), and groups of faces that merely follow each other in the file would span the model.  Faces without a bounded hit
// region (r^2 >= 3e38: always candidates) go last, in groups of their own, so that they make only their own rows always-candidates.
std::vector<uint32_t> face_group_order(const float4* bounds, uint32_t n, uint32_t group, uint32_t super) {
    std::vector<uint32_t> ids, unbounded;
    ids.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        const float4 b = bounds[i];
        if (std::isfinite(b.x) && std::isfinite(b.y) && std::isfinite(b.z) && b.w >= 0.0f && b.w < 3e38f) ids.push_back(i);
        else unbounded.push_back(i);
    }
    median_split_order(ids, reinterpret_cast<const float*>(bounds), group, super);
    ids.resize((ids.size() + group * super - 1) / (group * super) * (group * super), 0xFFFFFFFFu);
    ids.insert(ids.end(), unbounded.begin(), unbounded.end());
    ids.resize((ids.size() + group * super - 1) / (group * super) * (group * super), 0xFFFFFFFFu);
    return ids;
}

bool row_owned(const rt3_params* p, uint32_t y) {
    if (p->tile_count <= 1) return true;
    return ((y / p->tile_rows) % p->tile_count) == p->tile_index;
}

// Workgroups per CU of `kernel` with `lds` bytes of dynamic LDS; the attribute and the query run once per (kernel, lds).
int blocks_per_cu(rt3_ctx* ctx, const void* kernel, int block, size_t lds, int* out) {
    const auto key = std::make_pair(kernel, lds);
    const auto it = ctx->occupancy.find(key);
    if (it != ctx->occupancy.end()) { *out = it->second; return 0; }
    if (lds > 48 * 1024) RT3_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int n = 0;
    RT3_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, lds));
    ctx->occupancy[key] = n;
    *out = n;
    return 0;
}

bool same_bytes(const void* a, const void* b, size_t n) { return std::memcmp(a, b, n) == 0; }

int check_params(rt3_ctx* ctx, const rt3_params* p) {
    if (!p) return fail(ctx, RT3_E_ARG, "params is NULL");
    if (!(p->t_min >= 0.0f) || !(p->t_min < __builtin_inff())) return fail(ctx, RT3_E_ARG, "t_min must be finite and >= 0");
    if (p->flags & ~(RT3_FLAG_GAMMA2 | RT3_FLAG_BLACK_BACKGROUND | RT3_FLAG_REFERENCE_PRIMARY | RT3_FLAG_VARIANCE))
        return fail(ctx, RT3_E_ARG, "unknown bits in flags");
    if (p->width < 2 || p->height < 2) return fail(ctx, RT3_E_ARG, "width and height must be >= 2");
    if ((uint64_t)p->width * p->height > 0x7FFFFFFFull) return fail(ctx, RT3_E_ARG, "frame too large");
    if (p->spp < 1 || p->max_depth < 1) return fail(ctx, RT3_E_ARG, "spp and max_depth must be >= 1");
    if (p->tile_count > 1 && (p->tile_rows == 0 || p->tile_index >= p->tile_count))
        return fail(ctx, RT3_E_ARG, "bad tile_rows / tile_index / tile_count");
    return 0;
}

}  // namespace

extern "C" {

uint32_t rt3_abi_version(void) { return RT3_ABI_VERSION; }

uint32_t rt3_rows_owned(const rt3_params* p) {
    uint32_t n = 0;
    for (uint32_t y = 0; y < p->height; y++) n += row_owned(p, y) ? 1u : 0u;
    return n;
}
uint32_t rt3_row_of_local(const rt3_params* p, uint32_t local_row) {
    if (p->tile_count <= 1) return local_row;
    const uint32_t lb = local_row / p->tile_rows, in = local_row % p->tile_rows;
    return (lb * p->tile_count + p->tile_index) * p->tile_rows + in;
}

rt3_ctx* rt3_create(int device_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("rt3_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                         "); this library has no CPU fallback";
        return nullptr;
    }
    if (device_id < 0 || device_id >= count) { g_create_error = "rt3_create: device_id out of range"; return nullptr; }
    rt3_ctx* ctx = new rt3_ctx();
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreate(&ctx->stream)) != hipSuccess || (e = hipEventCreate(&ctx->ev_begin)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_end)) != hipSuccess || (e = hipEventCreateWithFlags(&ctx->ev_acc, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc((void**)&ctx->d_work, 64)) != hipSuccess || (e = hipMalloc((void**)&ctx->d_casts, 128)) != hipSuccess) {
        g_create_error = std::string("rt3_create: ") + hipGetErrorString(e);
        delete ctx;
        return nullptr;
    }
    ctx->num_cu = prop.multiProcessorCount;
    return ctx;
}

void rt3_destroy(rt3_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    void* bufs[] = { ctx->d_gfaces, ctx->d_verts, ctx->d_face_mats_in, ctx->d_error, ctx->d_tri, ctx->d_tri_mat, ctx->d_tri_kind, ctx->d_tri_bound, ctx->d_tri_frag, ctx->d_sph, ctx->d_sph_frag, ctx->d_sph_frag32, ctx->d_sph_invr, ctx->d_sph_mat, ctx->d_sph_kind,
                     ctx->d_rad, ctx->d_accum, ctx->d_accum_sq, ctx->d_out, ctx->d_work, ctx->d_casts, ctx->d_box, ctx->d_tri_frag_r,
                     ctx->d_tri_gfrag, ctx->d_sph_gfrag, ctx->d_sph_grp, ctx->d_sph_perm, ctx->d_strips, ctx->d_tri_grp, ctx->d_tri_perm, ctx->d_tri_leaf, ctx->d_sph_leaf, ctx->d_tri_rowb, ctx->d_sph_rowb,
                     ctx->d_tri_sfrag, ctx->d_sph_sfrag, ctx->d_tri_srowb, ctx->d_sph_srowb, ctx->d_tri_rec };
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (auto& p : ctx->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->ev_acc) (void)hipEventDestroy(ctx->ev_acc);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* rt3_last_error(const rt3_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt3_set_sample_storage_cap(rt3_ctx* ctx, uint64_t bytes) {
    if (!ctx) return RT3_E_ARG;
    if (bytes < (1ull << 20)) return fail(ctx, RT3_E_ARG, "sample storage cap must be >= 1 MiB");
    ctx->rad_cap_bytes = bytes;
    return 0;
}

int rt3_debug_force_plain_mode_r(rt3_ctx* ctx, int on) {
    if (!ctx) return RT3_E_ARG;
    ctx->force_plain_mode_r = on != 0;
    return 0;
}

int rt3_mesh_begin(rt3_ctx* ctx, uint32_t n_faces, uint32_t n_vertices) {
    if (!ctx) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    if (ctx->d_gfaces) { RT3_HIP(hipFree(ctx->d_gfaces)); ctx->d_gfaces = nullptr; }
    if (ctx->d_verts) { RT3_HIP(hipFree(ctx->d_verts)); ctx->d_verts = nullptr; }
    if (n_faces) RT3_HIP(hipMalloc((void**)&ctx->d_gfaces, (size_t)n_faces * sizeof(rt3_gface)));
    if (n_vertices) RT3_HIP(hipMalloc((void**)&ctx->d_verts, (size_t)n_vertices * sizeof(float4)));
    if (n_faces) RT3_HIP(hipMemsetAsync(ctx->d_gfaces, 0, (size_t)n_faces * sizeof(rt3_gface), ctx->stream));
    if (n_vertices) RT3_HIP(hipMemsetAsync(ctx->d_verts, 0, (size_t)n_vertices * sizeof(float4), ctx->stream));
    ctx->cap_gfaces = n_faces;
    ctx->cap_verts = n_vertices;
    return 0;
}

int rt3_mesh_put(rt3_ctx* ctx, const rt3_gface* faces, uint32_t n_faces, const float* vertices, uint32_t n_vertices,
                 uint32_t face_offset, uint32_t vertex_offset) {
    if (!ctx) return RT3_E_ARG;
    if ((n_faces && !faces) || (n_vertices && !vertices)) return fail(ctx, RT3_E_ARG, "faces / vertices is NULL");
    if ((uint64_t)face_offset + n_faces > ctx->cap_gfaces || (uint64_t)vertex_offset + n_vertices > ctx->cap_verts)
        return fail(ctx, RT3_E_ARG, "rt3_mesh_put: entity does not fit in the buffers sized by rt3_mesh_begin");
    RT3_HIP(hipSetDevice(ctx->device));
    std::vector<rt3_gface> rebased(faces, faces + n_faces);         // transfer_entity: indices += running vertex count
    for (rt3_gface& f : rebased) { f.v1 += vertex_offset; f.v2 += vertex_offset; f.v3 += vertex_offset; }
    if (n_faces) RT3_HIP(hipMemcpyAsync(ctx->d_gfaces + face_offset, rebased.data(), (size_t)n_faces * sizeof(rt3_gface), hipMemcpyHostToDevice, ctx->stream));
    if (n_vertices) RT3_HIP(hipMemcpyAsync(ctx->d_verts + vertex_offset, vertices, (size_t)n_vertices * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));                     // `rebased` is a temporary
    return 0;
}

int rt3_mesh_sphere(rt3_ctx* ctx, const float center[3], float radius, uint32_t n_meridians, uint32_t n_parallels, const float color[3],
                    uint32_t face_offset, uint32_t vertex_offset) {
    if (!ctx) return RT3_E_ARG;
    if (!center || !color || n_meridians < 3 || n_parallels < 3) return fail(ctx, RT3_E_ARG, "rt3_mesh_sphere: need >= 3 meridians and parallels");
    const uint32_t nf = 2 * n_meridians * (n_parallels - 2), nv = 2 + (n_parallels - 2) * n_meridians;
    if ((uint64_t)face_offset + nf > ctx->cap_gfaces || (uint64_t)vertex_offset + nv > ctx->cap_verts)
        return fail(ctx, RT3_E_ARG, "rt3_mesh_sphere: entity does not fit in the buffers sized by rt3_mesh_begin");
    RT3_HIP(hipSetDevice(ctx->device));
    const SphereGen g{ center[0], center[1], center[2], radius, n_meridians, n_parallels, color[0], color[1], color[2], face_offset, vertex_offset };
    const dim3 blk(32, 8);
    hipLaunchKernelGGL(k_prerender_sphere_vertices, dim3((n_meridians + 31) / 32, (n_parallels + 7) / 8), blk, 0, ctx->stream, g, ctx->d_verts);
    RT3_HIP(hipGetLastError());
    // same stream: the faces kernel starts after the vertices are written (the barrier of Sphere.cpp:446-450)
    hipLaunchKernelGGL(k_prerender_sphere_faces, dim3((n_meridians + 31) / 32, (n_parallels - 1 + 7) / 8), blk, 0, ctx->stream, g, ctx->d_gfaces, ctx->d_verts);
    RT3_HIP(hipGetLastError());
    return 0;
}

int rt3_mesh_commit(rt3_ctx* ctx, const rt3_material* face_materials) {
    if (!ctx) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    const uint32_t n = ctx->cap_gfaces, n_pad = (n + 3u) / 4u * 4u;
    for (void** b : { (void**)&ctx->d_tri, (void**)&ctx->d_tri_mat, (void**)&ctx->d_tri_kind, (void**)&ctx->d_tri_bound, (void**)&ctx->d_tri_frag, (void**)&ctx->d_face_mats_in,
                      (void**)&ctx->d_tri_frag_r, (void**)&ctx->d_tri_gfrag, (void**)&ctx->d_tri_grp, (void**)&ctx->d_tri_perm, (void**)&ctx->d_tri_leaf, (void**)&ctx->d_tri_rowb, (void**)&ctx->d_tri_sfrag, (void**)&ctx->d_tri_srowb, (void**)&ctx->d_tri_rec })
        if (*b) { RT3_HIP(hipFree(*b)); *b = nullptr; }
    ctx->n_faces = 0; ctx->n_tri_groups = 0; ctx->n_tri_super = 0;
    if (n == 0) return 0;
    if (face_materials)
        for (uint32_t i = 0; i < n; i++)
            if (face_materials[i].kind > RT3_MAT_DIELECTRIC) return fail(ctx, RT3_E_ARG, "unknown material kind");
    if (!ctx->d_error) RT3_HIP(hipMalloc((void**)&ctx->d_error, 4));
    RT3_HIP(hipMemsetAsync(ctx->d_error, 0, 4, ctx->stream));
    RT3_HIP(hipMalloc((void**)&ctx->d_tri, (size_t)n * 4 * sizeof(float4)));
    RT3_HIP(hipMalloc((void**)&ctx->d_tri_mat, (size_t)n * sizeof(float4)));
    RT3_HIP(hipMalloc((void**)&ctx->d_tri_kind, (size_t)n * sizeof(uint32_t)));
    RT3_HIP(hipMalloc((void**)&ctx->d_tri_bound, (size_t)n_pad * sizeof(float4)));
    const uint32_t n_frag_rows = (n + 31u) / 32u * 32u;
    RT3_HIP(hipMalloc((void**)&ctx->d_tri_frag, (size_t)n_frag_rows * 8 * sizeof(u32x4)));      // 4 operands x 2 lane halves per row
    if (face_materials) {
        RT3_HIP(hipMalloc((void**)&ctx->d_face_mats_in, (size_t)n * sizeof(rt3_material)));
        RT3_HIP(hipMemcpyAsync(ctx->d_face_mats_in, face_materials, (size_t)n * sizeof(rt3_material), hipMemcpyHostToDevice, ctx->stream));
    }
    // the box of the vertices: its centre is what the faces' filter coordinates are taken about (k_commit_mesh and the ray side agree on it
    // through box_centre())
    if (!ctx->d_box) RT3_HIP(hipMalloc((void**)&ctx->d_box, 6 * sizeof(uint32_t)));
    RT3_HIP(hipMemsetAsync(ctx->d_box, 0xFF, 3 * sizeof(uint32_t), ctx->stream));
    RT3_HIP(hipMemsetAsync(ctx->d_box + 3, 0, 3 * sizeof(uint32_t), ctx->stream));
    if (ctx->cap_verts) {
        hipLaunchKernelGGL(k_vertex_box, dim3(std::min<uint32_t>(256u, (ctx->cap_verts + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                           (const float4*)ctx->d_verts, ctx->cap_verts, ctx->d_box);
        RT3_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_commit_mesh, dim3((n_frag_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, ctx->d_gfaces, ctx->d_verts, n, n_pad,
                       ctx->cap_verts, ctx->d_face_mats_in, ctx->d_tri, ctx->d_tri_bound, ctx->d_tri_mat, ctx->d_tri_kind, ctx->d_error,
                       ctx->d_tri_frag, n_frag_rows, (const uint32_t*)ctx->d_box, 1.0f);
    RT3_HIP(hipGetLastError());
    // Mode R's rays all start at the world origin, the path tracer's mostly ON the scene: the filter's margin eps (|C - c|^2 + r^2 +
    // |o - c|^2) is smallest about the point half-way to the mesh here and about the mesh's own centre there.  Mode R therefore keeps
    // fragments of its own, built here by the same kernel (with centre_scale 0.5 it writes nothing else) while the merged entity
    // buffers are known to be the ones these faces came from: a render never reads d_gfaces / d_verts, so an rt3_mesh_begin without a
    // commit leaves the committed scene renderable.
    RT3_HIP(hipMalloc((void**)&ctx->d_tri_frag_r, (size_t)n_frag_rows * 8 * sizeof(u32x4)));
    hipLaunchKernelGGL(k_commit_mesh, dim3((n_frag_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, ctx->d_gfaces, ctx->d_verts, n, n_pad,
                       ctx->cap_verts, ctx->d_face_mats_in, ctx->d_tri, ctx->d_tri_bound, ctx->d_tri_mat, ctx->d_tri_kind, ctx->d_error,
                       (u32x4*)ctx->d_tri_frag_r, n_frag_rows, (const uint32_t*)ctx->d_box, 0.5f);
    RT3_HIP(hipGetLastError());
    uint32_t err = 0, box[6];
    RT3_HIP(hipMemcpyAsync(&err, ctx->d_error, 4, hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipMemcpyAsync(box, ctx->d_box, sizeof(box), hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    box_centre(box, ctx->tri_centre);
    if (err) return fail(ctx, RT3_E_ARG, "a face references a vertex out of range");
    // rows of the two-level filter: groups of kGroupTri faces in the order of a spatial median split of their bounds (read back once per commit:
    // 16 bytes per face); the members' bounds in group order + the face index of each
    {
        std::vector<float4> bounds(n);
        RT3_HIP(hipMemcpy(bounds.data(), ctx->d_tri_bound, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
        const std::vector<uint32_t> order = face_group_order(bounds.data(), n, kGroupTri, kSuper);
        std::vector<float4> grp(order.size(), kPadSphere);
        for (size_t k = 0; k < order.size(); k++) if (order[k] != 0xFFFFFFFFu) grp[k] = bounds[order[k]];
        int rc;
        if ((rc = upload(ctx, &ctx->d_tri_grp, grp)) || (rc = upload(ctx, &ctx->d_tri_perm, order))) return rc;
        RT3_HIP(hipMalloc((void**)&ctx->d_tri_rec, order.size() * 4 * sizeof(float4)));
        hipLaunchKernelGGL(k_gather_face_records, dim3(((uint32_t)order.size() * 4u + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_tri,
                           (const uint32_t*)ctx->d_tri_perm, (uint32_t)order.size(), ctx->d_tri_rec);
        RT3_HIP(hipGetLastError());
        ctx->n_tri_leaves = (uint32_t)(order.size() / kGroupTri);
        ctx->n_tri_groups = ctx->n_tri_leaves / kSuper;                // rows the matrix filter scans
        const float4* row_members = ctx->d_tri_grp;
        uint32_t row_entries = (uint32_t)order.size(), row_group = kGroupTri;
        if (kSuper > 1) {                                           // three levels: the leaves' bounds are records of their own, a row bounds kSuper of them
            RT3_HIP(hipMalloc((void**)&ctx->d_tri_leaf, (size_t)ctx->n_tri_leaves * sizeof(float4)));
            hipLaunchKernelGGL(k_group_bounds, dim3((ctx->n_tri_leaves + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_tri_grp, (uint32_t)order.size(),
                               kGroupTri, ctx->n_tri_leaves, (const uint32_t*)ctx->d_box, 0.0f, 0.0f, 0.0f, ctx->d_tri_leaf);
            RT3_HIP(hipGetLastError());
            row_members = ctx->d_tri_leaf; row_entries = ctx->n_tri_leaves; row_group = kSuper;
        }
        const uint32_t n_group_rows = (ctx->n_tri_groups + 31u) / 32u * 32u;
        RT3_HIP(hipMalloc((void**)&ctx->d_tri_gfrag, (size_t)n_group_rows * 4 * sizeof(u32x4)));
        hipLaunchKernelGGL(k_group_frags, dim3((n_group_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, row_members, row_entries,
                           row_group, n_group_rows, (const uint32_t*)ctx->d_box, 0.0f, 0.0f, 0.0f, ctx->d_tri_gfrag);
        RT3_HIP(hipGetLastError());
        if (kSuper > 1) {                                           // the same rows as f32 records
            RT3_HIP(hipMalloc((void**)&ctx->d_tri_rowb, (size_t)n_group_rows * sizeof(float4)));
            hipLaunchKernelGGL(k_group_bounds, dim3((n_group_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, row_members, row_entries,
                               row_group, n_group_rows, (const uint32_t*)ctx->d_box, 0.0f, 0.0f, 0.0f, ctx->d_tri_rowb);
            RT3_HIP(hipGetLastError());
            // four levels: super-rows of kSuper rows, as fragments for the matrix filter and as f32 records
            ctx->n_tri_super = (ctx->n_tri_groups + kSuper - 1u) / kSuper;
            const uint32_t n_super_rows = (ctx->n_tri_super + 31u) / 32u * 32u;
            RT3_HIP(hipMalloc((void**)&ctx->d_tri_sfrag, (size_t)n_super_rows * 4 * sizeof(u32x4)));
            RT3_HIP(hipMalloc((void**)&ctx->d_tri_srowb, (size_t)n_super_rows * sizeof(float4)));
            hipLaunchKernelGGL(k_group_frags, dim3((n_super_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_tri_rowb, n_group_rows,
                               kSuper, n_super_rows, (const uint32_t*)ctx->d_box, 0.0f, 0.0f, 0.0f, ctx->d_tri_sfrag);
            hipLaunchKernelGGL(k_group_bounds, dim3((n_super_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_tri_rowb, n_group_rows,
                               kSuper, n_super_rows, (const uint32_t*)ctx->d_box, 0.0f, 0.0f, 0.0f, ctx->d_tri_srowb);
            RT3_HIP(hipGetLastError());
        }
        RT3_HIP(hipStreamSynchronize(ctx->stream));                 // a render may come on another stream
    }
    ctx->n_faces = n;
    return 0;
}

int rt3_mesh_download(rt3_ctx* ctx, rt3_gface* faces, float* vertices) {
    if (!ctx) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    if (faces && ctx->cap_gfaces) RT3_HIP(hipMemcpy(faces, ctx->d_gfaces, (size_t)ctx->cap_gfaces * sizeof(rt3_gface), hipMemcpyDeviceToHost));
    if (vertices && ctx->cap_verts) RT3_HIP(hipMemcpy(vertices, ctx->d_verts, (size_t)ctx->cap_verts * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}

int rt3_set_mesh(rt3_ctx* ctx, const rt3_gface* faces, uint32_t n_faces, const float* vertices, uint32_t n_vertices,
                 const rt3_material* face_materials) {
    if (!ctx) return RT3_E_ARG;
    if (n_faces != 0 && (!faces || !vertices)) return fail(ctx, RT3_E_ARG, "faces / vertices is NULL");
    int rc;
    if ((rc = rt3_mesh_begin(ctx, n_faces, n_vertices))) return rc;
    if ((rc = rt3_mesh_put(ctx, faces, n_faces, vertices, n_vertices, 0, 0))) return rc;
    return rt3_mesh_commit(ctx, face_materials);
}

int rt3_set_spheres(rt3_ctx* ctx, const float* center_radius, const rt3_material* materials, uint32_t n) {
    if (!ctx) return RT3_E_ARG;
    if (n != 0 && (!center_radius || !materials)) return fail(ctx, RT3_E_ARG, "center_radius / materials is NULL");
    RT3_HIP(hipSetDevice(ctx->device));
    std::vector<float4> sph(((size_t)n + 3) / 4 * 4, kPadSphere), mat(n);                 // scan works in groups of 4
    std::vector<float> invr(n);
    std::vector<uint32_t> kind(n);
    for (uint32_t i = 0; i < n; i++) {
        const float* s = center_radius + 4 * (size_t)i;
        if (!(s[3] > 0.0f)) return fail(ctx, RT3_E_ARG, "sphere " + std::to_string(i) + " has a non-positive radius");
        if (materials[i].kind > RT3_MAT_DIELECTRIC) return fail(ctx, RT3_E_ARG, "unknown material kind");
        sph[i] = make_float4(s[0], s[1], s[2], s[3] * s[3]);
        invr[i] = 1.0f / s[3];
        mat[i] = pack_material(materials[i]);
        kind[i] = materials[i].kind;
    }
    int rc;
    sphere_filter_centre(center_radius, n, ctx->sph_centre);
    ctx->n_direct = sphere_direct_list(center_radius, n, ctx->sph_centre, ctx->direct);
    if ((rc = upload(ctx, &ctx->d_sph_frag, build_sphere_frags(center_radius, n, ctx->sph_centre, ctx->direct, ctx->n_direct)))) return rc;      // k_trace_mfma (K = 64, 32x32x16)
    if ((rc = upload(ctx, &ctx->d_sph_frag32, build_sphere_frags32(center_radius, n, ctx->sph_centre, ctx->direct, ctx->n_direct)))) return rc;  // K = 32 form
    if ((rc = upload(ctx, &ctx->d_sph, sph)) || (rc = upload(ctx, &ctx->d_sph_invr, invr)) ||
        (rc = upload(ctx, &ctx->d_sph_mat, mat)) || (rc = upload(ctx, &ctx->d_sph_kind, kind)))
        return rc;
    // rows of the two-level filter: groups of kGroupSph spheres in the order of a spatial median split
    {
        const std::vector<uint32_t> order = sphere_group_order(center_radius, n, ctx->direct, ctx->n_direct, kGroupSph, kSuper);
        std::vector<float4> grp(order.size(), kPadSphere);
        for (size_t k = 0; k < order.size(); k++) if (order[k] != 0xFFFFFFFFu) grp[k] = sph[order[k]];
        if ((rc = upload(ctx, &ctx->d_sph_grp, grp)) || (rc = upload(ctx, &ctx->d_sph_perm, order))) return rc;
        for (void** b : { (void**)&ctx->d_sph_gfrag, (void**)&ctx->d_sph_leaf, (void**)&ctx->d_sph_rowb, (void**)&ctx->d_sph_sfrag, (void**)&ctx->d_sph_srowb })
            if (*b) { RT3_HIP(hipFree(*b)); *b = nullptr; }
        ctx->n_sph_leaves = (uint32_t)(order.size() / kGroupSph);
        ctx->n_sph_groups = ctx->n_sph_leaves / kSuper;                // rows the matrix filter scans
        ctx->n_sph_super = 0;                                          // (no groups: every sphere is on the direct list or unusable)
        if (ctx->n_sph_groups) {
            const float4* row_members = ctx->d_sph_grp;
            uint32_t row_entries = (uint32_t)order.size(), row_group = kGroupSph;
            if (kSuper > 1) {
                RT3_HIP(hipMalloc((void**)&ctx->d_sph_leaf, (size_t)ctx->n_sph_leaves * sizeof(float4)));
                hipLaunchKernelGGL(k_group_bounds, dim3((ctx->n_sph_leaves + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_sph_grp,
                                   (uint32_t)order.size(), kGroupSph, ctx->n_sph_leaves, (const uint32_t*)nullptr, ctx->sph_centre[0], ctx->sph_centre[1], ctx->sph_centre[2],
                                   ctx->d_sph_leaf);
                RT3_HIP(hipGetLastError());
                row_members = ctx->d_sph_leaf; row_entries = ctx->n_sph_leaves; row_group = kSuper;
            }
            const uint32_t n_group_rows = (ctx->n_sph_groups + 31u) / 32u * 32u;
            RT3_HIP(hipMalloc((void**)&ctx->d_sph_gfrag, (size_t)n_group_rows * 4 * sizeof(u32x4)));
            hipLaunchKernelGGL(k_group_frags, dim3((n_group_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, row_members, row_entries, row_group,
                               n_group_rows, (const uint32_t*)nullptr, ctx->sph_centre[0], ctx->sph_centre[1], ctx->sph_centre[2], ctx->d_sph_gfrag);
            RT3_HIP(hipGetLastError());
            if (kSuper > 1) {
                RT3_HIP(hipMalloc((void**)&ctx->d_sph_rowb, (size_t)n_group_rows * sizeof(float4)));
                hipLaunchKernelGGL(k_group_bounds, dim3((n_group_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, row_members, row_entries, row_group,
                                   n_group_rows, (const uint32_t*)nullptr, ctx->sph_centre[0], ctx->sph_centre[1], ctx->sph_centre[2], ctx->d_sph_rowb);
                RT3_HIP(hipGetLastError());
                ctx->n_sph_super = (ctx->n_sph_groups + kSuper - 1u) / kSuper;
                const uint32_t n_super_rows = (ctx->n_sph_super + 31u) / 32u * 32u;
                RT3_HIP(hipMalloc((void**)&ctx->d_sph_sfrag, (size_t)n_super_rows * 4 * sizeof(u32x4)));
                RT3_HIP(hipMalloc((void**)&ctx->d_sph_srowb, (size_t)n_super_rows * sizeof(float4)));
                hipLaunchKernelGGL(k_group_frags, dim3((n_super_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_sph_rowb, n_group_rows, kSuper,
                                   n_super_rows, (const uint32_t*)nullptr, ctx->sph_centre[0], ctx->sph_centre[1], ctx->sph_centre[2], ctx->d_sph_sfrag);
                hipLaunchKernelGGL(k_group_bounds, dim3((n_super_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->d_sph_rowb, n_group_rows, kSuper,
                                   n_super_rows, (const uint32_t*)nullptr, ctx->sph_centre[0], ctx->sph_centre[1], ctx->sph_centre[2], ctx->d_sph_srowb);
                RT3_HIP(hipGetLastError());
            }
            RT3_HIP(hipStreamSynchronize(ctx->stream));             // a render may come on another stream
        }
    }
    ctx->n_sph = n;
    return 0;
}

int rt3_debug_force_brute(rt3_ctx* ctx, int on) {
    if (!ctx) return RT3_E_ARG;
    ctx->force_brute = on != 0;
    return 0;
}

int rt3_debug_force_flat_filter(rt3_ctx* ctx, int on) {
    if (!ctx) return RT3_E_ARG;
    ctx->force_flat = on != 0;
    return 0;
}

void* rt3_stream(rt3_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt3_synchronize(rt3_ctx* ctx) {
    if (!ctx) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

void* rt3_device_alloc_words(rt3_ctx* ctx, uint64_t n_words) {
    if (!ctx || n_words == 0) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p, n_words * 4) != hipSuccess) { ctx->err = "rt3_device_alloc_words: hipMalloc failed"; return nullptr; }
    return p;
}
void rt3_device_free(rt3_ctx* ctx, void* d_ptr) {
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipFree(d_ptr);
}
int rt3_device_read_words(rt3_ctx* ctx, const void* d_ptr, uint64_t n_words, uint32_t* out) {
    if (!ctx) return RT3_E_ARG;
    if (!d_ptr || !out) return fail(ctx, RT3_E_ARG, "rt3_device_read_words: NULL buffer");
    RT3_HIP(hipSetDevice(ctx->device));
    RT3_HIP(hipMemcpyAsync(out, d_ptr, n_words * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int rt3_render_device(rt3_ctx* ctx, const rt3_camera* cam, uint32_t width, uint32_t height, void* d_out, void* stream_) {
    if (!ctx) return RT3_E_ARG;
    if (!cam || !d_out) return fail(ctx, RT3_E_ARG, "cam / d_out_pixels is NULL");
    if (width < 2 || height < 2 || (uint64_t)width * height > 0x7FFFFFFFull) return fail(ctx, RT3_E_ARG, "bad frame size");
    RT3_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;     // rt3.h: NULL = the context's own stream
    ctx->rendered = false;                                          // stats are valid again once every launch below has been issued
    ctx->ev_used = 0;
    hipEvent_t a, b;
    int rc = take_event_pair(ctx, &a, &b);
    if (rc) return rc;
    const uint32_t npix = width * height;
    // k_mode_r_mfma / k_mode_r_fast need n.o == 0 exactly (camera at the origin, as Camera::update always builds it) and finite rays;
    // any other camera takes the plain brute-force kernel, which reproduces the reference for every input.
    const bool at_origin = cam->origin[0] == 0.0f && cam->origin[1] == 0.0f && cam->origin[2] == 0.0f;
    const bool use_mfma = at_origin && !ctx->force_plain_mode_r && !getenv("RT3_NO_MFMA") && ctx->n_faces > 0;
    if (use_mfma) {
        int per_cu = 0;
        if ((rc = blocks_per_cu(ctx, (const void*)k_mode_r_mfma, kMB, kTiledLdsBytes, &per_cu))) return rc;
        if (per_cu < 1) return fail(ctx, RT3_E_DEVICE, "k_mode_r_mfma does not fit on a CU");
        if (!ctx->d_tri_frag_r) return fail(ctx, RT3_E_STATE, "internal: the committed mesh has no Mode-R fragments");     // (built by rt3_mesh_commit)
    }
    RT3_HIP(hipEventRecord(ctx->ev_begin, stream));
    RT3_HIP(hipEventRecord(a, stream));
    if (use_mfma)
        hipLaunchKernelGGL(k_mode_r_mfma, dim3((npix + kMB - 1) / kMB), dim3(kMB), kTiledLdsBytes, stream,
                           ctx->d_tri, (const u32x4*)ctx->d_tri_frag_r, ctx->d_tri_mat, ctx->n_faces, cam_dev(cam), width, height, (uint32_t*)d_out,
                           0.5f * ctx->tri_centre[0], 0.5f * ctx->tri_centre[1], 0.5f * ctx->tri_centre[2]);
    else if (at_origin && !ctx->force_plain_mode_r)
        hipLaunchKernelGGL(k_mode_r_fast, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                           ctx->d_tri, ctx->d_tri_bound, ctx->d_tri_mat, ctx->n_faces, cam_dev(cam), width, height, (uint32_t*)d_out);
    else
        hipLaunchKernelGGL(k_mode_r, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                           ctx->d_tri, ctx->d_tri_mat, ctx->n_faces, cam_dev(cam), width, height, (uint32_t*)d_out);
    RT3_HIP(hipGetLastError());
    RT3_HIP(hipEventRecord(b, stream));
    RT3_HIP(hipEventRecord(ctx->ev_end, stream));
    ctx->last_stream = stream;
    ctx->last_samples = npix;
    ctx->last_was_path = false;
    ctx->rendered = true;
    return 0;
}

int rt3_render(rt3_ctx* ctx, const rt3_camera* cam, uint32_t width, uint32_t height, uint32_t* out_pixels) {
    if (!ctx) return RT3_E_ARG;
    if (!out_pixels) return fail(ctx, RT3_E_ARG, "out_pixels is NULL");
    if (width < 2 || height < 2 || (uint64_t)width * height > 0x7FFFFFFFull) return fail(ctx, RT3_E_ARG, "bad frame size");
    RT3_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)width * height;
    int rc = ensure(ctx, &ctx->d_out, &ctx->out_entries, npix);
    if (rc) return rc;
    if ((rc = rt3_render_device(ctx, cam, width, height, ctx->d_out, ctx->stream))) return rc;
    RT3_HIP(hipMemcpyAsync(out_pixels, ctx->d_out, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int rt3_render_path_range_device(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, uint32_t sample_begin, uint32_t sample_count,
                                 void* d_out, void* stream_) {
    if (!ctx) return RT3_E_ARG;
    if (!cam || !d_out) return fail(ctx, RT3_E_ARG, "cam / d_out_pixels is NULL");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (sample_count == 0 || (uint64_t)sample_begin + sample_count > p->spp) return fail(ctx, RT3_E_ARG, "sample range outside [0, spp)");
    if (ctx->n_sph == 0 && ctx->n_faces == 0) return fail(ctx, RT3_E_STATE, "no scene: call rt3_set_spheres / rt3_set_mesh first");
    const bool ref = (p->flags & RT3_FLAG_REFERENCE_PRIMARY) != 0, var = (p->flags & RT3_FLAG_VARIANCE) != 0;
    if (ref && ctx->n_sph != 0) return fail(ctx, RT3_E_ARG, "RT3_FLAG_REFERENCE_PRIMARY needs a triangle-only scene");
    if (ctx->n_faces >= (1u << kPairLaneShift) - 32u || ctx->n_sph >= (1u << kPairLaneShift) - 32u) return fail(ctx, RT3_E_ARG, "too many primitives");
    RT3_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;     // rt3.h: NULL = the context's own stream
    // d_rad, d_accum and the counters belong to the context, not to a stream: whatever stream the previous render ran on, this one
    // starts behind it (a wait on the device; nothing if it is the same stream)
    if (ctx->ev_acc_recorded) RT3_HIP(hipStreamWaitEvent(stream, ctx->ev_acc, 0));

    const uint32_t rows = rt3_rows_owned(p);
    const uint32_t npix = rows * p->width;
    if (sample_begin != 0) {                                        // a continuation: of this very accumulation?
        if (!ctx->acc_valid || ctx->acc_done != sample_begin || ctx->acc_npix != npix || !same_bytes(&ctx->acc_params, p, sizeof *p) ||
            !same_bytes(&ctx->acc_cam, cam, sizeof *cam))
            return fail(ctx, RT3_E_STATE, "sample_begin does not continue the accumulation held by this context (same camera, params and "
                                          "samples done are required; start with sample_begin = 0 or rt3_accum_upload)");
    }
    ctx->rendered = false;                                          // stats: valid again once every launch below has been issued
    ctx->acc_valid = false;                                         // accumulation: valid again once this call has been issued completely
    ctx->ev_used = 0;
    if (npix == 0) {
        RT3_HIP(hipEventRecord(ctx->ev_begin, stream));
        RT3_HIP(hipMemsetAsync(ctx->d_casts, 0, 64, stream));
        RT3_HIP(hipEventRecord(ctx->ev_end, stream));
        RT3_HIP(hipEventRecord(ctx->ev_acc, stream)); ctx->ev_acc_recorded = true;
        ctx->last_stream = stream; ctx->last_samples = 0; ctx->last_was_path = true; ctx->rendered = true;
        ctx->acc_valid = true; ctx->acc_params = *p; ctx->acc_cam = *cam; ctx->acc_done = sample_begin + sample_count; ctx->acc_npix = 0;
        return 0;
    }

    // batch size: per-sample storage of 12 B per (pixel, sample), capped
    uint64_t per_spp = (uint64_t)npix * sizeof(Rgb);
    uint32_t batch = (uint32_t)std::min<uint64_t>(sample_count, std::max<uint64_t>(1, ctx->rad_cap_bytes / per_spp));
    batch = (uint32_t)std::min<uint64_t>(batch, 0x7FFF0000ull / npix);
    if (batch == 0) return fail(ctx, RT3_E_ARG, "frame too large for one sample batch");
    if ((rc = ensure(ctx, &ctx->d_rad, &ctx->rad_entries, (size_t)npix * batch))) return rc;
    if (sample_begin == 0) {
        if ((rc = ensure(ctx, &ctx->d_accum, &ctx->accum_entries, (size_t)npix))) return rc;
        if (var && (rc = ensure(ctx, &ctx->d_accum_sq, &ctx->accum_sq_entries, (size_t)npix))) return rc;
    }

    TraceArgs A;
    std::memset(&A, 0, sizeof A);
    A.sph = ctx->d_sph; A.sph_invr = ctx->d_sph_invr; A.sph_mat = ctx->d_sph_mat; A.sph_kind = ctx->d_sph_kind; A.n_sph = ctx->n_sph;
    A.tri = ctx->d_tri; A.tri_mat = ctx->d_tri_mat; A.tri_kind = ctx->d_tri_kind; A.tri_bound = ctx->d_tri_bound; A.n_tri = ctx->n_faces;
    A.cam = cam_dev(cam);
    A.lens_radius = p->lens_radius;
    {
        const float* h = cam->horizontal; const float* v = cam->vertical;
        const float lh = std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
        const float lv = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        A.lux = h[0] / lh; A.luy = h[1] / lh; A.luz = h[2] / lh;
        A.lvx = v[0] / lv; A.lvy = v[1] / lv; A.lvz = v[2] / lv;
    }
    A.width = p->width; A.height = p->height; A.spp = p->spp; A.max_depth = p->max_depth; A.seed = p->seed; A.flags = p->flags;
    {
        uint32_t e = (uint32_t)std::sqrt((double)p->spp);
        while (e * e > p->spp) e--;
        while ((e + 1) * (e + 1) <= p->spp) e++;
        A.edge = (e * e == p->spp && p->spp > 1) ? e : 0;          // stratified only for perfect squares (v4:199)
    }
    A.t_min = p->t_min;
    A.tile_rows = p->tile_rows; A.tile_index = p->tile_index; A.tile_count = p->tile_count;
    A.npix = npix;
    A.div_npix = make_fastdiv(npix); A.div_width = make_fastdiv(p->width);
    A.div_edge = make_fastdiv(A.edge ? A.edge : 1); A.div_tile_rows = make_fastdiv(p->tile_count > 1 ? p->tile_rows : 1);
    if (!fastdiv_ok(npix, 0x7FFFFFFFu) || !fastdiv_ok(p->width, 0x7FFFFFFFu) || !fastdiv_ok(A.edge ? A.edge : 1, p->spp) ||
        !fastdiv_ok(p->tile_count > 1 ? p->tile_rows : 1, p->height))
        return fail(ctx, RT3_E_DEVICE, "internal: magic-number division self-check failed");
    A.rad = ctx->d_rad; A.work_counter = ctx->d_work; A.cast_counter = ctx->d_casts;
    A.fcx = ctx->sph_centre[0]; A.fcy = ctx->sph_centre[1]; A.fcz = ctx->sph_centre[2];
    A.tcx = ctx->tri_centre[0]; A.tcy = ctx->tri_centre[1]; A.tcz = ctx->tri_centre[2];
    A.n_direct = ctx->n_direct;
    for (int i = 0; i < 4; i++) A.direct[i] = ctx->direct[i];

    // ---- which kernel
    //   brute         every ray against every primitive (debug switch / RT3_BRUTE=1; REFERENCE_PRIMARY with a camera off the origin or a lens)
    //   mfma_single   sphere scenes of <= 512 spheres, everything in LDS (the bench kernel)
    //   mfma tiled    every other scene
    //   valu          RT3_NO_MFMA=1: the vector-ALU scans (A/B reference)
    using TraceKernel = void (*)(const TraceArgs);
    using TiledKernel = void (*)(const TraceArgs, const u32x4*, const u32x4*);
    const bool has_tri = ctx->n_faces > 0, has_sph = ctx->n_sph > 0;
    const bool cam_at_origin = cam->origin[0] == 0.0f && cam->origin[1] == 0.0f && cam->origin[2] == 0.0f;
    const bool brute = ctx->force_brute || getenv("RT3_BRUTE") || (ref && !(cam_at_origin && !(p->lens_radius > 0.0f)));
    const bool use_mfma = !brute && !getenv("RT3_NO_MFMA");
    const bool mfma_single = use_mfma && !has_tri && has_sph && ctx->n_sph <= kMfmaSphMax && !getenv("RT3_FORCE_TILED");   // (A/B knob)
    const bool single_k64 = mfma_single && getenv("RT3_MFMA_K64") != nullptr;
    const bool sph_lds = has_sph && ctx->n_sph <= kSphLdsMax;
    const bool grouped = kGroupTri > 1 && kGroupSph > 1 && !ctx->force_flat && !getenv("RT3_NO_GROUPS");
    A.n_tri_rows = grouped ? ctx->n_tri_groups : ctx->n_faces;
    A.n_sph_rows = grouped ? ctx->n_sph_groups : ctx->n_sph;
    A.sph_grp = ctx->d_sph_grp; A.sph_perm = ctx->d_sph_perm; A.tri_grp = ctx->d_tri_grp; A.tri_perm = ctx->d_tri_perm; A.tri_rec = ctx->d_tri_rec;
    A.tri_leaf = ctx->d_tri_leaf; A.sph_leaf = ctx->d_sph_leaf; A.n_tri_leaves = ctx->n_tri_leaves; A.n_sph_leaves = ctx->n_sph_leaves;
    A.tri_rowb = ctx->d_tri_rowb; A.sph_rowb = ctx->d_sph_rowb;
    const uint32_t mfma_blocks = (ctx->n_sph + 31u) / 32u;
    TraceKernel plain = nullptr;
    TiledKernel tiled = nullptr;
    bool resident = false;
    uint32_t levels = 0;                                            // k_trace_levels: 3 | 4
    size_t lds = 0;
    int block = kBlock;
    const void* kptr = nullptr;
    if (brute) {
        plain = ref ? k_trace_brute<true> : k_trace_brute<false>;
        kptr = (const void*)plain;
    } else if (mfma_single) {
        // k_trace_mfma32 (K = 32 form, pair list); RT3_MFMA_K64=1: k_trace_mfma, round 1's K = 64 form on v_mfma_f32_32x32x16_bf16 (A/B reference)
        lds = single_k64 ? (size_t)mfma_blocks * (4096 + 32 * (16 + 16 + 4 + 4)) + (size_t)kBitmapBytes
                         : (size_t)mfma_blocks * (2048 + 32 * (16 + 16 + 4 + 4)) + (size_t)kBitmapBytes + (size_t)kMB * 8 + (size_t)(kMB / 64) * kPairCap * 4;
        block = kMB;
        kptr = single_k64 ? (const void*)k_trace_mfma : (const void*)k_trace_mfma32;
    } else if (use_mfma) {
        // the two-level filter (rows = groups of primitives) unless RT3_NO_GROUPS=1 asks for the flat one (A/B reference, tests)
        constexpr uint32_t GT = kGroupTri, GS = kGroupSph, SUP = kSuper;
        const uint32_t row_blocks = (has_tri ? (A.n_tri_rows + 31u) / 32u : 0u) + (has_sph ? (A.n_sph_rows + 31u) / 32u : 0u);
        const uint32_t super_blocks = (has_tri ? (ctx->n_tri_super + 31u) / 32u : 0u) + (has_sph ? (ctx->n_sph_super + 31u) / 32u : 0u);
        // While the rows of 64 fit in LDS (<= kResidentBlocks row blocks, 112 000 primitives: both BASELINE scenes) k_trace_mfma_tiled's resident three-level
        // form runs; beyond, k_trace_levels (rt3_level_filter.hpp) with FOUR levels — the matrix cores scan super-rows of 512, resident up to 570 000
        // primitives, through a tile after that.  RT3_LEVELS=3|4 forces k_trace_levels with that many levels, RT3_OLD_GROUPS=1 the nested form (A/B, tests)
        const char* force_levels = getenv("RT3_LEVELS");
        const bool no_res_env = getenv("RT3_NO_RESIDENT") != nullptr;
        const bool lev_kernel = grouped && SUP > 1 && !getenv("RT3_OLD_GROUPS") && (force_levels != nullptr || row_blocks > kResidentBlocks);
        if (lev_kernel) {
            levels = force_levels ? (atoi(force_levels) == 4 ? 4u : 3u) : 4u;
            const uint32_t top_blocks = levels == 4 ? super_blocks : row_blocks;
            resident = !no_res_env && top_blocks <= lev_resident_blocks(levels);
            A.n_tri_top = levels == 4 ? ctx->n_tri_super : ctx->n_tri_groups; A.n_sph_top = levels == 4 ? ctx->n_sph_super : ctx->n_sph_groups;
            A.tri_topb = levels == 4 ? ctx->d_tri_srowb : ctx->d_tri_rowb; A.sph_topb = levels == 4 ? ctx->d_sph_srowb : ctx->d_sph_rowb;
#define RT3_LEV(L, R) (has_tri ? (has_sph ? k_trace_levels<true, true, false, L, R> : (ref ? k_trace_levels<true, false, true, L, R> : k_trace_levels<true, false, false, L, R>)) \
                               : k_trace_levels<false, true, false, L, R>)
            tiled = levels == 4 ? (resident ? RT3_LEV(4, true) : RT3_LEV(4, false)) : (resident ? RT3_LEV(3, true) : RT3_LEV(3, false));
#undef RT3_LEV
            lds = lev_lds_fixed(levels, resident) + (resident ? (size_t)top_blocks * 2048u : 0u);
        } else {
        resident = grouped && SUP > 1 && row_blocks <= kResidentBlocks && !getenv("RT3_NO_RESIDENT");      // all rows fit in LDS: no tiles, no barriers
        if (resident)
            tiled = has_tri ? (has_sph ? k_trace_mfma_tiled<true, true, false, GT, GS, SUP, true> : (ref ? k_trace_mfma_tiled<true, false, true, GT, 1, SUP, true> : k_trace_mfma_tiled<true, false, false, GT, 1, SUP, true>))
                            : k_trace_mfma_tiled<false, true, false, 1, GS, SUP, true>;
        else if (grouped)
            tiled = has_tri ? (has_sph ? k_trace_mfma_tiled<true, true, false, GT, GS, SUP> : (ref ? k_trace_mfma_tiled<true, false, true, GT, 1, SUP> : k_trace_mfma_tiled<true, false, false, GT, 1, SUP>))
                            : k_trace_mfma_tiled<false, true, false, 1, GS, SUP>;
        else
            tiled = has_tri ? (has_sph ? k_trace_mfma_tiled<true, true, false> : (ref ? k_trace_mfma_tiled<true, false, true> : k_trace_mfma_tiled<true, false, false>))
                            : k_trace_mfma_tiled<false, true, false>;
        lds = resident ? (size_t)row_blocks * 2048u + (size_t)kBmBlocksRes * kTB * 4u + (size_t)kTB * 8u + (size_t)(kTB / 64u) * kPairCap * 4u * 3u : kTraceTiledLdsBytes;
        }
        block = kTB;
        kptr = (const void*)tiled;

    } else {
        lds = sph_lds ? (size_t)ctx->n_sph * sizeof(float4) : 0;
        plain = has_tri ? (has_sph ? (sph_lds ? k_trace<true, true, true> : k_trace<true, true, false>)
                                   : (ref ? k_trace<true, false, false, true> : k_trace<true, false, false>))
                        : (sph_lds ? k_trace<false, true, true> : k_trace<false, true, false>);
        kptr = (const void*)plain;
    }
    int per_cu = 0;
    if ((rc = blocks_per_cu(ctx, kptr, block, lds, &per_cu))) return rc;
    if (per_cu < 1) return fail(ctx, RT3_E_DEVICE, "the trace kernel does not fit on a CU");
    per_cu = std::min(per_cu, 8);
    if (tiled && grouped) {                                         // one strip per wave of the largest grid this launch configuration can have
        if ((rc = ensure(ctx, &ctx->d_strips, &ctx->strip_entries, (size_t)ctx->num_cu * per_cu * (kTB / 64u) * kStripPairs))) return rc;
        A.pair_strips = ctx->d_strips;
    }

    RT3_HIP(hipEventRecord(ctx->ev_begin, stream));
    RT3_HIP(hipMemsetAsync(ctx->d_casts, 0, 128, stream));
#ifdef RT3_PROFILE
    RT3_HIP(hipMemsetAsync(ctx->d_casts + 6, 0xFF, 8, stream));
    RT3_HIP(hipMemsetAsync(ctx->d_casts + 8, 0xFF, 16, stream));       // [8] first wave start, [9] first time a wave found the queue empty
#endif
    for (uint32_t s0 = sample_begin; s0 < sample_begin + sample_count; s0 += batch) {
        const uint32_t ns = std::min(batch, sample_begin + sample_count - s0);
        A.s0 = s0;
        A.total = npix * ns;
        const uint32_t waves_per_block = (uint32_t)block / 64u;
        const uint32_t want_blocks = (A.total + kWorkChunk * waves_per_block - 1) / (kWorkChunk * waves_per_block);
        const uint32_t grid = std::max(1u, std::min<uint32_t>((uint32_t)(ctx->num_cu * per_cu), want_blocks));
        hipEvent_t a, b;
        if ((rc = take_event_pair(ctx, &a, &b))) return rc;
        RT3_HIP(hipMemsetAsync(ctx->d_work, 0, 4, stream));
        RT3_HIP(hipEventRecord(a, stream));
        if (mfma_single && single_k64) hipLaunchKernelGGL(k_trace_mfma, dim3(grid), dim3(kMB), lds, stream, A, (const u32x4*)ctx->d_sph_frag, mfma_blocks);
        else if (mfma_single) hipLaunchKernelGGL(k_trace_mfma32, dim3(grid), dim3(kMB), lds, stream, A, (const u32x4*)ctx->d_sph_frag32, mfma_blocks);
        else if (tiled && levels == 4) hipLaunchKernelGGL(tiled, dim3(grid), dim3(kTB), lds, stream, A, (const u32x4*)ctx->d_tri_sfrag, (const u32x4*)ctx->d_sph_sfrag);
        else if (tiled) hipLaunchKernelGGL(tiled, dim3(grid), dim3(kTB), lds, stream, A, (const u32x4*)(grouped ? ctx->d_tri_gfrag : ctx->d_tri_frag),
                                           (const u32x4*)(grouped ? ctx->d_sph_gfrag : (u32x4*)ctx->d_sph_frag32));
        else hipLaunchKernelGGL(plain, dim3(grid), dim3(kBlock), lds, stream, A);
        RT3_HIP(hipGetLastError());
        RT3_HIP(hipEventRecord(b, stream));
        const dim3 ag((npix + kBlock - 1) / kBlock), ab(kBlock);
        if (var) hipLaunchKernelGGL(k_accumulate<true>, ag, ab, 0, stream, ctx->d_rad, ctx->d_accum, ctx->d_accum_sq, npix, ns, s0 == 0 ? 1 : 0);
        else hipLaunchKernelGGL(k_accumulate<false>, ag, ab, 0, stream, ctx->d_rad, ctx->d_accum, (float4*)nullptr, npix, ns, s0 == 0 ? 1 : 0);
        RT3_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_resolve, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       ctx->d_accum, npix, sample_begin + sample_count, p->flags, (uint32_t*)d_out);
    RT3_HIP(hipGetLastError());
    RT3_HIP(hipEventRecord(ctx->ev_end, stream));
    RT3_HIP(hipEventRecord(ctx->ev_acc, stream)); ctx->ev_acc_recorded = true;
    ctx->last_stream = stream;
    ctx->last_samples = (uint64_t)npix * sample_count;
    ctx->last_was_path = true;
    ctx->last_mfma16 = tiled != nullptr || (mfma_single && !single_k64);
    ctx->last_filter_rows = levels ? (uint64_t)(has_tri ? A.n_tri_top : 0u) + (has_sph ? A.n_sph_top : 0u) : tiled ? (uint64_t)A.n_tri_rows + A.n_sph_rows : mfma_single ? ctx->n_sph : 0;
    ctx->rendered = true;
    ctx->acc_valid = true; ctx->acc_params = *p; ctx->acc_cam = *cam; ctx->acc_done = sample_begin + sample_count; ctx->acc_npix = npix;
    return 0;
}

int rt3_render_path_device(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, void* d_out, void* stream) {
    if (!ctx) return RT3_E_ARG;
    if (!p) return fail(ctx, RT3_E_ARG, "params is NULL");
    return rt3_render_path_range_device(ctx, cam, p, 0, p->spp, d_out, stream);
}

int rt3_render_path_range(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, uint32_t sample_begin, uint32_t sample_count, uint32_t* out_pixels) {
    if (!ctx) return RT3_E_ARG;
    if (!out_pixels) return fail(ctx, RT3_E_ARG, "out_pixels is NULL");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    RT3_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)rt3_rows_owned(p) * p->width;
    if ((rc = ensure(ctx, &ctx->d_out, &ctx->out_entries, std::max<size_t>(npix, 1)))) return rc;
    if ((rc = rt3_render_path_range_device(ctx, cam, p, sample_begin, sample_count, ctx->d_out, ctx->stream))) return rc;
    if (npix) RT3_HIP(hipMemcpyAsync(out_pixels, ctx->d_out, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int rt3_render_path(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, uint32_t* out_pixels) {
    if (!ctx) return RT3_E_ARG;
    if (!p) return fail(ctx, RT3_E_ARG, "params is NULL");
    return rt3_render_path_range(ctx, cam, p, 0, p->spp, out_pixels);
}

// Checkpoint of the accumulation: what rt3_render_path_range keeps between calls, 4 floats per owned pixel.
int rt3_accum_download(rt3_ctx* ctx, float* sum, float* sum_sq, uint32_t* samples_done) {
    if (!ctx) return RT3_E_ARG;
    if (!ctx->acc_valid) return fail(ctx, RT3_E_STATE, "no accumulation on this context");
    RT3_HIP(hipSetDevice(ctx->device));
    if (ctx->ev_acc_recorded) RT3_HIP(hipEventSynchronize(ctx->ev_acc));        // the render may have run on a caller's stream (which may be gone by now)
    const size_t bytes = (size_t)ctx->acc_npix * sizeof(float4);
    if (sum && bytes) RT3_HIP(hipMemcpy(sum, ctx->d_accum, bytes, hipMemcpyDeviceToHost));
    if (sum_sq) {
        if (!(ctx->acc_params.flags & RT3_FLAG_VARIANCE)) return fail(ctx, RT3_E_STATE, "the accumulation was not started with RT3_FLAG_VARIANCE");
        if (bytes) RT3_HIP(hipMemcpy(sum_sq, ctx->d_accum_sq, bytes, hipMemcpyDeviceToHost));
    }
    if (samples_done) *samples_done = ctx->acc_done;
    return 0;
}

int rt3_accum_upload(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, const float* sum, const float* sum_sq, uint32_t samples_done) {
    if (!ctx) return RT3_E_ARG;
    if (!cam || !sum) return fail(ctx, RT3_E_ARG, "cam / sum_rgba is NULL");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (samples_done == 0 || samples_done > p->spp) return fail(ctx, RT3_E_ARG, "samples_done outside (0, spp]");
    const bool var = (p->flags & RT3_FLAG_VARIANCE) != 0;
    if (var && !sum_sq) return fail(ctx, RT3_E_ARG, "params ask for RT3_FLAG_VARIANCE: sum_sq_rgba is required");
    RT3_HIP(hipSetDevice(ctx->device));
    const uint32_t npix = rt3_rows_owned(p) * p->width;
    ctx->acc_valid = false;
    if (ctx->ev_acc_recorded) RT3_HIP(hipEventSynchronize(ctx->ev_acc));        // a render still in flight reads and writes what is overwritten here
    if (npix) {
        if ((rc = ensure(ctx, &ctx->d_accum, &ctx->accum_entries, (size_t)npix))) return rc;
        RT3_HIP(hipMemcpy(ctx->d_accum, sum, (size_t)npix * sizeof(float4), hipMemcpyHostToDevice));
        if (var) {
            if ((rc = ensure(ctx, &ctx->d_accum_sq, &ctx->accum_sq_entries, (size_t)npix))) return rc;
            RT3_HIP(hipMemcpy(ctx->d_accum_sq, sum_sq, (size_t)npix * sizeof(float4), hipMemcpyHostToDevice));
        }
    }
    ctx->acc_valid = true; ctx->acc_params = *p; ctx->acc_cam = *cam; ctx->acc_done = samples_done; ctx->acc_npix = npix;
    return 0;
}

// The copies of rt3_gather_rows as plain arithmetic (no device): row block lb of the shard (tile_rows rows, compact in the tile) is row block
// lb * tile_count + tile_index of the frame — one 2-D copy whose "rows" are whole row blocks, with the tile's block size as the source pitch and
// tile_count times that as the destination pitch; only the frame's very last row block can be ragged, and it then travels as one more 1-D copy.
int rt3_gather_plan(const rt3_params* p, rt3_gather_copy out[2]) {
    if (!p || !out || p->width == 0 || p->height == 0) return RT3_E_ARG;
    if (p->tile_count > 1 && (p->tile_rows == 0 || p->tile_index >= p->tile_count)) return RT3_E_ARG;
    const uint64_t row_bytes = (uint64_t)p->width * 4;
    if (p->tile_count <= 1) {
        out[0] = rt3_gather_copy{ 0, 0, row_bytes * p->height, row_bytes * p->height, row_bytes * p->height, 1 };
        return 1;
    }
    const uint32_t n_blocks_frame = (p->height + p->tile_rows - 1) / p->tile_rows;
    if (p->tile_index >= n_blocks_frame) return 0;                                  // more shards than row blocks: this one owns nothing
    const uint32_t my_blocks = (n_blocks_frame - 1 - p->tile_index) / p->tile_count + 1;
    const uint32_t last_block = (my_blocks - 1) * p->tile_count + p->tile_index;    // frame index of this shard's last block
    const uint32_t last_rows = std::min(p->tile_rows, p->height - last_block * p->tile_rows);
    const uint32_t full = last_rows == p->tile_rows ? my_blocks : my_blocks - 1;
    const uint64_t block_bytes = row_bytes * p->tile_rows;
    int n = 0;
    if (full) out[n++] = rt3_gather_copy{ (uint64_t)p->tile_index * block_bytes, 0, block_bytes * p->tile_count, block_bytes, block_bytes, full };
    if (full != my_blocks) out[n++] = rt3_gather_copy{ (uint64_t)last_block * block_bytes, (uint64_t)full * block_bytes, row_bytes * last_rows, row_bytes * last_rows,
                                                       row_bytes * last_rows, 1 };
    return n;
}

// The gather of final pixels, device to device.  Row block lb of the shard (tile_rows rows, compact in d_tile)
// is row block lb * tile_count + tile_index of the frame: one 2-D copy with the tile's pitch on one side and tile_count times that on
// the other; only the frame's very last row block can be ragged, and it then travels as one more 1-D copy.
int rt3_gather_rows(rt3_ctx* root, void* d_frame, rt3_ctx* shard, const void* d_tile, const rt3_params* p, void* stream_) {
    if (!root || !shard) return RT3_E_ARG;
    rt3_ctx* ctx = shard;
    if (!d_frame || !d_tile) return fail(ctx, RT3_E_ARG, "rt3_gather_rows: NULL buffer");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    RT3_HIP(hipSetDevice(shard->device));
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : shard->stream;
    if (root->device != shard->device && !shard->peers_enabled.count(root->device)) {
        int can = 0;
        RT3_HIP(hipDeviceCanAccessPeer(&can, shard->device, root->device));
        if (!can) return fail(ctx, RT3_E_DEVICE, "rt3_gather_rows: no peer access from device " + std::to_string(shard->device) + " to " + std::to_string(root->device));
        const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { ctx->err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return RT3_E_DEVICE; }
        (void)hipGetLastError();
        shard->peers_enabled.insert(root->device);
    }
    rt3_gather_copy plan[2];
    const int n_copies = rt3_gather_plan(p, plan);
    if (n_copies < 0) return fail(ctx, RT3_E_ARG, "rt3_gather_rows: bad shard parameters");
    uint8_t* frame = (uint8_t*)d_frame;
    const uint8_t* tile = (const uint8_t*)d_tile;
    for (int i = 0; i < n_copies; i++) {
        const rt3_gather_copy& c = plan[i];
        if (c.rows == 1) RT3_HIP(hipMemcpyAsync(frame + c.dst_offset, tile + c.src_offset, c.row_bytes, hipMemcpyDeviceToDevice, stream));
        else RT3_HIP(hipMemcpy2DAsync(frame + c.dst_offset, c.dst_pitch, tile + c.src_offset, c.src_pitch, c.row_bytes, c.rows, hipMemcpyDeviceToDevice, stream));
    }
    return 0;
}

int rt3_get_stats(rt3_ctx* ctx, rt3_stats* out) {
    if (!ctx || !out) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof *out);
    if (!ctx->rendered) return fail(ctx, RT3_E_STATE, "no render has been issued on this context (or the last one failed)");
    RT3_HIP(hipEventSynchronize(ctx->ev_end));
    float ms = 0.0f;
    for (uint32_t i = 0; i < ctx->ev_used; i++) {
        float t = 0.0f;
        RT3_HIP(hipEventElapsedTime(&t, ctx->ev[i].first, ctx->ev[i].second));
        ms += t;
    }
    out->trace_ms = ms;
    RT3_HIP(hipEventElapsedTime(&out->total_ms, ctx->ev_begin, ctx->ev_end));
    out->launches = ctx->ev_used;
    out->samples = ctx->last_samples;
    out->n_spheres = ctx->n_sph;
    out->n_faces = ctx->n_faces;
    if (ctx->last_was_path) {
        unsigned long long counters[16] = { 0 };
        RT3_HIP(hipMemcpy(counters, ctx->d_casts, 128, hipMemcpyDeviceToHost));
#ifdef RT3_PROFILE_PHASES
        if (counters[13] != 0) {                                     // k_trace_mfma: where its waves spend their time
            const double all = (double)(counters[11] + counters[12] + counters[13] + counters[14] + counters[15]);
            fprintf(stderr, "[rt3 profile] k_trace_mfma32 / k_trace_mfma, share of wave time: refill %.1f %%, ray operands (+ direct spheres) %.1f %%, scan %.1f %%, "
                            "push + exact tests %.1f %%, decode + shade %.1f %%\n",
                    100.0 * counters[11] / all, 100.0 * counters[12] / all, 100.0 * counters[13] / all, 100.0 * counters[14] / all, 100.0 * counters[15] / all);
        }
#endif
#ifdef RT3_PROFILE
        if (counters[13] == 0)
        fprintf(stderr, "[rt3 profile] wave iterations %llu, flush iterations/wave-iter %.2f, candidates/ray %.2f, live lanes/wave-iter %.1f, "
                        "fresh paths/wave-iter %.1f\n", counters[4], (double)counters[2] / (double)counters[4],
                (double)counters[3] / (double)counters[0], (double)counters[0] / (double)counters[4], (double)counters[5] / (double)counters[4]);
        if (counters[13] != 0) {                                     // tiled kernel: where its waves spend their time
            const double all = (double)(counters[12] + counters[13] + counters[14] + counters[15] + counters[4]);
            fprintf(stderr, "[rt3 profile] tiled kernel, share of wave time: barriers + tile fill %.1f %%, scan %.1f %%, push %.1f %%, exact tests %.1f %%, "
                            "refill / operands / shade %.1f %%\n", 100.0 * counters[12] / all, 100.0 * counters[13] / all, 100.0 * counters[14] / all,
                    100.0 * counters[15] / all, 100.0 * counters[4] / all);
        } else {
        fprintf(stderr, "[rt3 profile] first wave ended %.1f us before the last one\n", (double)(counters[7] - counters[6]) / 100.0);
        fprintf(stderr, "[rt3 profile] timeline from the first wave's start (us): last wave start %.1f, queue first seen empty %.1f, last seen empty %.1f, "
                        "first wave end %.1f, last wave end %.1f\n", (double)(counters[11] - counters[8]) / 100.0, (double)(counters[9] - counters[8]) / 100.0,
                (double)(counters[10] - counters[8]) / 100.0, (double)(counters[6] - counters[8]) / 100.0, (double)(counters[7] - counters[8]) / 100.0);
        }
#endif
        out->ray_casts = counters[0];
        out->prim_tests = counters[0] * ((uint64_t)ctx->n_sph + ctx->n_faces);
        out->mfma_instructions = counters[1];
        out->mfma_flop_per_instruction = counters[1] ? (ctx->last_mfma16 ? 16384u : 32768u) : 0u;
#ifndef RT3_PROFILE
        out->exact_tests = counters[2];
        out->bound_tests = counters[3];
#endif
        out->filter_tests = counters[0] * ctx->last_filter_rows;
    } else {
        out->ray_casts = ctx->last_samples;
        out->prim_tests = ctx->last_samples * (uint64_t)ctx->n_faces;
    }
    return 0;
}

// Debug probe (tests only): element-wise device arithmetic, see tests/test_gpu_arith.py.
int rt3_debug_arith(rt3_ctx* ctx, const float* a, const float* b, uint32_t n, float* div, float* sq, float* fm,
                    float* cs, float* sn, float* sk3, uint32_t* pk) {
    if (!ctx) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    float* d = nullptr;
    const size_t N = n;
    RT3_HIP(hipMalloc((void**)&d, N * 4 * 11));
    struct Free { float* p; ~Free() { (void)hipFree(p); } } free_on_return{ d };   // every early return below releases it
    float *da = d, *db = d + N, *ddiv = d + 2 * N, *dsq = d + 3 * N, *dfm = d + 4 * N, *dcs = d + 5 * N, *dsn = d + 6 * N, *dsk = d + 7 * N;
    uint32_t* dpk = (uint32_t*)(d + 10 * N);
    RT3_HIP(hipMemcpy(da, a, N * 4, hipMemcpyHostToDevice));
    RT3_HIP(hipMemcpy(db, b, N * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_arith, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, da, db, n, ddiv, dsq, dfm, dcs, dsn, dsk, dpk);
    RT3_HIP(hipGetLastError());
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    RT3_HIP(hipMemcpy(div, ddiv, N * 4, hipMemcpyDeviceToHost));
    RT3_HIP(hipMemcpy(sq, dsq, N * 4, hipMemcpyDeviceToHost));
    RT3_HIP(hipMemcpy(fm, dfm, N * 4, hipMemcpyDeviceToHost));
    RT3_HIP(hipMemcpy(cs, dcs, N * 4, hipMemcpyDeviceToHost));
    RT3_HIP(hipMemcpy(sn, dsn, N * 4, hipMemcpyDeviceToHost));
    RT3_HIP(hipMemcpy(sk3, dsk, N * 12, hipMemcpyDeviceToHost));
    RT3_HIP(hipMemcpy(pk, dpk, N * 4, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
