// rt3_kernel_common.hpp — constants, arithmetic helpers (hash RNG, sky, pixel packing, sin/cos), Mode-X launch arguments and start_path()
// Part of rt3_device.hip (one translation unit, gfx950 only); included from there, in this order.
#pragma once

namespace {

constexpr int      kBlock       = 256;     // 4 wavefronts of 64
constexpr int      kCandSlots   = 16;      // deferred sphere candidates per lane (LDS), flushed when full
constexpr uint32_t kWorkChunk   = 256;     // samples a wave takes from the global queue per atomic
constexpr uint32_t kSphLdsMax   = 2048;    // spheres mirrored in LDS for the exact-evaluation gathers (32 KiB)
constexpr uint32_t kModeRTile   = 512;     // faces per LDS tile in k_mode_r (32 KiB)

// filler for the tail of a sphere tile: r^2 = -1e30 makes the discriminant negative for every ray
#define kPadSphere make_float4(0.0f, 0.0f, 0.0f, -1e30f)

struct CamDev { float ox, oy, oz, hx, hy, hz, vx, vy, vz, lx, ly, lz; };

// ------------------------------------------------------------------------------------------------------
// Small device helpers
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// dot3 of SequentialRenderer.cpp:32-33 / glm::dot: unfused, left to right.
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + ay * by + az * bz;
}
// Mode-X dot: z*z' + (y*y' + x*x') as two fused multiply-adds.
__device__ __forceinline__ float dotf(float ax, float ay, float az, float bx, float by, float bz) {
    return fma_(az, bz, fma_(ay, by, ax * bx));
}
__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of a 64-bit lane mask below the calling lane (exclusive prefix count)
__device__ __forceinline__ uint32_t prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// n / d for a divisor fixed per launch: multiply-high by a precomputed magic (branch-free round-up method of
// Granlund & Montgomery as used by libdivide); exact for every 32-bit n.  d == 1 is encoded as shift == 0xFFFFFFFF.
struct FastDiv { uint32_t magic, shift; };
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
    if (f.shift == 0xFFFFFFFFu) return n;                           // wave-uniform
    const uint32_t q = __umulhi(f.magic, n);
    return (((n - q) >> 1) + q) >> f.shift;
}

// random_v1.glsl:22-31, :37-52
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x += x << 10; x ^= x >> 6; x += x << 3; x ^= x >> 11; x += x << 15;
    return x;
}
__device__ __forceinline__ uint32_t hash2(uint32_t a, uint32_t b) { return hash_u32(a ^ hash_u32(b)); }
__device__ __forceinline__ float u01(uint32_t m) { return __uint_as_float((m & 0x007FFFFFu) | 0x3F800000u) - 1.0f; }
__device__ __forceinline__ float rnd(uint32_t base, uint32_t ctr) { return u01(hash2(base, ctr)); }

// sky gradient, SequentialRenderer.cpp:105-107 (float form of raytracer_v3.glsl:139-141; same bits)
__device__ __forceinline__ void sky(float dx, float dy, float dz, float& r, float& g, float& b) {
    const float len = __builtin_sqrtf(dot3(dx, dy, dz, dx, dy, dz));
    const float uy = dy / len;
    const float t = 0.5f * (uy + 1.0f);
    const float a = 1.0f - t;
    r = a * 1.0f + t * 0.5f;
    g = a * 1.0f + t * 0.7f;
    b = a * 1.0f + t * 1.0f;
}
// glm::packUnorm4x8(vec4(1, b, g, r)), glm/detail/func_packing.inl:67-83
__device__ __forceinline__ uint32_t pack_channel(float c) {
    float m = c < 0.0f ? 0.0f : c;
    m = 1.0f < m ? 1.0f : m;
    return (uint32_t)(__builtin_roundf(m * 255.0f)) & 0xFFu;
}
__device__ __forceinline__ uint32_t pack_pixel(float r, float g, float b) {
    return 0xFFu | (pack_channel(b) << 8) | (pack_channel(g) << 16) | (pack_channel(r) << 24);
}

// (cos, sin)(2*pi*u), u in [0,1): quadrant + Taylor/Horner in fma
__device__ __forceinline__ void sincos2pi(float u, float& c_out, float& s_out) {
    const float a = u * 4.0f;
    const int k = (int)a;
    const float f = a - (float)k;
    const float x = f * 1.57079637f;
    const float x2 = x * x;
    float p = fma_(x2, -2.50521084e-8f, 2.75573192e-6f);
    p = fma_(x2, p, -1.98412698e-4f);
    p = fma_(x2, p, 8.33333333e-3f);
    p = fma_(x2, p, -1.66666667e-1f);
    const float s = fma_(x * x2, p, x);
    float q = fma_(x2, 2.08767570e-9f, -2.75573192e-7f);
    q = fma_(x2, q, 2.48015873e-5f);
    q = fma_(x2, q, -1.38888889e-3f);
    q = fma_(x2, q, 4.16666667e-2f);
    q = fma_(x2, q, -0.5f);
    const float c = fma_(x2, q, 1.0f);
    const int kk = k & 3;
    c_out = kk == 0 ? c : kk == 1 ? -s : kk == 2 ? -c : s;
    s_out = kk == 0 ? s : kk == 1 ? c : kk == 2 ? -s : -c;
}
__device__ __forceinline__ void unit_vector(float xi0, float xi1, float& x, float& y, float& z) {
    z = fma_(-2.0f, xi0, 1.0f);
    const float rr = fma_(-z, z, 1.0f);
    const float r = __builtin_sqrtf(rr > 0.0f ? rr : 0.0f);
    float c, s;
    sincos2pi(xi1, c, s);
    x = r * c;
    y = r * s;
}

// ------------------------------------------------------------------------------------------------------
// Mode X
// ------------------------------------------------------------------------------------------------------
struct Rgb { float r, g, b; };   // one radiance record of the sample storage (three dwords: a quarter less HBM traffic than float4)
struct TraceArgs {
    const float4* sph;       const float* sph_invr;  const float4* sph_mat;  const uint32_t* sph_kind;  uint32_t n_sph;
    const float4* tri;       const float4* tri_mat;  const uint32_t* tri_kind;  const float4* tri_bound; uint32_t n_tri;
    CamDev cam;
    float lens_radius, lux, luy, luz, lvx, lvy, lvz;
    uint32_t width, height, spp, max_depth, seed, flags, edge;
    FastDiv div_npix, div_width, div_edge, div_tile_rows;
    float t_min;
    uint32_t tile_rows, tile_index, tile_count;
    uint32_t npix;           // pixels owned by this shard
    uint32_t s0;             // first sample of this batch
    uint32_t total;          // npix * samples in this batch
    Rgb* rad;                // per-sample radiance, [sample in batch][owned pixel], 12 B each
    // Centres the matrix filter's coordinates are taken about: the filter's margin is eps (|C|^2 + r^2 + |o|^2), so a scene far from the
    // world origin would otherwise drown in candidates.  Spheres: the median of their centres; faces: the centre of the vertices' box.
    float fcx, fcy, fcz;     // spheres (k_trace_mfma, the K = 32 pass of k_trace_mfma_tiled)
    float tcx, tcy, tcz;     // faces (the faces' pass of k_trace_mfma_tiled)
    // Spheres that nearly every ray is a candidate for (a ground sphere; a sphere around the middle of the scene) gain nothing from the
    // filter and cost the pair list 64 entries per ray cast each: they are tested directly, every lane its own ray, and their fragment
    // rows can never be candidates (sphere_direct_list in rt3_device.hip).  Matrix-filter kernels only.
    uint32_t n_direct; uint32_t direct[4];
    // Two-level filter of k_trace_mfma_tiled: with a group size G > 1 a row of the matrix filter is the bounding sphere of G
    // primitives, grouped in the order of a spatial median split (rt3_set_spheres / rt3_mesh_commit): *_grp holds the members' (cx, cy, cz, r^2)
    // records — the spheres themselves, the faces' bounding spheres — in GROUP order (row g = entries [g G, (g + 1) G)), *_perm the primitive
    // index of each (0xFFFFFFFF: padding; the direct spheres are in no group).  *_rows: rows the pass scans (= primitives when G is 1).
    const float4* sph_grp; const uint32_t* sph_perm; const float4* tri_grp; const uint32_t* tri_perm; uint32_t n_sph_rows, n_tri_rows;
    // Three levels (kSuper > 1): a row bounds kSuper consecutive LEAF groups; *_leaf holds the leaves' bounding spheres (cx, cy, cz, R_eff^2), which a
    // candidate row's rays are tested against in f32 before the leaves' members are.  n_*_leaves: leaf groups (= rows x kSuper).
    const float4* sph_leaf; const float4* tri_leaf; uint32_t n_sph_leaves, n_tri_leaves;
    const float4* sph_rowb; const float4* tri_rowb;                 // three-level filter: the rows' own bounds (C, R_eff^2) in f32, or null
    const float4* tri_rec;                                          // the faces' 64-byte records once more, in GROUP order (a leaf's 8 faces: 512 contiguous bytes)
    const float4* sph_topb; const float4* tri_topb; uint32_t n_sph_top, n_tri_top;   // k_trace_levels: what the matrix cores scan (rows, or super-rows of 8 rows) and its bounds in f32
    uint32_t* pair_strips;   // [wave of the grid][kStripPairs]: candidate (ray lane, row) pairs set aside for the end of a pass (deferred member tests)
    uint32_t* work_counter;
    unsigned long long* cast_counter;
};

struct Path {
    float ox, oy, oz, dx, dy, dz;
    float tr, tg, tb, lr, lg, lb;
    uint32_t slot, base, depth;
};

// The reference's plane + three-edge test of one face (SequentialRenderer.cpp:53-98; hit_vertex, raytracer_v4.glsl:116-153), in
// its own operation order.  n = (normal, n.p1) as k_commit_mesh stores it.
//   face_t       the ray parameter of the plane: (n.p1 - n.o) / n.d — Mode X, sign of n.o corrected — or, `literal`, the
//                reference's own (n.o + n.p1) / n.d (:70, sic), which only means the plane for an origin of 0; false if n.d == 0 (:56)
//   face_inside  the three edge functions at the point o + t d (:77-98)
__device__ __forceinline__ bool face_t(const float4 n, float ox, float oy, float oz, float dx, float dy, float dz, bool literal, float& t) {
    const float nd = dot3(dx, dy, dz, n.x, n.y, n.z);
    if (nd == 0.0f) return false;
    const float no = dot3(n.x, n.y, n.z, ox, oy, oz);
    t = (literal ? no + n.w : n.w - no) / nd;
    return true;
}
__device__ __forceinline__ bool face_inside(const float4 n, const float4 p1, const float4 p2, const float4 p3, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float t) {
    const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;
    float ex, ey, ez, qx, qy, qz, cx, cy, cz;
    ex = p2.x - p1.x; ey = p2.y - p1.y; ez = p2.z - p1.z; qx = hx - p1.x; qy = hy - p1.y; qz = hz - p1.z;
    cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
    if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return false;
    ex = p3.x - p2.x; ey = p3.y - p2.y; ez = p3.z - p2.z; qx = hx - p2.x; qy = hy - p2.y; qz = hz - p2.z;
    cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
    if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return false;
    ex = p1.x - p3.x; ey = p1.y - p3.y; ez = p1.z - p3.z; qx = hx - p3.x; qy = hy - p3.y; qz = hz - p3.z;
    cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
    return -dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f;
}
// Both, with the vertices fetched only when t lies in [t_lo, t_hi) (STRICT) or [t_lo, t_hi].
template <bool STRICT>
__device__ __forceinline__ bool face_hit(const float4 n, const float4* __restrict__ f, float ox, float oy, float oz, float dx, float dy, float dz,
                                         float t_lo, float t_hi, bool literal, float& t_out) {
    float t;
    if (!face_t(n, ox, oy, oz, dx, dy, dz, literal, t)) return false;
    if (!(t >= t_lo && (STRICT ? t < t_hi : t <= t_hi))) return false;
    if (!face_inside(n, f[1], f[2], f[3], ox, oy, oz, dx, dy, dz, t)) return false;
    t_out = t;
    return true;
}
// Exact ray-sphere test of Mode X (hit_sphere, raytracer_v4.glsl:157-178, unit direction, the book's far-root rule):
// s = (centre, r^2); true with the accepted root in t when t_min < t.
__device__ __forceinline__ bool sphere_root(const float4 s, float ox, float oy, float oz, float dx, float dy, float dz, float t_min, float& t_out) {
    const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
    const float h = fma_(cz, dz, fma_(cy, dy, cx * dx));
    const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -s.w)));
    const float disc = fma_(h, h, -c);
    if (!((c < 0.0f) | ((disc > 0.0f) & (h > 0.0f)))) return false;
    const float sq = __builtin_sqrtf(disc);
    float t = h - sq;
    if (!(t > t_min)) t = h + sq;
    t_out = t;
    return t > t_min;
}

__device__ __forceinline__ uint32_t frame_row(const TraceArgs& A, uint32_t local_row) {
    if (A.tile_count <= 1) return local_row;
    const uint32_t lb = fdiv(local_row, A.div_tile_rows), in = local_row - lb * A.tile_rows;
    return (lb * A.tile_count + A.tile_index) * A.tile_rows + in;
}

// sample -> primary ray (raytracer_v4.glsl:190-214 with the jitter in pixel units), unit direction.
// REF (RT3_FLAG_REFERENCE_PRIMARY): the direction stays unnormalised, as SequentialRenderer.cpp:293 leaves it.
template <bool REF = false>
__device__ __forceinline__ void start_path(const TraceArgs& A, uint32_t item, Path& P) {
    const uint32_t sb = fdiv(item, A.div_npix), pix = item - sb * A.npix;
    const uint32_t s = A.s0 + sb;
    const uint32_t lrow = fdiv(pix, A.div_width), x = pix - lrow * A.width;
    const uint32_t y = frame_row(A, lrow);
    const uint32_t base = hash2(y * A.width + x, hash2(s, A.seed));
    float jx = 0.0f, jy = 0.0f;
    if (A.spp > 1) {
        const float xi0 = rnd(base, 1), xi1 = rnd(base, 2);
        if (A.edge != 0) {
            const uint32_t sy = fdiv(s, A.div_edge), sx = s - sy * A.edge;
            jx = ((float)sx + xi0) / (float)A.edge - 0.5f;
            jy = ((float)sy + xi1) / (float)A.edge - 0.5f;
        } else { jx = xi0 - 0.5f; jy = xi1 - 0.5f; }
    }
    const float u = ((float)x + jx) / ((float)A.width - 1.0f);
    const float v = ((float)(A.height - 1 - y) + jy) / ((float)A.height - 1.0f);
    const CamDev& c = A.cam;
    float rx = ((c.lx + u * c.hx) + v * c.vx) - c.ox;
    float ry = ((c.ly + u * c.hy) + v * c.vy) - c.oy;
    float rz = ((c.lz + u * c.hz) + v * c.vz) - c.oz;
    float ox = c.ox, oy = c.oy, oz = c.oz;
    if (A.lens_radius > 0.0f) {
        const float xi2 = rnd(base, 3), xi3 = rnd(base, 4);
        const float r = A.lens_radius * __builtin_sqrtf(xi2);
        float cs, sn;
        sincos2pi(xi3, cs, sn);
        const float a = r * cs, b = r * sn;
        const float fx = a * A.lux + b * A.lvx, fy = a * A.luy + b * A.lvy, fz = a * A.luz + b * A.lvz;
        ox = ox + fx; oy = oy + fy; oz = oz + fz;
        rx = rx - fx; ry = ry - fy; rz = rz - fz;
    }
    P.ox = ox; P.oy = oy; P.oz = oz;
    if (REF) { P.dx = rx; P.dy = ry; P.dz = rz; }
    else {
        const float inv = 1.0f / __builtin_sqrtf(dot3(rx, ry, rz, rx, ry, rz));
        P.dx = rx * inv; P.dy = ry * inv; P.dz = rz * inv;
    }
    P.tr = P.tg = P.tb = 1.0f;
    P.lr = P.lg = P.lb = 0.0f;
    P.slot = item; P.base = base; P.depth = 0;
}

}  // namespace
