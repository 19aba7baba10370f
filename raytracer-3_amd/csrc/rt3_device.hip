// rt3_device.hip — the render path on gfx950 (CDNA4): HIP kernels + the device half of the C ABI.
//
//   k_mode_r      Mode R: SequentialRenderer::render + ray_color (src/lib/renderer/SequentialRenderer.cpp:47-109,
//                 269-308; GLSL twin src/lib/shaders/raytracer_v3.glsl:91-143,187-203).  One thread per pixel,
//                 de-indexed faces staged through LDS in index order, IEEE arithmetic in the reference's order.
//   k_trace       Mode X: the recursion of the (unfinished) raytracer_v4.glsl:187-290 flattened into an iterative
//                 per-wavefront loop.  Every lane carries one path; each iteration traces all 64 paths of the
//                 wave against LDS-staged primitive tiles, shades, and refills the lanes whose path ended with
//                 fresh samples handed out by ballot + prefix-count (mbcnt), so waves stay full between bounces
//                 without ever spilling ray state to HBM.  Finished samples land in a per-sample storage buffer
//                 (SampleStorage of raytracer_v4.glsl:107-111) and are averaged in sample order by k_accumulate /
//                 k_resolve (the reduce pass reduce_v1.glsl never got) — the image is bitwise independent of
//                 scheduling, block size and GPU count.
//
// Compiled with -ffp-contract=off: a*b+c is two roundings unless written __builtin_fmaf.  Division and sqrt are
// the correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), f32 denormals are kept.
// No MFMA: the path is branchy scalar FP32; the bounding roofline is the FP32 vector ALU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rt3.h"

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
// Mode R
// ------------------------------------------------------------------------------------------------------
// tri: 4 float4 per face — (n.xyz, n.p1), p1, p2, p3 — i.e. the reference's unused de-indexed Face struct
// (src/lib/renderer/Vertex.hpp:24-36) with the plane distance of SequentialRenderer.cpp:67 precomputed.
__global__ __launch_bounds__(kBlock) void k_mode_r(const float4* __restrict__ tri, const float4* __restrict__ face_rgb,
                                                  uint32_t n_faces, CamDev cam, uint32_t width, uint32_t height,
                                                  uint32_t* __restrict__ out) {
    __shared__ float4 tile[kModeRTile * 4];
    const uint32_t pixel = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = pixel < width * height;
    const uint32_t x = valid ? pixel % width : 0u, y = valid ? pixel / width : 0u;

    // SequentialRenderer.cpp:289-293 (u, v evaluated in double exactly as the C++ does, then rounded)
    const float u = (float)((double)(float)x / ((double)(float)width - 1.0));
    const float v = (float)((double)(float)(height - 1 - y) / ((double)(float)height - 1.0));
    const float ox = cam.ox, oy = cam.oy, oz = cam.oz;
    const float dx = ((cam.lx + u * cam.hx) + v * cam.vx) - ox;
    const float dy = ((cam.ly + u * cam.hy) + v * cam.vy) - oy;
    const float dz = ((cam.lz + u * cam.hz) + v * cam.vz) - oz;

    uint32_t min_i = 0;
    float min_t = __builtin_inff();                                 // (float)1e99, :52
    for (uint32_t t0 = 0; t0 < n_faces; t0 += kModeRTile) {
        const uint32_t cnt = min(kModeRTile, n_faces - t0);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt * 4; k += kBlock) tile[k] = tri[(size_t)t0 * 4 + k];
        __syncthreads();
        if (!valid) continue;
        for (uint32_t j = 0; j < cnt; j++) {                        // ascending face index == reference order
            const float4 n = tile[4 * j];
            const float nd = dot3(dx, dy, dz, n.x, n.y, n.z);       // :56
            if (nd == 0.0f) continue;
            const float t = (dot3(n.x, n.y, n.z, ox, oy, oz) + n.w) / nd;      // :70 (sic: plus)
            if (t < 0.0f || t >= min_t) continue;                   // :71
            const float4 p1 = tile[4 * j + 1], p2 = tile[4 * j + 2], p3 = tile[4 * j + 3];
            const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;  // :77
            float ex, ey, ez, qx, qy, qz, cx, cy, cz;
            ex = p2.x - p1.x; ey = p2.y - p1.y; ez = p2.z - p1.z; qx = hx - p1.x; qy = hy - p1.y; qz = hz - p1.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) continue;
            ex = p3.x - p2.x; ey = p3.y - p2.y; ez = p3.z - p2.z; qx = hx - p2.x; qy = hy - p2.y; qz = hz - p2.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) continue;
            ex = p1.x - p3.x; ey = p1.y - p3.y; ez = p1.z - p3.z; qx = hx - p3.x; qy = hy - p3.y; qz = hz - p3.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) continue;
            min_i = t0 + j;
            min_t = t;
        }
    }
    if (!valid) return;
    float r, g, b;
    if (min_t < __builtin_inff()) { const float4 c = face_rgb[min_i]; r = c.x; g = c.y; b = c.z; }
    else sky(dx, dy, dz, r, g, b);
    out[pixel] = pack_pixel(r, g, b);                               // :297
}

// ------------------------------------------------------------------------------------------------------
// Mode X
// ------------------------------------------------------------------------------------------------------
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
    float4* rad;             // per-sample radiance, [sample in batch][owned pixel]
    uint32_t* work_counter;
    unsigned long long* cast_counter;
};

struct Path {
    float ox, oy, oz, dx, dy, dz;
    float tr, tg, tb, lr, lg, lb;
    uint32_t slot, base, depth;
};

__device__ __forceinline__ uint32_t frame_row(const TraceArgs& A, uint32_t local_row) {
    if (A.tile_count <= 1) return local_row;
    const uint32_t lb = fdiv(local_row, A.div_tile_rows), in = local_row - lb * A.tile_rows;
    return (lb * A.tile_count + A.tile_index) * A.tile_rows + in;
}

// sample -> primary ray (raytracer_v4.glsl:190-214 with the jitter in pixel units), unit direction
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
    const float inv = 1.0f / __builtin_sqrtf(dot3(rx, ry, rz, rx, ry, rz));
    P.ox = ox; P.oy = oy; P.oz = oz;
    P.dx = rx * inv; P.dy = ry * inv; P.dz = rz * inv;
    P.tr = P.tg = P.tb = 1.0f;
    P.lr = P.lg = P.lb = 0.0f;
    P.slot = item; P.base = base; P.depth = 0;
}

// One LDS tile of bounding spheres (cx, cy, cz, r^2) against the ray of every lane — the hot loop of k_trace.
// Per sphere: one broadcast ds_read_b128 + 10 FMA-class VALU ops giving the discriminant of the line-sphere quadratic,
// and one v_alignbit_b32 that shifts its sign bit into a per-lane mask (32 spheres per mask; no branches, loads
// batched).  MARGIN adds one fma that biases the discriminant by 1e-5*c, for spheres that only BOUND a primitive:
// the test must never lose a true hit to rounding.  Lanes then push their candidate indices into a
// per-lane LDS queue and `eval(j)` runs on every queued index, all lanes together, in ascending index order.
// read-only scene data is addressed through the constant address space: wave-uniform loads from it become scalar
// (s_load_dwordx16 = 4 spheres per instruction, served by the scalar cache) and their results are SGPR operands.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f32x4* scene_ptr;
__device__ __forceinline__ scene_ptr as_scene(const float4* p) { return (scene_ptr)p; }

template <bool MARGIN, bool PREFETCH, class Fetch, class Eval>
__device__ __forceinline__ void scan_tile(scene_ptr tile, uint32_t cnt, uint32_t* __restrict__ cand, uint32_t tid,
                                          float ox, float oy, float oz, float dx, float dy, float dz, Fetch&& fetch, Eval&& eval) {
    uint32_t ncand = 0;
    // `fetch(j)` loads the first 16 bytes the exact test of primitive j needs.  With PREFETCH it is issued one candidate
    // ahead of `eval(j, record)`, so that a gather from global memory overlaps the previous candidate's arithmetic
    // (faces: -6 % on the 47k-face scene; for spheres the extra bookkeeping costs more than it hides: +2..3 %).
    auto flush = [&]() {
        if (!PREFETCH) {
            for (uint32_t q = 0; q < ncand; q++) { const uint32_t j = cand[q * kBlock + tid]; eval(j, fetch(j)); }
        } else if (ncand != 0) {
            uint32_t j = cand[tid];
            float4 rec = fetch(j);
            for (uint32_t q = 0; q < ncand; q++) {
                const uint32_t jn = q + 1 < ncand ? cand[(q + 1) * kBlock + tid] : j;
                const float4 recn = q + 1 < ncand ? fetch(jn) : rec;
                eval(j, rec);
                j = jn; rec = recn;
            }
        }
        ncand = 0;
    };
    auto test = [&](const f32x4 s, uint32_t neg) {
        const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
        const float h = fma_(cz, dz, fma_(cy, dy, cx * dx));
        const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -s.w)));
        float disc = fma_(h, h, -c);
        if (MARGIN) disc = fma_(1e-5f, c, disc);
        return __builtin_amdgcn_alignbit(neg, __float_as_uint(disc), 31);
    };
    // N consecutive spheres: the sign bits end up in the low N bits of `neg` (bit N-1-k <-> sphere b0+k), the bits
    // above stay set; candidates are pushed in ascending index.  Groups of 4 spheres (one 64-byte scalar load) are
    // fetched one group ahead of the arithmetic.
    auto block = [&](uint32_t b0, auto n_tag) {
        constexpr uint32_t N = decltype(n_tag)::value;
        uint32_t neg = 0xFFFFFFFFu;
        f32x4 g0 = tile[b0], g1 = tile[b0 + 1], g2 = tile[b0 + 2], g3 = tile[b0 + 3];
#pragma unroll
        for (uint32_t k = 0; k < N; k += 4) {
            f32x4 n0 = g0, n1 = g1, n2 = g2, n3 = g3;
            if (k + 4 < N) { n0 = tile[b0 + k + 4]; n1 = tile[b0 + k + 5]; n2 = tile[b0 + k + 6]; n3 = tile[b0 + k + 7]; }
            neg = test(g0, neg); neg = test(g1, neg); neg = test(g2, neg); neg = test(g3, neg);
            g0 = n0; g1 = n1; g2 = n2; g3 = n3;
        }
        uint32_t cm = ~neg;
        while (cm != 0) {
            const uint32_t top = 31u - (uint32_t)__builtin_clz(cm);
            cm &= ~(1u << top);
            if (ncand == kCandSlots) flush();
            cand[ncand * kBlock + tid] = b0 + (N - 1u - top);
            ncand++;
        }
    };
    uint32_t b0 = 0;                                                // arrays are padded to a multiple of 4 with never-hit spheres
    for (; b0 + 32 <= cnt; b0 += 32) block(b0, std::integral_constant<uint32_t, 32>());
    for (; b0 < cnt; b0 += 4) block(b0, std::integral_constant<uint32_t, 4>());
    flush();
}

// Mode R with the camera at the origin (the only camera Camera::update can build, Camera.cpp:89): n.o == 0, so the
// reference's hit point lies on the ray's line and the bounding-sphere scan is a valid conservative filter.  The
// reference's own test (same code as k_mode_r) runs on the surviving faces, in ascending face index.
__global__ __launch_bounds__(kBlock) void k_mode_r_fast(const float4* __restrict__ tri, const float4* __restrict__ tri_bound,
                                                       const float4* __restrict__ face_rgb, uint32_t n_faces, CamDev cam,
                                                       uint32_t width, uint32_t height, uint32_t* __restrict__ out) {
    __shared__ uint32_t cand[kCandSlots * kBlock];
    const uint32_t tid = threadIdx.x;
    const uint32_t pixel = blockIdx.x * kBlock + tid;
    const bool valid = pixel < width * height;
    const uint32_t x = valid ? pixel % width : 0u, y = valid ? pixel / width : 0u;
    const float u = (float)((double)(float)x / ((double)(float)width - 1.0));
    const float v = (float)((double)(float)(height - 1 - y) / ((double)(float)height - 1.0));
    const float ox = cam.ox, oy = cam.oy, oz = cam.oz;              // all zero (checked by the host)
    const float dx = ((cam.lx + u * cam.hx) + v * cam.vx) - ox;
    const float dy = ((cam.ly + u * cam.hy) + v * cam.vy) - oy;
    const float dz = ((cam.lz + u * cam.hz) + v * cam.vz) - oz;
    const float inv = 1.0f / __builtin_sqrtf(dot3(dx, dy, dz, dx, dy, dz));     // unit direction for the filter only
    const float ux = dx * inv, uy = dy * inv, uz = dz * inv;

    uint32_t min_i = 0;
    float min_t = __builtin_inff();
    if (valid) {
        const uint32_t t0 = 0;
        scan_tile<true, true>(as_scene(tri_bound), n_faces, cand, tid, ox, oy, oz, ux, uy, uz,
                        [&](uint32_t j) { return tri[(size_t)(t0 + j) * 4]; }, [&](uint32_t j, const float4 n) {
            const float4* f = tri + (size_t)(t0 + j) * 4;
            const float nd = dot3(dx, dy, dz, n.x, n.y, n.z);       // SequentialRenderer.cpp:56
            if (nd == 0.0f) return;
            const float t = (dot3(n.x, n.y, n.z, ox, oy, oz) + n.w) / nd;      // :70
            if (t < 0.0f || t >= min_t) return;                     // :71
            const float4 p1 = f[1], p2 = f[2], p3 = f[3];
            const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;
            float ex, ey, ez, qx, qy, qz, cx, cy, cz;
            ex = p2.x - p1.x; ey = p2.y - p1.y; ez = p2.z - p1.z; qx = hx - p1.x; qy = hy - p1.y; qz = hz - p1.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
            ex = p3.x - p2.x; ey = p3.y - p2.y; ez = p3.z - p2.z; qx = hx - p2.x; qy = hy - p2.y; qz = hz - p2.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
            ex = p1.x - p3.x; ey = p1.y - p3.y; ez = p1.z - p3.z; qx = hx - p3.x; qy = hy - p3.y; qz = hz - p3.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
            min_i = t0 + j;
            min_t = t;
        });
    }
    if (!valid) return;
    float r, g, b;
    if (min_t < __builtin_inff()) { const float4 c = face_rgb[min_i]; r = c.x; g = c.y; b = c.z; }
    else sky(dx, dy, dz, r, g, b);
    out[pixel] = pack_pixel(r, g, b);
}

// HAS_TRI / HAS_SPH compile the face loop / sphere loop (and the matching shading) in or out, so that a sphere-only
// scene does not pay registers or code for the triangle path.
// Refill: lanes whose path ended take the next samples of the wave's chunk (ballot + prefix count); a chunk of
// kWorkChunk samples is fetched from the global queue with one atomic when the wave runs dry.
__device__ __forceinline__ void refill_lanes(const TraceArgs& A, uint32_t lane, bool& alive, Path& P, uint32_t& chunk_next,
                                             uint32_t& chunk_end, bool& exhausted) {
    const unsigned long long need = __ballot(!alive);
    if (need != 0ull && !exhausted) {
        const uint32_t n_need = (uint32_t)__popcll(need);
        const uint32_t rank = prefix_count(need);
        uint32_t item = 0xFFFFFFFFu, taken = 0;
        while (taken < n_need) {
            if (chunk_next == chunk_end) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(A.work_counter, kWorkChunk);
                b = __builtin_amdgcn_readfirstlane(b);
                if (b >= A.total) { exhausted = true; break; }
                chunk_next = b;
                chunk_end = min(b + kWorkChunk, A.total);
            }
            const uint32_t k = min(n_need - taken, chunk_end - chunk_next);
            if (!alive && rank >= taken && rank < taken + k) item = chunk_next + (rank - taken);
            chunk_next += k;
            taken += k;
        }
        if (item != 0xFFFFFFFFu) { start_path(A, item, P); alive = true; }
    }
}

// Refill through a wave-wide stock of primary rays: start_path() runs for all 64 lanes at once (lane k of the stock holds sample
// chunk_next + k) and lanes whose path ended pop entries off the top with ds_bpermute — the ~250 instructions of start_path are
// then paid per 64 new paths instead of per loop iteration (in which about a third of the lanes end).
struct RayStock { float ox, oy, oz, dx, dy, dz; uint32_t slot, base; uint32_t n; };   // n: wave-uniform count, entries in lanes [0, n)
__device__ __forceinline__ void stock_pop(const RayStock& Q, uint32_t src, bool take, Path& P, bool& alive) {
    const int s = (int)src;
    const float ox = __shfl(Q.ox, s), oy = __shfl(Q.oy, s), oz = __shfl(Q.oz, s), dx = __shfl(Q.dx, s), dy = __shfl(Q.dy, s), dz = __shfl(Q.dz, s);
    const uint32_t slot = (uint32_t)__shfl((int)Q.slot, s), base = (uint32_t)__shfl((int)Q.base, s);
    if (take) {
        P.ox = ox; P.oy = oy; P.oz = oz; P.dx = dx; P.dy = dy; P.dz = dz; P.slot = slot; P.base = base;
        P.tr = P.tg = P.tb = 1.0f; P.lr = P.lg = P.lb = 0.0f; P.depth = 0;
        alive = true;
    }
}
__device__ __forceinline__ void refill_from_stock(const TraceArgs& A, uint32_t lane, bool& alive, Path& P, RayStock& Q, uint32_t& chunk_next,
                                                  uint32_t& chunk_end, bool& exhausted) {
    const unsigned long long need = __ballot(!alive);
    if (need == 0ull) return;
    const uint32_t n_need = (uint32_t)__popcll(need), rank = prefix_count(need);
    uint32_t served = 0;
    for (;;) {
        const uint32_t k = min(Q.n, n_need - served);
        if (k != 0) {
            const bool take = !alive && rank >= served && rank < served + k;
            stock_pop(Q, Q.n - 1u - (rank - served), take, P, alive);       // (the index only matters where take is set)
            Q.n -= k;
            served += k;
        }
        if (served == n_need || exhausted) return;
        // the stock is empty: restock from the wave's chunk (one atomic per kWorkChunk samples)
        if (chunk_next == chunk_end) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(A.work_counter, kWorkChunk);
            b = __builtin_amdgcn_readfirstlane(b);
            if (b >= A.total) { exhausted = true; return; }
            chunk_next = b;
            chunk_end = min(b + kWorkChunk, A.total);
        }
        const uint32_t n_new = min(64u, chunk_end - chunk_next);
        Path T;
        start_path(A, min(chunk_next + lane, chunk_end - 1u), T);
        Q.ox = T.ox; Q.oy = T.oy; Q.oz = T.oz; Q.dx = T.dx; Q.dy = T.dy; Q.dz = T.dz; Q.slot = T.slot; Q.base = T.base;
        Q.n = n_new;
        chunk_next += n_new;
    }
}

// Shade / scatter one ray cast of every live lane (book materials).  kind: 0 miss, 1 face, 2 sphere.
// The four per-sphere arrays read at a hit are parameters: global memory in k_trace, LDS copies in k_trace_mfma.
template <bool HAS_TRI, bool HAS_SPH>
__device__ __forceinline__ void shade_lane(const TraceArgs& A, Path& P, bool& alive, uint32_t kind, uint32_t ibest, float tbest,
                                           const float4* sph, const float* sph_invr, const float4* sph_mat, const uint32_t* sph_kind) {
    const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;
    if (alive) {
        bool done = false;
        if (kind == 0) {
            if (!(A.flags & RT3_FLAG_BLACK_BACKGROUND)) {
                float r, g, b;
                sky(dx, dy, dz, r, g, b);
                P.lr = fma_(P.tr, r, P.lr); P.lg = fma_(P.tg, g, P.lg); P.lb = fma_(P.tb, b, P.lb);
            }
            done = true;
        } else {
            float4 m; uint32_t mk;
            float px, py, pz, nx, ny, nz;
            if (HAS_TRI && (!HAS_SPH || kind == 1)) {
                m = A.tri_mat[ibest]; mk = A.tri_kind[ibest];
                const float4 n = A.tri[(size_t)ibest * 4];
                px = ox + tbest * dx; py = oy + tbest * dy; pz = oz + tbest * dz;
                nx = n.x; ny = n.y; nz = n.z;
            } else {
                m = sph_mat[ibest]; mk = sph_kind[ibest];
                const float4 s = sph[ibest];
                const float invr = sph_invr[ibest];
                px = fma_(tbest, dx, ox); py = fma_(tbest, dy, oy); pz = fma_(tbest, dz, oz);
                nx = (px - s.x) * invr; ny = (py - s.y) * invr; nz = (pz - s.z) * invr;
            }
            if (mk == RT3_MAT_FLAT) {
                P.lr = fma_(P.tr, m.x, P.lr); P.lg = fma_(P.tg, m.y, P.lg); P.lb = fma_(P.tb, m.z, P.lb);
                done = true;
            } else if (P.depth + 1 == A.max_depth) {
                done = true;
            } else {
                const bool front = dotf(dx, dy, dz, nx, ny, nz) < 0.0f;
                if (!front) { nx = -nx; ny = -ny; nz = -nz; }           // (the dot product is recomputed with the flipped normal below:
                const uint32_t ctr = 1u + 8u * (P.depth + 1u);
                float sx, sy, sz;                               // scattered direction before normalisation
                float ar = m.x, ag = m.y, ab = m.z;
                // work shared between material branches is done once for all lanes that need it: the random unit vector
                // (Lambert, fuzzy metal) and the mirror direction (metal, dielectric) — the wave executes every branch
                // that any lane takes, so merging them shortens the serialised shading
                const float dn = dotf(dx, dy, dz, nx, ny, nz);
                float vx = 0.0f, vy = 0.0f, vz = 0.0f;
                if ((mk == RT3_MAT_LAMBERT) | ((mk == RT3_MAT_METAL) & (m.w > 0.0f)))
                    unit_vector(rnd(P.base, ctr), rnd(P.base, ctr + 1), vx, vy, vz);
                const float k2 = 2.0f * dn;
                const float mx = fma_(-k2, nx, dx), my = fma_(-k2, ny, dy), mz = fma_(-k2, nz, dz);   // reflect(d, n)
                if (mk == RT3_MAT_LAMBERT) {
                    sx = nx + vx; sy = ny + vy; sz = nz + vz;
                    if (__builtin_fabsf(sx) < 1e-8f && __builtin_fabsf(sy) < 1e-8f && __builtin_fabsf(sz) < 1e-8f) { sx = nx; sy = ny; sz = nz; }
                } else if (mk == RT3_MAT_METAL) {
                    const float inv = 1.0f / __builtin_sqrtf(dotf(mx, my, mz, mx, my, mz));
                    const float rx = mx * inv, ry = my * inv, rz = mz * inv;
                    sx = rx; sy = ry; sz = rz;
                    if (m.w > 0.0f) { sx = fma_(m.w, vx, rx); sy = fma_(m.w, vy, ry); sz = fma_(m.w, vz, rz); }
                    if (!(dotf(sx, sy, sz, nx, ny, nz) > 0.0f)) done = true;       // absorbed
                } else {                                        // dielectric: m = (1/ior, r0(1/ior), r0(ior), ior), see rt3_set_spheres
                    const float ri = front ? m.x : m.w;
                    float cosv = -dn;
                    if (cosv > 1.0f) cosv = 1.0f;
                    const float s2 = fma_(-cosv, cosv, 1.0f);
                    const float sinv = __builtin_sqrtf(s2 > 0.0f ? s2 : 0.0f);
                    const bool cannot = ri * sinv > 1.0f;
                    const float r0 = front ? m.y : m.z;
                    const float xx = 1.0f - cosv, x2 = xx * xx, x5 = x2 * x2 * xx;
                    const float R = fma_(1.0f - r0, x5, r0);
                    if (cannot || R > rnd(P.base, ctr + 2)) {
                        sx = mx; sy = my; sz = mz;
                    } else {
                        const float ex = fma_(cosv, nx, dx) * ri, ey = fma_(cosv, ny, dy) * ri, ez = fma_(cosv, nz, dz) * ri;
                        const float par = -__builtin_sqrtf(__builtin_fabsf(1.0f - dotf(ex, ey, ez, ex, ey, ez)));
                        sx = fma_(par, nx, ex); sy = fma_(par, ny, ey); sz = fma_(par, nz, ez);
                    }
                    ar = ag = ab = 1.0f;
                }
                if (!done) {
                    const float inv = 1.0f / __builtin_sqrtf(dotf(sx, sy, sz, sx, sy, sz));
                    P.dx = sx * inv; P.dy = sy * inv; P.dz = sz * inv;
                    P.ox = px; P.oy = py; P.oz = pz;
                    P.tr *= ar; P.tg *= ag; P.tb *= ab;
                    P.depth += 1;
                }
            }
        }
        if (done) {
            A.rad[P.slot] = make_float4(P.lr, P.lg, P.lb, 0.0f);
            alive = false;
        }
    }
}

// SPH_LDS: the sphere array (<= kSphLdsMax entries) is also copied to LDS once per block, for the per-lane gathers of
// the exact evaluation (an LDS gather costs ~64 cycles, a global one an L2 round trip per candidate).
template <bool HAS_TRI, bool HAS_SPH, bool SPH_LDS>
__global__ __launch_bounds__(kBlock) void k_trace(const TraceArgs A) {
    __shared__ uint32_t cand[kCandSlots * kBlock];                  // per-lane candidate queues, [slot][thread]
    extern __shared__ float4 s_sph[];                               // SPH_LDS only
    const uint32_t tid = threadIdx.x, lane = lane_id();
    if (SPH_LDS) {
        for (uint32_t k = tid; k < A.n_sph; k += kBlock) s_sph[k] = A.sph[k];
        __syncthreads();                                            // the only barrier: after it the waves never meet again
    }

    Path P;
    P.ox = P.oy = P.oz = 0.0f; P.dx = P.dy = 0.0f; P.dz = 1.0f;
    P.tr = P.tg = P.tb = 0.0f; P.lr = P.lg = P.lb = 0.0f; P.slot = 0; P.base = 0; P.depth = 0;
    bool alive = false;
    uint32_t chunk_next = 0, chunk_end = 0;                         // wave-uniform
    bool exhausted = false;                                         // wave-uniform
    unsigned long long casts = 0;                                   // wave-uniform

    for (;;) {
        refill_lanes(A, lane, alive, P, chunk_next, chunk_end, exhausted);
        if (__ballot(alive) == 0ull) break;                         // waves are independent: no block-level barrier anywhere
        casts += (unsigned long long)__popcll(__ballot(alive));

        // ---- nearest hit.  kind: 0 none, 1 triangle, 2 sphere; strict '<' keeps the earlier primitive.
        float tbest = __builtin_inff();
        uint32_t ibest = 0, kind = 0;
        const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;

        // Faces: hit_vertex, raytracer_v4.glsl:116-153 (= ray_color's test, SequentialRenderer.cpp:53-98, with the sign of
        // n.o corrected).  The hot loop tests the ray against a slightly inflated bounding sphere of every face; the
        // reference's plane + three-edge test, in its own operation order, runs only for the faces that survive.
        if (HAS_TRI) {
            const uint32_t t0 = 0;
            if (alive) {
                scan_tile<true, true>(as_scene(A.tri_bound), A.n_tri, cand, tid, ox, oy, oz, dx, dy, dz,
                                [&](uint32_t j) { return A.tri[(size_t)(t0 + j) * 4]; }, [&](uint32_t j, const float4 n) {
                    const float4* f = A.tri + (size_t)(t0 + j) * 4;
                    const float nd = dot3(dx, dy, dz, n.x, n.y, n.z);
                    if (nd == 0.0f) return;
                    const float t = (n.w - dot3(n.x, n.y, n.z, ox, oy, oz)) / nd;
                    if (!(t >= A.t_min && t < tbest)) return;
                    const float4 p1 = f[1], p2 = f[2], p3 = f[3];
                    const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;
                    float ex, ey, ez, qx, qy, qz, cx, cy, cz;
                    ex = p2.x - p1.x; ey = p2.y - p1.y; ez = p2.z - p1.z; qx = hx - p1.x; qy = hy - p1.y; qz = hz - p1.z;
                    cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
                    if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
                    ex = p3.x - p2.x; ey = p3.y - p2.y; ez = p3.z - p2.z; qx = hx - p2.x; qy = hy - p2.y; qz = hz - p2.z;
                    cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
                    if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
                    ex = p1.x - p3.x; ey = p1.y - p3.y; ez = p1.z - p3.z; qx = hx - p3.x; qy = hy - p3.y; qz = hz - p3.z;
                    cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
                    if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
                    tbest = t; ibest = t0 + j; kind = 1;
                });
            }
        }

        // Analytic spheres: hit_sphere, raytracer_v4.glsl:157-178 with a unit direction.  Exact roots only for the few
        // spheres whose line the ray crosses; the exact candidate rule of DESIGN.md §4.4 is re-checked there.
        if (HAS_SPH) {
            const uint32_t t0 = 0;
            if (alive) {
                scan_tile<false, false>(as_scene(A.sph), A.n_sph, cand, tid, ox, oy, oz, dx, dy, dz,
                                 [&](uint32_t j) { return SPH_LDS ? s_sph[j] : A.sph[j]; }, [&](uint32_t j, const float4 s) {
                    const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
                    const float h = fma_(cz, dz, fma_(cy, dy, cx * dx));
                    const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -s.w)));
                    const float disc = fma_(h, h, -c);
                    if (!((c < 0.0f) | ((disc > 0.0f) & (h > 0.0f)))) return;
                    const float sq = __builtin_sqrtf(disc);
                    float t = h - sq;
                    if (!(t > A.t_min)) t = h + sq;
                    if (t > A.t_min && t < tbest) { tbest = t; ibest = t0 + j; kind = 2; }
                });
            }
        }

        shade_lane<HAS_TRI, HAS_SPH>(A, P, alive, kind, ibest, tbest, A.sph, A.sph_invr, A.sph_mat, A.sph_kind);
    }
    if (lane == 0 && casts != 0) atomicAdd(A.cast_counter, casts);
}

// ------------------------------------------------------------------------------------------------------
// The candidate filter on the MATRIX cores
// ------------------------------------------------------------------------------------------------------
// For a unit direction d the discriminant of every (sphere, ray) pair is ONE dense contraction of 11 bilinear terms:
//     disc_ij = (d_i.(C_j - o_i))^2 - |C_j - o_i|^2 + r_j^2
//             = sum_{a<=b} (d_a d_b [x2 if a != b]) (C_a C_b)  +  sum_a (2 o_a - 2 (o.d) d_a) C_a  +  1 K_j  +  E_i 1
//     K_j = (r_j^2 - |C_j|^2) + eps (|C_j|^2 + r_j^2),    E_i = (o.d)^2 - |o|^2 (1 - eps)
// i.e. disc + margin with margin_ij = eps (|C_j|^2 + r_j^2 + |o_i|^2).  It runs on v_mfma_f32_32x32x16_bf16 with every f32 factor
// split into three bf16 parts (x = H + M + L) and the six leading cross products (HH, HM, MH, HL, LH, MM) laid out along K:
// 9 x 6 + 3 + 3 = 60 of the 64 K-slots of four chained MFMAs, so the accumulator holds the margin-inflated discriminant itself and
// its SIGN BIT is the candidate flag — one v_alignbit per pair on the vector ALU instead of the 10 instructions of the scalar test.
// The expanded form cancels catastrophically and is therefore used ONLY as a conservative filter: eps = 2e-5 covers its error
// (measured <= 0.05 eps (|C|^2 + r^2 + |o|^2) in tools/filter_model.py's pessimistic model) twenty times over, and the surviving
// pairs go through the same exact f32 evaluation as in k_trace, so images stay bit-identical.
// A = spheres (rows), B = rays (columns): lane l holds, for ray (l & 31) of the current column set, 16 results in its accumulator
// registers (rows (g&3) + 8(g>>2) + 4(l>>5)).  Spheres are assigned to rows so that accumulator register g of lane half w is sphere
// 16 w + 15 - g of the row block: after one v_permlane32_swap every lane owns the 32-bit candidate word of ITS OWN ray for the
// block, bit b <-> sphere 32 blk + b, which it parks in a lane-private LDS column until the exact tests run.
constexpr int      kMB        = 1024;      // threads per workgroup of the matrix-filter kernels (one workgroup per CU, 4 waves per SIMD)
constexpr uint32_t kMfmaSphMax = 512;      // 16 row blocks x 4 operand fragments x 1 KiB = 64 KiB of LDS
constexpr uint32_t kBitmapBytes = 16 * kMB * 4;   // candidate words: [16 row blocks][kMB lanes]
constexpr float    kFilterEps = 2e-5f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// K-slot layout.  Operand q (0..3) of the chain holds 16 K-elements, lane half hh supplies elements 8 hh .. 8 hh + 7 as four
// dwords i = 0..3 of two bf16 each.  With x_t / y_t the ray / sphere factor of term t (t = 0..8), E and K the two constants:
//     (q = 0,1,2; hh = 0)  ray (H x_2i, H x_2i+1)                      sphere (P y_2i, P y_2i+1),  P = H, M, L for q = 0, 1, 2
//     (q = 0,1;   hh = 1)  ray (M x_2i, M x_2i+1)                      sphere (P y_2i, P y_2i+1),  P = H, M
//     (q = 2;     hh = 1)  ray (L x_2i, L x_2i+1)                      sphere (H y_2i, H y_2i+1)
//     (q = 3;     hh = 0)  ray (H x_8, 1) x3, (M x_8, H E)             sphere (H y_8, H K), (M y_8, M K), (L y_8, L K), (H y_8, 1)
//     (q = 3;     hh = 1)  ray (M x_8, M E), (L x_8, L E), 0, 0        sphere (M y_8, 1), (H y_8, 1), 0, 0
// so a lane needs only three distinct ray-side register quads per column set (operands 0 and 1 share one).
// row of the A operand that holds sphere b (0..31) of a row block
__host__ __device__ constexpr uint32_t frag_row_of(uint32_t b) { return ((15u - (b & 15u)) & 3u) + 8u * ((15u - (b & 15u)) >> 2) + 4u * (b >> 4); }

__host__ __device__ inline uint32_t bf16_rn(float x) {            // round to nearest even, finite inputs
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__host__ __device__ inline float bf16_up(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
__host__ __device__ inline void split3(float x, uint32_t* parts /*[3]*/) {
    parts[0] = bf16_rn(x);
    const float r1 = x - bf16_up(parts[0]);
    parts[1] = bf16_rn(r1);
    parts[2] = bf16_rn(r1 - bf16_up(parts[1]));
}

// Sphere-side (A operand) fragment of one bounding sphere: out[q][hh][dword] = K elements 8 hh .. 8 hh + 7 of MFMA operand q.
// kj = filter_kj(|C|^2, r^2); a padding row uses C = 0, kj = -1e30 (never a candidate), an unbounded one kj = +1e30 (always).
__host__ __device__ inline void bound_frag_row(float cx, float cy, float cz, float kj, uint32_t out[4][2][4]) {
    uint32_t y[9][3], k[3];                                         // [term][part]
    const double x = cx, yy = cy, z = cz;
    split3((float)(x * x), y[0]); split3((float)(yy * yy), y[1]); split3((float)(z * z), y[2]);
    split3((float)(x * yy), y[3]); split3((float)(x * z), y[4]); split3((float)(yy * z), y[5]);
    split3(cx, y[6]); split3(cy, y[7]); split3(cz, y[8]);
    split3(kj, k);
    const uint32_t one = 0x3F80u;
    auto pk = [](uint32_t lo, uint32_t hi) { return lo | (hi << 16); };
    for (int i = 0; i < 4; i++) {
        for (int q = 0; q < 3; q++) out[q][0][i] = pk(y[2 * i][q], y[2 * i + 1][q]);
        for (int q = 0; q < 2; q++) out[q][1][i] = pk(y[2 * i][q], y[2 * i + 1][q]);
        out[2][1][i] = pk(y[2 * i][0], y[2 * i + 1][0]);
    }
    for (int i = 0; i < 3; i++) out[3][0][i] = pk(y[8][i], k[i]);
    out[3][0][3] = pk(y[8][0], one);
    out[3][1][0] = pk(y[8][1], one);
    out[3][1][1] = pk(y[8][0], one);
    out[3][1][2] = 0u; out[3][1][3] = 0u;
}
__host__ __device__ inline float filter_kj(double c2, double r2) { return (float)((r2 - c2) + (double)kFilterEps * (c2 + r2)); }
constexpr float kNeverCandidate = -1e30f, kAlwaysCandidate = 1e30f;

// Ray-side (B operand) fragments of the 64 rays of a wave: [column set (rays 0..31 / 32..63)][operands 0 and 1, operand 2, operand 3].
struct RayOperands { u32x4 b[2][3]; };
// v_cvt_pk_bf16_f32 (round to nearest even, two floats -> one dword).  HARDWARE NOTE (measured on MI355X, ROCm 7.2): a VALU
// instruction that consumes the result straight after the conversion can read a stale register — about one ray in 10^6 lost a
// candidate, differently in every run, until wait states were added; hipcc's hazard recognizer inserts none for this opcode
// (it does for v_permlane32_swap).  The conversion is therefore issued through inline asm with its own `s_nop 3`, which also
// covers the two wait states a following v_permlane32_swap needs.  tests/test_gpu_repeatability.py guards the property.
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 3" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// v_permlane32_swap(x, y): x's upper half-wave <-> y's lower half-wave; set0 = new x, set1 = new y
__device__ __forceinline__ void swap32(uint32_t x, uint32_t y, uint32_t& set0, uint32_t& set1) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    set0 = r[0]; set1 = r[1];
}
__device__ __forceinline__ float pk_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float pk_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ void build_ray_operands(float ox, float oy, float oz, float dx, float dy, float dz, bool alive, RayOperands& R) {
    const float od = dotf(ox, oy, oz, dx, dy, dz), oo = dotf(ox, oy, oz, ox, oy, oz);
    float x[9] = { dx * dx, dy * dy, dz * dz, 2.0f * dx * dy, 2.0f * dx * dz, 2.0f * dy * dz,
                   2.0f * (ox - od * dx), 2.0f * (oy - od * dy), 2.0f * (oz - od * dz) };
    float e = alive ? od * od - oo * (1.0f - kFilterEps) : -3e30f;      // a dead lane's column can never produce a candidate
    // three-way bf16 split of the ten factors, two at a time: part = cvt_pk(residuals), residual -= part
    uint32_t ph[4], pm[4], pl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { ph[i] = pk_bf16(x[2 * i], x[2 * i + 1]); x[2 * i] -= pk_lo(ph[i]); x[2 * i + 1] -= pk_hi(ph[i]); }
    const uint32_t h8 = pk_bf16(x[8], 1.0f);
    x[8] -= pk_lo(h8);
#pragma unroll
    for (int i = 0; i < 4; i++) { pm[i] = pk_bf16(x[2 * i], x[2 * i + 1]); x[2 * i] -= pk_lo(pm[i]); x[2 * i + 1] -= pk_hi(pm[i]); }
    const uint32_t m8a = pk_bf16(x[8], e);
    e -= pk_hi(m8a);
    const uint32_t m8b = pk_bf16(x[8], e);
    e -= pk_hi(m8b);
    x[8] -= pk_lo(m8a);
#pragma unroll
    for (int i = 0; i < 4; i++) pl[i] = pk_bf16(x[2 * i], x[2 * i + 1]);
    const uint32_t l8 = pk_bf16(x[8], e);
    // Lane (w, col) supplies elements 8 w .. 8 w + 7 of ray 32 S + col for column set S: lanes 0-31 keep their hh = 0 dwords for
    // set 0 and need their partner's for set 1, lanes 32-63 the mirror image with hh = 1 — v_permlane32_swap(a, b) exchanges a's
    // upper half with b's lower half, so swapping an (hh = 0 dword, hh = 1 dword) pair leaves the set-0 dword in a, set 1 in b.
    uint32_t s0[3][4], s1[3][4];
    const uint32_t lo3[4] = { h8, h8, h8, m8a }, hi3[4] = { m8b, l8, 0u, 0u };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        swap32(ph[i], pm[i], s0[0][i], s1[0][i]);
        swap32(ph[i], pl[i], s0[1][i], s1[1][i]);
        swap32(lo3[i], hi3[i], s0[2][i], s1[2][i]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        R.b[0][k] = u32x4{ s0[k][0], s0[k][1], s0[k][2], s0[k][3] };
        R.b[1][k] = u32x4{ s1[k][0], s1[k][1], s1[k][2], s1[k][3] };
    }
}

// The matrix-core scan of one LDS-resident tile of up to 16 row blocks (512 bounding spheres) against the 64 rays of the wave:
// per row block 8 MFMAs, 32 v_alignbit and one exchange.  Candidate word `blk` of this lane's ray goes to bm[blk * kMB] (bit b
// CLEAR <-> sphere 32 blk + b is a candidate); the return value has bit blk set when that word holds any candidate.
__device__ __forceinline__ uint32_t mfma_scan_tile(const u32x4* s_frag, uint32_t n_blocks, const RayOperands& R, uint32_t* bm, uint32_t lane) {
    uint32_t nz = 0;                                                // block blk -> bit n_blocks - 1 - blk
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 4) {                 // four row blocks per trip: their LDS offsets are immediates
        const u32x4* fr0 = s_frag + (size_t)b0 * 256 + lane;
        uint32_t* bm0 = bm + b0 * kMB;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if (b0 + u >= n_blocks) break;
            const u32x4* fr = fr0 + u * 256;
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, fr[0]), a1 = __builtin_bit_cast(bf16x8, fr[64]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, fr[128]), a3 = __builtin_bit_cast(bf16x8, fr[192]);
            const f32x16 zero = { 0 };
            f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, R.b[0][0]), zero, 0, 0, 0);
            f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, R.b[1][0]), zero, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, R.b[0][0]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, R.b[1][0]), d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(bf16x8, R.b[0][1]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(bf16x8, R.b[1][1]), d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, __builtin_bit_cast(bf16x8, R.b[0][2]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, __builtin_bit_cast(bf16x8, R.b[1][2]), d1, 0, 0, 0);
            uint32_t n0 = 0xFFFFFFFFu, n1 = 0xFFFFFFFFu;           // sign bits: register g -> bit 15 - g
#pragma unroll
            for (int g = 0; g < 16; g++) n0 = __builtin_amdgcn_alignbit(n0, __float_as_uint(d0[g]), 31);
#pragma unroll
            for (int g = 0; g < 16; g++) n1 = __builtin_amdgcn_alignbit(n1, __float_as_uint(d1[g]), 31);
            // lower lanes: own set-0 signs (rows of half 0) + the partner's set-0 signs (rows of half 1); upper lanes: set 1
            const auto sw = __builtin_amdgcn_permlane32_swap(n0, n1, false, false);
            const uint32_t w = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);
            bm0[u * kMB] = w;
            // nz = 2 nz + (w != ~0): compare into VCC, add with carry (the two wait states between a VALU write of VCC and a
            // VALU read of it are what hipcc itself inserts on gfx950)
            asm("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(nz) : "v"(w) : "vcc");
        }
    }
    return nz;
}

// Exact tests of the parked candidates of this lane's ray, in ascending sphere order.
struct CandIter { uint32_t nz, bits, blk, nb; };
__device__ __forceinline__ bool cand_next(CandIter& it, const uint32_t* bm, uint32_t& row) {
    if (it.bits == 0u) {
        if (it.nz == 0u) return false;
        const uint32_t hb = 31u - (uint32_t)__builtin_clz(it.nz);
        it.blk = it.nb - 1u - hb;
        it.nz ^= 1u << hb;
        it.bits = ~bm[it.blk * kMB];                                // non-zero: the scan set this word's nz bit
    }
    row = it.blk * 32u + (uint32_t)__builtin_ctz(it.bits);
    it.bits &= it.bits - 1u;
    return true;
}
template <class Eval>
__device__ __forceinline__ void mfma_flush(uint32_t nz, uint32_t n_blocks, const uint32_t* bm, Eval&& eval) {
    CandIter it = { nz, 0u, 0u, n_blocks };
    uint32_t row;
    while (cand_next(it, bm, row)) eval(row);
}
// Same, for exact tests that gather from global memory: `fetch(row)` (the first 16 bytes of the record) is issued one
// candidate ahead of `eval(row, record)`.
template <class Fetch, class Eval>
__device__ __forceinline__ void mfma_flush_prefetch(uint32_t nz, uint32_t n_blocks, const uint32_t* bm, Fetch&& fetch, Eval&& eval) {
    CandIter it = { nz, 0u, 0u, n_blocks };
    uint32_t row = 0, rown = 0;
    bool have = cand_next(it, bm, row);
    float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f), recn = rec;
    if (have) rec = fetch(row);
    while (have) {
        const bool haven = cand_next(it, bm, rown);
        if (haven) recn = fetch(rown);
        eval(row, rec);
        row = rown; rec = recn; have = haven;
    }
}

// The reference's plane + three-edge test of one face (same operations, same order as k_trace's face evaluation); returns t or NaN.
__device__ __forceinline__ bool face_hit(const float4 n, const float4* __restrict__ f, float ox, float oy, float oz, float dx, float dy, float dz,
                                         float t_lo, float t_hi, float& t_out) {
    const float nd = dot3(dx, dy, dz, n.x, n.y, n.z);
    if (nd == 0.0f) return false;
    const float t = (n.w - dot3(n.x, n.y, n.z, ox, oy, oz)) / nd;
    if (!(t >= t_lo && t <= t_hi)) return false;
    const float4 p1 = f[1], p2 = f[2], p3 = f[3];
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
    if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return false;
    t_out = t;
    return true;
}

// Sphere scenes of <= 512 spheres: everything the loop touches lives in LDS, waves never synchronise after the prologue.
__global__ __launch_bounds__(kMB) void k_trace_mfma(const TraceArgs A, const u32x4* __restrict__ frags, uint32_t n_blocks) {
    extern __shared__ u32x4 lds_dyn[];
    u32x4* s_frag = lds_dyn;                                                   // [n_blocks][4][64]
    float4* s_sph = reinterpret_cast<float4*>(s_frag + (size_t)n_blocks * 256);   // [n_blocks * 32] (cx, cy, cz, r^2) for the exact test
    float4* s_mat = s_sph + (size_t)n_blocks * 32;                               // materials, kinds, 1/r: read at every hit
    float* s_invr = reinterpret_cast<float*>(s_mat + (size_t)n_blocks * 32);
    uint32_t* s_kind = reinterpret_cast<uint32_t*>(s_invr + (size_t)n_blocks * 32);
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_kind + (size_t)n_blocks * 32); // [16][kMB] candidate words
    const uint32_t tid = threadIdx.x, lane = lane_id();
    for (uint32_t k = tid; k < n_blocks * 256; k += kMB) s_frag[k] = frags[k];
    for (uint32_t k = tid; k < n_blocks * 32; k += kMB) {
        const bool in = k < A.n_sph;
        s_sph[k] = in ? A.sph[k] : kPadSphere;
        s_mat[k] = in ? A.sph_mat[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_invr[k] = in ? A.sph_invr[k] : 0.0f;
        s_kind[k] = in ? A.sph_kind[k] : 0u;
    }
    __syncthreads();                                                            // the only barrier

    Path P;
    P.ox = P.oy = P.oz = 0.0f; P.dx = P.dy = 0.0f; P.dz = 1.0f;
    P.tr = P.tg = P.tb = 0.0f; P.lr = P.lg = P.lb = 0.0f; P.slot = 0; P.base = 0; P.depth = 0;
    bool alive = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    RayStock Q;
    Q.ox = Q.oy = Q.oz = 0.0f; Q.dx = Q.dy = 0.0f; Q.dz = 1.0f; Q.slot = 0; Q.base = 0; Q.n = 0;
    unsigned long long casts = 0, iters = 0;
#ifdef RT3_PROFILE
    unsigned long long prof_flush_iters = 0, prof_cands = 0, prof_refills = 0;
#endif

    for (;;) {
        refill_from_stock(A, lane, alive, P, Q, chunk_next, chunk_end, exhausted);
        const unsigned long long live = __ballot(alive);
        if (live == 0ull) break;
        casts += (unsigned long long)__popcll(live);
        iters++;
        const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;
        RayOperands R;
        build_ray_operands(ox, oy, oz, dx, dy, dz, alive, R);

        // nearest hit: exact evaluation of queued candidates (ties: lower sphere index, as the sequential loop)
        float tbest = __builtin_inff();
        uint32_t ibest = 0, kind = 0;
        auto eval = [&](uint32_t j) {
            const float4 s = s_sph[j];
            const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
            const float h = fma_(cz, dz, fma_(cy, dy, cx * dx));
            const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -s.w)));
            const float disc = fma_(h, h, -c);
            if (!((c < 0.0f) | ((disc > 0.0f) & (h > 0.0f)))) return;
            const float sq = __builtin_sqrtf(disc);
            float t = h - sq;
            if (!(t > A.t_min)) t = h + sq;
            if (t > A.t_min && (t < tbest || (t == tbest && j < ibest))) { tbest = t; ibest = j; kind = 2; }
        };
        const uint32_t nz = mfma_scan_tile(s_frag, n_blocks, R, s_bm + tid, lane);
#ifdef RT3_PROFILE
        {
            uint32_t mine = 0;
            CandIter it = { nz, 0u, 0u, n_blocks };
            uint32_t row;
            while (cand_next(it, s_bm + tid, row)) mine++;
            uint32_t mx = mine, sm = mine;
            for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); sm += (uint32_t)__shfl_xor((int)sm, o); }
            prof_flush_iters += mx; prof_cands += sm;
            prof_refills += (uint32_t)__popcll(__ballot(P.depth == 0 && alive));
        }
#endif
        mfma_flush(nz, n_blocks, s_bm + tid, eval);
        shade_lane<false, true>(A, P, alive, kind, ibest, tbest, s_sph, s_invr, s_mat, s_kind);
    }
    if (lane == 0 && casts != 0) { atomicAdd(A.cast_counter, casts); atomicAdd(A.cast_counter + 1, iters * n_blocks * 8ull); }
#ifdef RT3_PROFILE
    if (lane == 0) { atomicAdd(A.cast_counter + 2, prof_flush_iters); atomicAdd(A.cast_counter + 3, prof_cands); atomicAdd(A.cast_counter + 4, iters); atomicAdd(A.cast_counter + 5, prof_refills); }
#endif
}

// Any scene: faces (through their bounding spheres) and spheres, streamed through LDS in tiles of 512 rows.  The 16 waves of the
// workgroup move through the tiles together (two barriers per tile); the exact tests gather their records from global memory.
// Ties resolve as in the sequential loops: faces before spheres, then the lower index.
template <bool HAS_TRI, bool HAS_SPH>
__global__ __launch_bounds__(kMB) void k_trace_mfma_tiled(const TraceArgs A, const u32x4* __restrict__ tri_frags, const u32x4* __restrict__ sph_frags) {
    extern __shared__ u32x4 lds_dyn[];
    u32x4* s_frag = lds_dyn;                                                   // [16][4][64]
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_frag + 16 * 256);           // [16][kMB] candidate words
    const uint32_t tid = threadIdx.x, lane = lane_id();

    Path P;
    P.ox = P.oy = P.oz = 0.0f; P.dx = P.dy = 0.0f; P.dz = 1.0f;
    P.tr = P.tg = P.tb = 0.0f; P.lr = P.lg = P.lb = 0.0f; P.slot = 0; P.base = 0; P.depth = 0;
    bool alive = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    RayStock Q;
    Q.ox = Q.oy = Q.oz = 0.0f; Q.dx = Q.dy = 0.0f; Q.dz = 1.0f; Q.slot = 0; Q.base = 0; Q.n = 0;
    unsigned long long casts = 0, mfmas = 0;

    for (;;) {
        refill_from_stock(A, lane, alive, P, Q, chunk_next, chunk_end, exhausted);
        const unsigned long long live = __ballot(alive);
        if (!__syncthreads_or(live != 0ull ? 1 : 0)) break;                      // the workgroup ends together
        casts += (unsigned long long)__popcll(live);
        const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;
        RayOperands R;
        build_ray_operands(ox, oy, oz, dx, dy, dz, alive, R);
        float tbest = __builtin_inff();
        uint32_t ibest = 0, kind = 0;

        auto pass = [&](const u32x4* __restrict__ frags, uint32_t n_rows, auto&& fetch_row, auto&& eval_row) {
            const uint32_t total_blocks = (n_rows + 31u) / 32u;
            for (uint32_t b0 = 0; b0 < total_blocks; b0 += 16) {
                const uint32_t nb = min(16u, total_blocks - b0);
                __syncthreads();                                                // every wave is done with the previous tile
                for (uint32_t k = tid; k < nb * 256; k += kMB) s_frag[k] = frags[(size_t)b0 * 256 + k];
                __syncthreads();
                const uint32_t nz = mfma_scan_tile(s_frag, nb, R, s_bm + tid, lane);
                mfma_flush_prefetch(nz, nb, s_bm + tid, [&](uint32_t row) { return fetch_row(b0 * 32u + row); },
                                    [&](uint32_t row, const float4 rec) { eval_row(b0 * 32u + row, rec); });
                mfmas += nb * 8ull;
            }
        };
        if (HAS_TRI)
            pass(tri_frags, A.n_tri, [&](uint32_t j) { return A.tri[(size_t)min(j, A.n_tri - 1u) * 4]; }, [&](uint32_t j, const float4 n) {
                if (j >= A.n_tri) return;
                float t;
                if (!face_hit(n, A.tri + (size_t)j * 4, ox, oy, oz, dx, dy, dz, A.t_min, tbest, t)) return;
                if (t < tbest || j < ibest) { tbest = t; ibest = j; kind = 1; }      // t <= tbest here: equal t keeps the lower face index
            });
        if (HAS_SPH)
            pass(sph_frags, A.n_sph, [&](uint32_t j) { return A.sph[min(j, A.n_sph - 1u)]; }, [&](uint32_t j, const float4 s) {
                if (j >= A.n_sph) return;
                const float cx = s.x - ox, cy = s.y - oy, cz = s.z - oz;
                const float h = fma_(cz, dz, fma_(cy, dy, cx * dx));
                const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -s.w)));
                const float disc = fma_(h, h, -c);
                if (!((c < 0.0f) | ((disc > 0.0f) & (h > 0.0f)))) return;
                const float sq = __builtin_sqrtf(disc);
                float t = h - sq;
                if (!(t > A.t_min)) t = h + sq;
                if (t > A.t_min && (t < tbest || (t == tbest && kind == 2 && j < ibest))) { tbest = t; ibest = j; kind = 2; }
            });
        shade_lane<HAS_TRI, HAS_SPH>(A, P, alive, kind, ibest, tbest, A.sph, A.sph_invr, A.sph_mat, A.sph_kind);
    }
    if (lane == 0 && casts != 0) { atomicAdd(A.cast_counter, casts); atomicAdd(A.cast_counter + 1, mfmas); }
}

// Mode R through the matrix-core filter (camera at the origin, as k_mode_r_fast): one thread per pixel, 1024 pixels per
// workgroup, face bounding spheres streamed through LDS in tiles of 512; the reference's literal test runs on the candidates.
__global__ __launch_bounds__(kMB) void k_mode_r_mfma(const float4* __restrict__ tri, const u32x4* __restrict__ tri_frags,
                                                    const float4* __restrict__ face_rgb, uint32_t n_faces, CamDev cam,
                                                    uint32_t width, uint32_t height, uint32_t* __restrict__ out) {
    extern __shared__ u32x4 lds_dyn[];
    u32x4* s_frag = lds_dyn;
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_frag + 16 * 256);
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint32_t pixel = blockIdx.x * kMB + tid;
    const bool valid = pixel < width * height;
    const uint32_t x = valid ? pixel % width : 0u, y = valid ? pixel / width : 0u;
    const float u = (float)((double)(float)x / ((double)(float)width - 1.0));
    const float v = (float)((double)(float)(height - 1 - y) / ((double)(float)height - 1.0));
    const float ox = cam.ox, oy = cam.oy, oz = cam.oz;              // all zero (checked by the host)
    const float dx = ((cam.lx + u * cam.hx) + v * cam.vx) - ox;
    const float dy = ((cam.ly + u * cam.hy) + v * cam.vy) - oy;
    const float dz = ((cam.lz + u * cam.hz) + v * cam.vz) - oz;
    const float inv = 1.0f / __builtin_sqrtf(dot3(dx, dy, dz, dx, dy, dz));     // unit direction for the filter only
    RayOperands R;
    build_ray_operands(ox, oy, oz, dx * inv, dy * inv, dz * inv, valid, R);

    uint32_t min_i = 0;
    float min_t = __builtin_inff();
    const uint32_t total_blocks = (n_faces + 31u) / 32u;
    for (uint32_t b0 = 0; b0 < total_blocks; b0 += 16) {
        const uint32_t nb = min(16u, total_blocks - b0);
        __syncthreads();
        for (uint32_t k = tid; k < nb * 256; k += kMB) s_frag[k] = tri_frags[(size_t)b0 * 256 + k];
        __syncthreads();
        auto eval = [&](uint32_t row, const float4 n) {
            const uint32_t j = b0 * 32u + row;
            if (j >= n_faces) return;
            const float4* f = tri + (size_t)j * 4;
            const float nd = dot3(dx, dy, dz, n.x, n.y, n.z);       // SequentialRenderer.cpp:56
            if (nd == 0.0f) return;
            const float t = (dot3(n.x, n.y, n.z, ox, oy, oz) + n.w) / nd;      // :70
            if (t < 0.0f || t > min_t) return;                      // :71, with equality kept for the index rule below
            const float4 p1 = f[1], p2 = f[2], p3 = f[3];
            const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;
            float ex, ey, ez, qx, qy, qz, cx, cy, cz;
            ex = p2.x - p1.x; ey = p2.y - p1.y; ez = p2.z - p1.z; qx = hx - p1.x; qy = hy - p1.y; qz = hz - p1.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
            ex = p3.x - p2.x; ey = p3.y - p2.y; ez = p3.z - p2.z; qx = hx - p2.x; qy = hy - p2.y; qz = hz - p2.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
            ex = p1.x - p3.x; ey = p1.y - p3.y; ez = p1.z - p3.z; qx = hx - p3.x; qy = hy - p3.y; qz = hz - p3.z;
            cx = ey * qz - qy * ez; cy = ez * qx - qz * ex; cz = ex * qy - qx * ey;
            if (!(-dot3(n.x, n.y, n.z, cx, cy, cz) >= 0.0f)) return;
            if (t < min_t || j < min_i) { min_i = j; min_t = t; }   // "t >= min_t rejects" of :71 == the lowest index wins ties
        };
        const uint32_t nz = mfma_scan_tile(s_frag, nb, R, s_bm + tid, lane);
        mfma_flush_prefetch(nz, nb, s_bm + tid, [&](uint32_t row) { return tri[(size_t)min(b0 * 32u + row, n_faces - 1u) * 4]; }, eval);
    }
    if (!valid) return;
    float r, g, b;
    if (min_t < __builtin_inff()) { const float4 c = face_rgb[min_i]; r = c.x; g = c.y; b = c.z; }
    else sky(dx, dy, dz, r, g, b);
    out[pixel] = pack_pixel(r, g, b);
}

// reduce pass (what reduce_v1.glsl:66-76 was meant to be): samples are summed per pixel in sample order.
__global__ __launch_bounds__(kBlock) void k_accumulate(const float4* __restrict__ rad, float4* __restrict__ accum,
                                                      uint32_t npix, uint32_t ns, int first) {
    const uint32_t pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= npix) return;
    float4 a = first ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : accum[pix];
    for (uint32_t s = 0; s < ns; s++) {
        const float4 r = rad[(size_t)s * npix + pix];
        a.x = a.x + r.x; a.y = a.y + r.y; a.z = a.z + r.z;
    }
    accum[pix] = a;
}
__global__ __launch_bounds__(kBlock) void k_resolve(const float4* __restrict__ accum, uint32_t npix, uint32_t spp,
                                                   uint32_t flags, uint32_t* __restrict__ out) {
    const uint32_t pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= npix) return;
    const float4 a = accum[pix];
    const float n = (float)spp;
    float r = a.x / n, g = a.y / n, b = a.z / n;
    if (flags & RT3_FLAG_GAMMA2) {
        r = r > 0.0f ? __builtin_sqrtf(r) : 0.0f;
        g = g > 0.0f ? __builtin_sqrtf(g) : 0.0f;
        b = b > 0.0f ? __builtin_sqrtf(b) : 0.0f;
    }
    out[pix] = pack_pixel(r, g, b);
}

// device arithmetic probes for tests/test_gpu_arith.py
__global__ void k_debug_arith(const float* a, const float* b, uint32_t n, float* div, float* sq, float* fm,
                              float* cs, float* sn, float* sk, uint32_t* pk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    div[i] = a[i] / b[i];
    sq[i] = __builtin_sqrtf(__builtin_fabsf(a[i]));
    fm[i] = fma_(a[i], b[i], a[i]);
    const float u = u01(__float_as_uint(a[i]));
    sincos2pi(u, cs[i], sn[i]);
    float r, g, bl;
    sky(a[i], b[i], -2.0f, r, g, bl);
    sk[3 * i] = r; sk[3 * i + 1] = g; sk[3 * i + 2] = bl;
    pk[i] = pack_pixel(a[i], b[i], u);
}


// ------------------------------------------------------------------------------------------------------
// Device-side scene assembly: HIP equivalents of the reference's pre-render shaders and of the merge
// ------------------------------------------------------------------------------------------------------
struct SphereGen { float cx, cy, cz, radius; uint32_t m, p; float r, g, b; uint32_t face_offset, vertex_offset; };

// compute_point (Sphere.cpp:69-79 / pre_render_sphere_v2_vertices.glsl:77-83).  The CPU form is followed (double
// trig on a float ratio, rounded to float per component), not the shader's float trig, so that a device-tessellated
// sphere equals a host-tessellated one.
__device__ __forceinline__ float4 sphere_point(const SphereGen& s, float fx, float fy) {
    const double ty = M_PI * (double)(fy / (float)(s.p - 1));
    const double tx = 2 * M_PI * (double)(fx / (float)s.m);
    const float ux = (float)(sin(ty) * cos(tx)), uy = (float)cos(ty), uz = (float)(sin(ty) * sin(tx));
    return make_float4(s.cx + s.radius * ux, s.cy + s.radius * uy, s.cz + s.radius * uz, 0.0f);
}

// pre_render_sphere_v2_vertices.glsl:88-113: one thread per (meridian x, parallel y)
__global__ void k_prerender_sphere_vertices(SphereGen s, float4* __restrict__ verts) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x == 0 && y == 0) verts[s.vertex_offset] = sphere_point(s, 0.0f, 0.0f);
    else if (x < s.m && y > 0 && y < s.p - 1) verts[s.vertex_offset + 1 + (y - 1) * s.m + x] = sphere_point(s, (float)x, (float)y);
    else if (x == 0 && y == s.p - 1) verts[s.vertex_offset + 1 + (y - 1) * s.m] = sphere_point(s, 0.0f, (float)y);
}

__device__ __forceinline__ void store_face(rt3_gface* f, uint32_t a, uint32_t b, uint32_t c, float4 pa, float4 pb, float4 pc,
                                           const SphereGen& s) {
    // normal = normalize(cross(c - a, b - a)) with glm's evaluation order; colour = colour * |n . (0,0,-1)| (Sphere.cpp:153-155)
    const float ex = pc.x - pa.x, ey = pc.y - pa.y, ez = pc.z - pa.z, fx = pb.x - pa.x, fy = pb.y - pa.y, fz = pb.z - pa.z;
    const float nx = ey * fz - fy * ez, ny = ez * fx - fz * ex, nz = ex * fy - fx * ey;
    const float inv = 1.0f / __builtin_sqrtf(dot3(nx, ny, nz, nx, ny, nz));
    const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
    const float shade = __builtin_fabsf(ux * 0.0f + uy * 0.0f + uz * -1.0f);
    f->v1 = a; f->v2 = b; f->v3 = c; f->_pad0 = 0;
    f->normal[0] = ux; f->normal[1] = uy; f->normal[2] = uz; f->_pad1 = 0;
    f->color[0] = s.r * shade; f->color[1] = s.g * shade; f->color[2] = s.b * shade; f->_pad2 = 0;
}

// pre_render_sphere_v2_faces.glsl:83-194: one thread per (x, y >= 1); reads the vertices the first kernel wrote
__global__ void k_prerender_sphere_faces(SphereGen s, rt3_gface* __restrict__ faces, const float4* __restrict__ verts) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (x >= s.m || y >= s.p) return;
    const uint32_t xm1 = x > 0 ? x - 1 : s.m - 1, vo = s.vertex_offset;
    rt3_gface* out = faces + s.face_offset;
    if (y == 1) {
        const uint32_t a = vo, b = vo + 1 + xm1, c = vo + 1 + x;
        store_face(out + x, a, b, c, verts[a], verts[b], verts[c], s);
    } else if (y < s.p - 1) {
        const uint32_t base = s.m + 2 * (y - 2) * s.m;
        const uint32_t p1 = vo + 1 + (y - 2) * s.m + xm1, p2 = vo + 1 + (y - 2) * s.m + x;
        const uint32_t p3 = vo + 1 + (y - 1) * s.m + xm1, p4 = vo + 1 + (y - 1) * s.m + x;
        store_face(out + base + 2 * x, p1, p3, p4, verts[p1], verts[p3], verts[p4], s);
        store_face(out + base + 2 * x + 1, p1, p2, p4, verts[p1], verts[p2], verts[p4], s);
    } else {
        const uint32_t base = s.m + 2 * (y - 2) * s.m;
        const uint32_t a = vo + 1 + (y - 1) * s.m, b = vo + 1 + (y - 2) * s.m + xm1, c = vo + 1 + (y - 2) * s.m + x;
        store_face(out + base + x, a, b, c, verts[a], verts[b], verts[c], s);
    }
}

// De-indexes the merged GFace[] / vec4[] into what the render kernels read: 4 float4 per face (n + plane distance, p1, p2,
// p3), the bounding sphere of §5.1, the material.  Entries [n_faces, n_pad) of `bound` become never-hit records.
__global__ void k_commit_mesh(const rt3_gface* __restrict__ faces, const float4* __restrict__ verts, uint32_t n_faces, uint32_t n_pad,
                              uint32_t n_verts, const rt3_material* __restrict__ mats, float4* __restrict__ tri, float4* __restrict__ bound,
                              float4* __restrict__ mat, uint32_t* __restrict__ kind, uint32_t* __restrict__ error_flag,
                              u32x4* __restrict__ frag, uint32_t n_frag_rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // matrix-filter fragments: row i of block i/32, for all four operands and both lane halves (padding rows: never candidates)
    auto write_frag = [&](float cx, float cy, float cz, float kj) {
        if (i >= n_frag_rows) return;
        uint32_t fr[4][2][4];
        bound_frag_row(cx, cy, cz, kj, fr);
        for (int q = 0; q < 4; q++)
            for (int hh = 0; hh < 2; hh++)
                frag[((size_t)(i / 32) * 4 + q) * 64 + hh * 32 + frag_row_of(i % 32)] = u32x4{ fr[q][hh][0], fr[q][hh][1], fr[q][hh][2], fr[q][hh][3] };
    };
    if (i >= n_pad && i >= n_frag_rows) return;
    if (i >= n_faces) { if (i < n_pad) bound[i] = kPadSphere; write_frag(0.0f, 0.0f, 0.0f, kNeverCandidate); return; }
    const rt3_gface f = faces[i];
    if (f.v1 >= n_verts || f.v2 >= n_verts || f.v3 >= n_verts) { atomicOr(error_flag, 1u); bound[i] = kPadSphere; write_frag(0.0f, 0.0f, 0.0f, kNeverCandidate); return; }
    const float4 p1 = verts[f.v1], p2 = verts[f.v2], p3 = verts[f.v3];
    tri[4 * (size_t)i] = make_float4(f.normal[0], f.normal[1], f.normal[2], dot3(f.normal[0], f.normal[1], f.normal[2], p1.x, p1.y, p1.z));
    tri[4 * (size_t)i + 1] = make_float4(p1.x, p1.y, p1.z, 0.0f);
    tri[4 * (size_t)i + 2] = make_float4(p2.x, p2.y, p2.z, 0.0f);
    tri[4 * (size_t)i + 3] = make_float4(p3.x, p3.y, p3.z, 0.0f);
    // bounding sphere: centroid + largest vertex distance in double, inflated (0.1 % + 1e-5 * (1 + max |coordinate|)), r^2 rounded up
    const double cx = ((double)p1.x + p2.x + p3.x) / 3.0, cy = ((double)p1.y + p2.y + p3.y) / 3.0, cz = ((double)p1.z + p2.z + p3.z) / 3.0;
    double r2 = 0.0, big = 0.0;
    const float4 ps[3] = { p1, p2, p3 };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ddx = ps[k].x - cx, ddy = ps[k].y - cy, ddz = ps[k].z - cz;
        r2 = fmax(r2, ddx * ddx + ddy * ddy + ddz * ddz);
        big = fmax(big, fmax(fabs((double)ps[k].x), fmax(fabs((double)ps[k].y), fabs((double)ps[k].z))));
    }
    const double r = sqrt(r2) * 1.001 + 1e-5 * (1.0 + big);
    float r2f = (float)(r * r);
    if ((double)r2f < r * r) r2f = __uint_as_float(__float_as_uint(r2f) + 1u);
    if (!(r2f >= 0.0f)) r2f = __builtin_inff();                    // NaN / inf vertices: always a candidate, the exact test decides
    bound[i] = make_float4((float)cx, (float)cy, (float)cz, r2f);
    {
        const float fx = (float)cx, fy = (float)cy, fz = (float)cz;
        const double c2 = (double)fx * fx + (double)fy * fy + (double)fz * fz;
        write_frag(fx, fy, fz, r2f < __builtin_inff() ? filter_kj(c2, (double)r2f) : kAlwaysCandidate);   // r^2 = inf: the exact test decides
    }
    if (mats) {
        const rt3_material m = mats[i];
        if (m.kind == RT3_MAT_DIELECTRIC) {                         // same packing as pack_material() on the host
            const float ri_f = 1.0f / m.param, ri_b = m.param;
            float r0f = (1.0f - ri_f) / (1.0f + ri_f), r0b = (1.0f - ri_b) / (1.0f + ri_b);
            mat[i] = make_float4(ri_f, r0f * r0f, r0b * r0b, m.param);
        } else mat[i] = make_float4(m.rgb[0], m.rgb[1], m.rgb[2], m.param);
        kind[i] = m.kind;
    }
    else { mat[i] = make_float4(f.color[0], f.color[1], f.color[2], 0.0f); kind[i] = RT3_MAT_FLAT; }
}

}  // namespace

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
    float4* d_sph = nullptr; uint32_t* d_sph_frag = nullptr; float* d_sph_invr = nullptr; float4* d_sph_mat = nullptr; uint32_t* d_sph_kind = nullptr;

    // work buffers
    float4* d_rad = nullptr; size_t rad_entries = 0;
    float4* d_accum = nullptr; size_t accum_entries = 0;
    uint32_t* d_out = nullptr; size_t out_entries = 0;
    uint32_t* d_work = nullptr;                                     // [0] work counter
    unsigned long long* d_casts = nullptr;
    uint64_t rad_cap_bytes = 16ull << 30;
    bool force_plain_mode_r = false;                                // tests: compare the two Mode-R kernels

    // stats of the last render
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;              // per dominant-kernel launch
    uint32_t ev_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipStream_t last_stream = nullptr;
    uint64_t last_samples = 0;
    bool last_was_path = false;
    bool rendered = false;
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
// padding rows can never be candidates.
std::vector<uint32_t> build_sphere_frags(const float* center_radius, uint32_t n) {
    const uint32_t blocks = (n + 31u) / 32u;
    std::vector<uint32_t> out((size_t)blocks * 4 * 64 * 4, 0u);
    for (uint32_t j = 0; j < blocks * 32; j++) {
        uint32_t fr[4][2][4];
        if (j < n) {
            const float* s = center_radius + 4 * (size_t)j;
            const double c2 = (double)s[0] * s[0] + (double)s[1] * s[1] + (double)s[2] * s[2], r2 = (double)s[3] * s[3];
            bound_frag_row(s[0], s[1], s[2], filter_kj(c2, r2), fr);
        } else bound_frag_row(0.0f, 0.0f, 0.0f, kNeverCandidate, fr);
        for (int q = 0; q < 4; q++)
            for (int hh = 0; hh < 2; hh++)
                std::memcpy(&out[((((size_t)(j / 32) * 4 + q) * 64) + hh * 32 + frag_row_of(j % 32)) * 4], fr[q][hh], 16);
    }
    return out;
}

bool row_owned(const rt3_params* p, uint32_t y) {
    if (p->tile_count <= 1) return true;
    return ((y / p->tile_rows) % p->tile_count) == p->tile_index;
}

int check_params(rt3_ctx* ctx, const rt3_params* p) {
    if (!p) return fail(ctx, RT3_E_ARG, "params is NULL");
    if (p->width < 2 || p->height < 2) return fail(ctx, RT3_E_ARG, "width and height must be >= 2");
    if ((uint64_t)p->width * p->height > 0x7FFFFFFFull) return fail(ctx, RT3_E_ARG, "frame too large");
    if (p->spp < 1 || p->max_depth < 1) return fail(ctx, RT3_E_ARG, "spp and max_depth must be >= 1");
    if (p->tile_count > 1 && (p->tile_rows == 0 || p->tile_index >= p->tile_count))
        return fail(ctx, RT3_E_ARG, "bad tile_rows / tile_index / tile_count");
    return 0;
}

}  // namespace

extern "C" {

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
        (e = hipEventCreate(&ctx->ev_end)) != hipSuccess ||
        (e = hipMalloc((void**)&ctx->d_work, 64)) != hipSuccess || (e = hipMalloc((void**)&ctx->d_casts, 64)) != hipSuccess) {
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
    void* bufs[] = { ctx->d_gfaces, ctx->d_verts, ctx->d_face_mats_in, ctx->d_error, ctx->d_tri, ctx->d_tri_mat, ctx->d_tri_kind, ctx->d_tri_bound, ctx->d_tri_frag, ctx->d_sph, ctx->d_sph_frag, ctx->d_sph_invr, ctx->d_sph_mat, ctx->d_sph_kind,
                     ctx->d_rad, ctx->d_accum, ctx->d_out, ctx->d_work, ctx->d_casts };
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (auto& p : ctx->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
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
    for (void** b : { (void**)&ctx->d_tri, (void**)&ctx->d_tri_mat, (void**)&ctx->d_tri_kind, (void**)&ctx->d_tri_bound, (void**)&ctx->d_tri_frag, (void**)&ctx->d_face_mats_in })
        if (*b) { RT3_HIP(hipFree(*b)); *b = nullptr; }
    ctx->n_faces = 0;
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
    hipLaunchKernelGGL(k_commit_mesh, dim3((n_frag_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, ctx->d_gfaces, ctx->d_verts, n, n_pad,
                       ctx->cap_verts, ctx->d_face_mats_in, ctx->d_tri, ctx->d_tri_bound, ctx->d_tri_mat, ctx->d_tri_kind, ctx->d_error,
                       ctx->d_tri_frag, n_frag_rows);
    RT3_HIP(hipGetLastError());
    uint32_t err = 0;
    RT3_HIP(hipMemcpyAsync(&err, ctx->d_error, 4, hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    if (err) return fail(ctx, RT3_E_ARG, "a face references a vertex out of range");
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
    if ((rc = upload(ctx, &ctx->d_sph_frag, build_sphere_frags(center_radius, n)))) return rc;
    if ((rc = upload(ctx, &ctx->d_sph, sph)) || (rc = upload(ctx, &ctx->d_sph_invr, invr)) ||
        (rc = upload(ctx, &ctx->d_sph_mat, mat)) || (rc = upload(ctx, &ctx->d_sph_kind, kind)))
        return rc;
    ctx->n_sph = n;
    return 0;
}

int rt3_render_device(rt3_ctx* ctx, const rt3_camera* cam, uint32_t width, uint32_t height, void* d_out, void* stream_) {
    if (!ctx) return RT3_E_ARG;
    if (!cam || !d_out) return fail(ctx, RT3_E_ARG, "cam / d_out_pixels is NULL");
    if (width < 2 || height < 2 || (uint64_t)width * height > 0x7FFFFFFFull) return fail(ctx, RT3_E_ARG, "bad frame size");
    RT3_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    ctx->ev_used = 0;
    hipEvent_t a, b;
    int rc = take_event_pair(ctx, &a, &b);
    if (rc) return rc;
    const uint32_t npix = width * height;
    RT3_HIP(hipEventRecord(ctx->ev_begin, stream));
    RT3_HIP(hipEventRecord(a, stream));
    // k_mode_r_fast needs n.o == 0 exactly (camera at the origin, as Camera::update always builds it) and finite rays;
    // any other camera takes the plain brute-force kernel, which reproduces the reference for every input.
    const bool at_origin = cam->origin[0] == 0.0f && cam->origin[1] == 0.0f && cam->origin[2] == 0.0f;
    if (at_origin && !ctx->force_plain_mode_r && !getenv("RT3_NO_MFMA") && ctx->n_faces > 0) {
        const size_t lds = (size_t)16 * 4096 + (size_t)kBitmapBytes;
        RT3_HIP(hipFuncSetAttribute((const void*)k_mode_r_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mode_r_mfma, dim3((npix + kMB - 1) / kMB), dim3(kMB), lds, stream,
                           ctx->d_tri, (const u32x4*)ctx->d_tri_frag, ctx->d_tri_mat, ctx->n_faces, cam_dev(cam), width, height, (uint32_t*)d_out);
    } else if (at_origin && !ctx->force_plain_mode_r)
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

int rt3_render_path_device(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, void* d_out, void* stream_) {
    if (!ctx) return RT3_E_ARG;
    if (!cam || !d_out) return fail(ctx, RT3_E_ARG, "cam / d_out_pixels is NULL");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (ctx->n_sph == 0 && ctx->n_faces == 0) return fail(ctx, RT3_E_STATE, "no scene: call rt3_set_spheres / rt3_set_mesh first");
    RT3_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;

    const uint32_t rows = rt3_rows_owned(p);
    const uint32_t npix = rows * p->width;
    ctx->ev_used = 0;
    ctx->last_stream = stream;
    ctx->last_samples = (uint64_t)npix * p->spp;
    ctx->last_was_path = true;
    ctx->rendered = true;
    RT3_HIP(hipEventRecord(ctx->ev_begin, stream));
    RT3_HIP(hipMemsetAsync(ctx->d_casts, 0, 64, stream));
    if (npix == 0) { RT3_HIP(hipEventRecord(ctx->ev_end, stream)); return 0; }

    // batch size: per-sample storage of 16 B per (pixel, sample), capped
    uint64_t per_spp = (uint64_t)npix * sizeof(float4);
    uint32_t batch = (uint32_t)std::min<uint64_t>(p->spp, std::max<uint64_t>(1, ctx->rad_cap_bytes / per_spp));
    batch = (uint32_t)std::min<uint64_t>(batch, 0x7FFF0000ull / npix);
    if (batch == 0) return fail(ctx, RT3_E_ARG, "frame too large for one sample batch");
    if ((rc = ensure(ctx, &ctx->d_rad, &ctx->rad_entries, (size_t)npix * batch))) return rc;
    if ((rc = ensure(ctx, &ctx->d_accum, &ctx->accum_entries, (size_t)npix))) return rc;

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

    using TraceKernel = void (*)(const TraceArgs);
    const bool has_tri = ctx->n_faces > 0, has_sph = ctx->n_sph > 0;
    const bool sph_lds = has_sph && ctx->n_sph <= kSphLdsMax;
    const size_t lds_bytes = sph_lds ? (size_t)ctx->n_sph * sizeof(float4) : 0;
    const TraceKernel kernel = has_tri ? (has_sph ? (sph_lds ? k_trace<true, true, true> : k_trace<true, true, false>) : k_trace<true, false, false>)
                                       : (sph_lds ? k_trace<false, true, true> : k_trace<false, true, false>);
    // candidate filter on the matrix cores (RT3_NO_MFMA=1 keeps the VALU scan, for A/B runs): the all-in-LDS kernel for sphere
    // scenes of <= 512 spheres, the tiled kernel for everything else
    const bool use_mfma = !getenv("RT3_NO_MFMA");
    const bool mfma_single = use_mfma && !has_tri && has_sph && ctx->n_sph <= kMfmaSphMax && !getenv("RT3_FORCE_TILED");   // (A/B knob)
    const uint32_t mfma_blocks = (ctx->n_sph + 31u) / 32u;
    const size_t mfma_lds = mfma_single ? (size_t)mfma_blocks * (4096 + 32 * (16 + 16 + 4 + 4)) + (size_t)kBitmapBytes
                                        : (size_t)16 * 4096 + (size_t)kBitmapBytes;
    using TiledKernel = void (*)(const TraceArgs, const u32x4*, const u32x4*);
    const TiledKernel tiled = has_tri ? (has_sph ? k_trace_mfma_tiled<true, true> : k_trace_mfma_tiled<true, false>) : k_trace_mfma_tiled<false, true>;
    int per_cu = 0;
    if (mfma_single) {
        RT3_HIP(hipFuncSetAttribute((const void*)k_trace_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_lds));
        RT3_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_mfma, kMB, mfma_lds));
    } else if (use_mfma) {
        RT3_HIP(hipFuncSetAttribute((const void*)tiled, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_lds));
        RT3_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tiled, kMB, mfma_lds));
    } else RT3_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds_bytes));
    if (per_cu < 1) return fail(ctx, RT3_E_DEVICE, "k_trace does not fit on a CU");
    per_cu = std::min(per_cu, 8);

    for (uint32_t s0 = 0; s0 < p->spp; s0 += batch) {
        const uint32_t ns = std::min(batch, p->spp - s0);
        A.s0 = s0;
        A.total = npix * ns;
        const uint32_t waves_per_block = use_mfma ? kMB / 64 : kBlock / 64;
        const uint32_t want_blocks = (A.total + kWorkChunk * waves_per_block - 1) / (kWorkChunk * waves_per_block);
        const uint32_t grid = std::max(1u, std::min<uint32_t>((uint32_t)(ctx->num_cu * per_cu), want_blocks));
        hipEvent_t a, b;
        if ((rc = take_event_pair(ctx, &a, &b))) return rc;
        RT3_HIP(hipMemsetAsync(ctx->d_work, 0, 4, stream));
        RT3_HIP(hipEventRecord(a, stream));
        if (mfma_single) hipLaunchKernelGGL(k_trace_mfma, dim3(grid), dim3(kMB), mfma_lds, stream, A, (const u32x4*)ctx->d_sph_frag, mfma_blocks);
        else if (use_mfma) hipLaunchKernelGGL(tiled, dim3(grid), dim3(kMB), mfma_lds, stream, A, (const u32x4*)ctx->d_tri_frag, (const u32x4*)ctx->d_sph_frag);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds_bytes, stream, A);
        RT3_HIP(hipGetLastError());
        RT3_HIP(hipEventRecord(b, stream));
        hipLaunchKernelGGL(k_accumulate, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                           ctx->d_rad, ctx->d_accum, npix, ns, s0 == 0 ? 1 : 0);
        RT3_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_resolve, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       ctx->d_accum, npix, p->spp, p->flags, (uint32_t*)d_out);
    RT3_HIP(hipGetLastError());
    RT3_HIP(hipEventRecord(ctx->ev_end, stream));
    return 0;
}

int rt3_render_path(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* p, uint32_t* out_pixels) {
    if (!ctx) return RT3_E_ARG;
    if (!out_pixels) return fail(ctx, RT3_E_ARG, "out_pixels is NULL");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    RT3_HIP(hipSetDevice(ctx->device));
    const size_t npix = (size_t)rt3_rows_owned(p) * p->width;
    if ((rc = ensure(ctx, &ctx->d_out, &ctx->out_entries, std::max<size_t>(npix, 1)))) return rc;
    if ((rc = rt3_render_path_device(ctx, cam, p, ctx->d_out, ctx->stream))) return rc;
    if (npix) RT3_HIP(hipMemcpyAsync(out_pixels, ctx->d_out, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT3_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int rt3_get_stats(rt3_ctx* ctx, rt3_stats* out) {
    if (!ctx || !out) return RT3_E_ARG;
    RT3_HIP(hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof *out);
    if (!ctx->rendered) return fail(ctx, RT3_E_STATE, "no render has been issued on this context");
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
        unsigned long long counters[8] = { 0 };
        RT3_HIP(hipMemcpy(counters, ctx->d_casts, 64, hipMemcpyDeviceToHost));
#ifdef RT3_PROFILE
        fprintf(stderr, "[rt3 profile] wave iterations %llu, flush iterations/wave-iter %.2f, candidates/ray %.2f, live lanes/wave-iter %.1f, "
                        "fresh paths/wave-iter %.1f\n", counters[4], (double)counters[2] / (double)counters[4],
                (double)counters[3] / (double)counters[0], (double)counters[0] / (double)counters[4], (double)counters[5] / (double)counters[4]);
#endif
        out->ray_casts = counters[0];
        out->prim_tests = counters[0] * ((uint64_t)ctx->n_sph + ctx->n_faces);
        out->mfma_instructions = counters[1];
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
    RT3_HIP(hipFree(d));
    return 0;
}

}  // extern "C"
