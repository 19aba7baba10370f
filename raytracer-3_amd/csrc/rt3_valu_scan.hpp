// rt3_valu_scan.hpp — the VALU scan (scan_tile) and the kernels built on it: k_mode_r, k_mode_r_fast, k_trace
// Part of rt3_device.hip (one translation unit, gfx950 only); included from there, in this order.
#pragma once

namespace {

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

// SPH_LDS: the sphere array (<= kSphLdsMax entries) is also copied to LDS once per block, for the per-lane gathers of
// the exact evaluation (an LDS gather costs ~64 cycles, a global one an L2 round trip per candidate).
template <bool HAS_TRI, bool HAS_SPH, bool SPH_LDS, bool REF = false>
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
        refill_lanes<REF>(A, lane, alive, P, chunk_next, chunk_end, exhausted);
        if (__ballot(alive) == 0ull) break;                         // waves are independent: no block-level barrier anywhere
        casts += (unsigned long long)__popcll(__ballot(alive));

        // ---- nearest hit.  kind: 0 none, 1 triangle, 2 sphere; strict '<' keeps the earlier primitive.
        float tbest = __builtin_inff();
        uint32_t ibest = 0, kind = 0;
        const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;

        // Faces: hit_vertex, raytracer_v4.glsl:116-153 (= ray_color's test, SequentialRenderer.cpp:53-98, with the sign of
        // n.o corrected).  The hot loop tests the ray against a slightly inflated bounding sphere of every face; the
        // reference's plane + three-edge test, in its own operation order, runs only for the faces that survive.
        // REF: ray cast 0 keeps the reference's unnormalised direction and literal formula; the scan sees the unit direction
        // (the host sends such a render here only when the camera sits at the origin, where the bound is valid for that formula).
        if (HAS_TRI) {
            if (alive) {
                const bool ref0 = REF && P.depth == 0;
                float ux = dx, uy = dy, uz = dz;
                if (ref0) { const float inv = 1.0f / __builtin_sqrtf(dot3(dx, dy, dz, dx, dy, dz)); ux = dx * inv; uy = dy * inv; uz = dz * inv; }
                scan_tile<true, true>(as_scene(A.tri_bound), A.n_tri, cand, tid, ox, oy, oz, ux, uy, uz,
                                [&](uint32_t j) { return A.tri[(size_t)j * 4]; }, [&](uint32_t j, const float4 n) {
                    float t;
                    if (!face_hit<true>(n, A.tri + (size_t)j * 4, ox, oy, oz, dx, dy, dz, A.t_min, tbest, ref0, t)) return;
                    tbest = t; ibest = j; kind = 1;
                });
            }
        }

        // Analytic spheres: hit_sphere, raytracer_v4.glsl:157-178 with a unit direction.  Exact roots only for the few
        // spheres whose line the ray crosses; the exact candidate rule of DESIGN.md §4.4 is re-checked there.
        if (HAS_SPH) {
            if (alive) {
                scan_tile<false, false>(as_scene(A.sph), A.n_sph, cand, tid, ox, oy, oz, dx, dy, dz,
                                 [&](uint32_t j) { return SPH_LDS ? s_sph[j] : A.sph[j]; }, [&](uint32_t j, const float4 s) {
                    float t;
                    if (sphere_root(s, ox, oy, oz, dx, dy, dz, A.t_min, t) && t < tbest) { tbest = t; ibest = j; kind = 2; }
                });
            }
        }

        shade_lane<HAS_TRI, HAS_SPH, REF>(A, P, alive, kind, ibest, tbest, A.sph, A.sph_invr, A.sph_mat, A.sph_kind);
    }
    if (lane == 0 && casts != 0) atomicAdd(A.cast_counter, casts);
}

// Mode X with NO candidate filter: every ray against every face, then every sphere, in index order, exactly as the sequential
// oracle loops — no bounding spheres, no matrix cores, no queues.  It is the on-GPU arbiter for the filtered kernels (tests,
// tools/fuzz_filter.py; rt3_debug_force_brute / RT3_BRUTE=1) and the only kernel that can serve RT3_FLAG_REFERENCE_PRIMARY with a
// camera off the origin (the reference's literal formula then puts the "hit point" off the face's plane, where no bound holds).
// Scene records are wave-uniform loads through the constant address space (scalar cache); one path per lane, refill by ballot.
template <bool REF>
__global__ __launch_bounds__(kBlock) void k_trace_brute(const TraceArgs A) {
    const uint32_t lane = lane_id();
    Path P;
    P.ox = P.oy = P.oz = 0.0f; P.dx = P.dy = 0.0f; P.dz = 1.0f;
    P.tr = P.tg = P.tb = 0.0f; P.lr = P.lg = P.lb = 0.0f; P.slot = 0; P.base = 0; P.depth = 0;
    bool alive = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    unsigned long long casts = 0;
    const scene_ptr tri = as_scene(A.tri), sph = as_scene(A.sph);
    auto f4 = [](const f32x4 v) { return make_float4(v.x, v.y, v.z, v.w); };

    for (;;) {
        refill_lanes<REF>(A, lane, alive, P, chunk_next, chunk_end, exhausted);
        if (__ballot(alive) == 0ull) break;
        casts += (unsigned long long)__popcll(__ballot(alive));
        float tbest = __builtin_inff();
        uint32_t ibest = 0, kind = 0;
        const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;
        const bool ref0 = REF && P.depth == 0;
        for (uint32_t j = 0; j < A.n_tri; j++) {
            const float4 n = f4(tri[4 * (size_t)j]), p1 = f4(tri[4 * (size_t)j + 1]), p2 = f4(tri[4 * (size_t)j + 2]), p3 = f4(tri[4 * (size_t)j + 3]);
            float t;
            if (!alive || !face_t(n, ox, oy, oz, dx, dy, dz, ref0, t)) continue;
            if (!(t >= A.t_min && t < tbest)) continue;
            if (!face_inside(n, p1, p2, p3, ox, oy, oz, dx, dy, dz, t)) continue;
            tbest = t; ibest = j; kind = 1;
        }
        for (uint32_t j = 0; j < A.n_sph; j++) {
            const float4 s = f4(sph[j]);
            float t;
            if (alive && sphere_root(s, ox, oy, oz, dx, dy, dz, A.t_min, t) && t < tbest) { tbest = t; ibest = j; kind = 2; }
        }
        shade_lane<true, true, REF>(A, P, alive, kind, ibest, tbest, A.sph, A.sph_invr, A.sph_mat, A.sph_kind);   // kind says which arrays to read
    }
    if (lane == 0 && casts != 0) atomicAdd(A.cast_counter, casts);
}

}  // namespace
