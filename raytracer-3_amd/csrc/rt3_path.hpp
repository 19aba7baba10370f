// rt3_path.hpp — per-wave path bookkeeping of Mode X: refill (plain and through the ray stock) and shade_lane()
// Part of rt3_device.hip (one translation unit, gfx950 only); included from there, in this order.
#pragma once

namespace {

// HAS_TRI / HAS_SPH compile the face loop / sphere loop (and the matching shading) in or out, so that a sphere-only
// scene does not pay registers or code for the triangle path.
// Refill: lanes whose path ended take the next samples of the wave's chunk (ballot + prefix count); a chunk of
// kWorkChunk samples is fetched from the global queue with one atomic when the wave runs dry.
template <bool REF = false>
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
        if (item != 0xFFFFFFFFu) { start_path<REF>(A, item, P); alive = true; }
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
template <bool REF = false>
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
        start_path<REF>(A, min(chunk_next + lane, chunk_end - 1u), T);
        Q.ox = T.ox; Q.oy = T.oy; Q.oz = T.oz; Q.dx = T.dx; Q.dy = T.dy; Q.dz = T.dz; Q.slot = T.slot; Q.base = T.base;
        Q.n = n_new;
        chunk_next += n_new;
    }
}

// Shade / scatter one ray cast of every live lane (book materials).  kind: 0 miss, 1 face, 2 sphere.
// The four per-sphere arrays read at a hit are parameters: global memory in k_trace, LDS copies in k_trace_mfma.
// REF (RT3_FLAG_REFERENCE_PRIMARY): at ray cast 0 the direction is the reference's unnormalised one — the sky and the hit point use
// it as it is (SequentialRenderer.cpp:77,105-107); the scatter formulas then get the unit direction.
template <bool HAS_TRI, bool HAS_SPH, bool REF = false>
__device__ __forceinline__ void shade_lane(const TraceArgs& A, Path& P, bool& alive, uint32_t kind, uint32_t ibest, float tbest,
                                           const float4* sph, const float* sph_invr, const float4* sph_mat, const uint32_t* sph_kind) {
    const float ox = P.ox, oy = P.oy, oz = P.oz;
    float dx = P.dx, dy = P.dy, dz = P.dz;
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
                if (REF && P.depth == 0) {
                    const float inv = 1.0f / __builtin_sqrtf(dot3(dx, dy, dz, dx, dy, dz));
                    dx = dx * inv; dy = dy * inv; dz = dz * inv;
                }
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
            A.rad[P.slot] = Rgb{ P.lr, P.lg, P.lb };
            alive = false;
        }
    }
}

}  // namespace
