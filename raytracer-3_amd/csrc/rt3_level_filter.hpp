// rt3_level_filter.hpp — k_trace_levels: the multi-level candidate filter of DESIGN.md 5.2e as ONE loop over its stages
// Part of rt3_device.hip (one translation unit, gfx950 only); included after rt3_matrix_filter.hpp, whose scan / push / pair-list pieces it uses.
//
// Levels (8 children per node; primitives in the order of the spatial median split of rt3_device.hip):
//   top     what the matrix cores scan (K = 32 filter, mfma32k_scan_tile): LEVELS == 3: a row = 64 primitives; LEVELS == 4: a super-row = 512
//   rows    LEVELS == 4 only: the 8 rows of a candidate super-row, their bounds tested in f32                       (A.*_rowb)
//   leaves  the 8 leaf groups of a candidate row, bounds in f32                                                      (A.*_leaf)
//   members the 8 primitives of a candidate leaf: spheres — the exact test's own candidate rule; faces — the face's bound (A.*_grp)
//   exact   survivors, 64 at a time: sphere_root | the reference's plane + three-edge test
// Every level hands its survivors to the next through a wave-private pair list in LDS; the (ray lane, top row) pairs of the scan are set aside in the
// wave's strip of global memory (after one look at the top row's own bound: rows behind the ray cost nothing further).  k_trace_mfma_tiled nested
// the levels as lambdas — every level's body inlined once per unrolled child of the level above it, 12 000-30 000 lines of ISA per kernel.  Here
// each stage is ONE piece of code and `pump` runs whichever has a full batch, deepest first; the lists take a whole batch's worst case (8 children of 32
// parents: 64 KiB of LDS for the four lists of 16 waves).  Smaller lists with a make-room branch in every stage were built and measured 8-20 % slower (the
// survivors' ballots and pushes in two phases); so were four levels on the BASELINE scenes (100 000 spheres x1.24, 47 106 faces x1.28: 8 row tests per
// candidate super-row cost more than scanning 1 500 rows).  The host therefore keeps k_trace_mfma_tiled's resident three-level form while the rows of 64 fit
// in LDS (<= 112 000 primitives) and runs this kernel with four levels beyond (300 000 spheres x0.95, 10^6 x0.83, 3 10^6 x0.64 against the tiled three-level
// form).  Frames are k_trace_mfma_tiled's bit for bit: the nearest-hit key does not care in which order pairs are tested.
#pragma once

namespace {

constexpr uint32_t kLevFan = 8;                                     // children per node at every level (kGroupTri = kGroupSph = kSuper = 8)
constexpr uint32_t kLevListCap = 31u + 32u * kLevFan + 1u;          // a list below its consumer's threshold (32) + one batch's worst case: 288
constexpr uint32_t kLevExactCap = 63u + 32u * kLevFan + 1u;         // the exact stage takes 64 at a time: 320
constexpr uint32_t kLevBmBlocks = 4;                                // row blocks per push (resident rows) ...
constexpr uint32_t kLevBmBlocksTiled = 8;                           // ... and when the top rows stream through a tile
constexpr uint32_t kLevTileLoads = 2;                               // 16-byte vectors per thread and tile: 32 KiB = 16 row blocks
__host__ __device__ constexpr uint32_t lev_list_words(uint32_t levels) { return kPairCap + (levels == 4u ? kLevListCap : 0u) + kLevListCap + kLevExactCap; }
__host__ __device__ constexpr size_t lev_lds_fixed(uint32_t levels, bool res) {     // everything but the resident rows
    return (size_t)kTB * 8u + (size_t)(res ? kLevBmBlocks : kLevBmBlocksTiled) * kTB * 4u + (size_t)(kTB / 64u) * lev_list_words(levels) * 4u +
           (res ? 0u : (size_t)kTB * kLevTileLoads * 16u);
}
// (one block of headroom: a kernel with any static LDS beside the dynamic request — __syncthreads_or's word, say — is refused at exactly 160 KiB)
__host__ __device__ constexpr uint32_t lev_resident_blocks(uint32_t levels) { return (uint32_t)((160u * 1024u - lev_lds_fixed(levels, true)) / 2048u) - 1u; }
static_assert(lev_lds_fixed(3, true) + lev_resident_blocks(3) * 2048u < 160u * 1024u && lev_lds_fixed(4, true) + lev_resident_blocks(4) * 2048u < 160u * 1024u &&
              lev_lds_fixed(3, false) < 160u * 1024u && lev_lds_fixed(4, false) < 160u * 1024u, "k_trace_levels: LDS");

template <bool HAS_TRI, bool HAS_SPH, bool REF, uint32_t LEVELS, bool RES>
__global__ __launch_bounds__(kTB) void k_trace_levels(const TraceArgs A, const u32x4* __restrict__ tri_frags, const u32x4* __restrict__ sph_frags) {
    static_assert(LEVELS == 3 || LEVELS == 4, "three or four levels");
    static_assert(kGroupTri == kLevFan && kGroupSph == kLevFan && kSuper == kLevFan, "k_trace_levels: 8 children per node");
    constexpr uint32_t BM = RES ? kLevBmBlocks : kLevBmBlocksTiled;
    constexpr uint32_t LPP = 2u, PER = 64u / LPP, MPL = kLevFan / LPP;      // lanes per pair, pairs per batch, children per lane (profiles/r03_ab_lanes_per_pair.log)
    extern __shared__ u32x4 lds_dyn[];
    const uint32_t tri_blocks = HAS_TRI ? (A.n_tri_top + 31u) / 32u : 0u, sph_blocks = HAS_SPH ? (A.n_sph_top + 31u) / 32u : 0u;
    u32x4* s_frag = lds_dyn;                                                   // RES: every top row block (faces' first); else one tile
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_frag + (RES ? (size_t)(tri_blocks + sph_blocks) * 128u : (size_t)kTB * kLevTileLoads));
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_bm + BM * kTB);
    uint32_t* s_lists = reinterpret_cast<uint32_t*>(s_key + kTB);
    const uint32_t tid = threadIdx.x, lane = lane_id();
    uint32_t* pairs = s_lists + (tid / 64u) * lev_list_words(LEVELS);          // raw (pushing lane, bit) pairs of the scan
    uint32_t* spairs = pairs + kPairCap;                                       // LEVELS == 4: (ray lane, row) pairs
    uint32_t* lpairs = spairs + (LEVELS == 4 ? kLevListCap : 0u);              // (ray lane, leaf) pairs
    uint32_t* fpairs = lpairs + kLevListCap;                                   // (ray lane, face index | sphere position) pairs for the exact test
    unsigned long long* keys = s_key + (tid & ~63u);
    uint32_t* strip = A.pair_strips + ((size_t)blockIdx.x * (kTB / 64u) + tid / 64u) * kStripPairs;
    if constexpr (RES) {
        for (uint32_t k = tid; k < tri_blocks * 128u; k += kTB) s_frag[k] = tri_frags[k];
        for (uint32_t k = tid; k < sph_blocks * 128u; k += kTB) s_frag[tri_blocks * 128u + k] = sph_frags[k];
        __syncthreads();                                                        // the only barrier
    }

    Path P;
    P.ox = P.oy = P.oz = 0.0f; P.dx = P.dy = 0.0f; P.dz = 1.0f;
    P.tr = P.tg = P.tb = 0.0f; P.lr = P.lg = P.lb = 0.0f; P.slot = 0; P.base = 0; P.depth = 0;
    bool alive = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    unsigned long long casts = 0, mfmas = 0, exact = 0, bound_tests = 0;

    for (;;) {
        refill_lanes<REF>(A, lane, alive, P, chunk_next, chunk_end, exhausted);
        const unsigned long long live = __ballot(alive);
        if constexpr (RES) { if (live == 0ull) break; }
        else if (!__syncthreads_or(live != 0ull ? 1 : 0)) break;
        casts += (unsigned long long)__popcll(live);
        const LaneRay ray = { P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, REF && P.depth == 0 };
        float ux = ray.dx, uy = ray.dy, uz = ray.dz;                            // what the filter and the bounds see: a unit direction
        if (REF && ray.literal) { const float inv = 1.0f / __builtin_sqrtf(dot3(ux, uy, uz, ux, uy, uz)); ux = ux * inv; uy = uy * inv; uz = uz * inv; }
        keys[lane] = HAS_SPH ? direct_tests(A, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, [&](uint32_t j) { return A.sph[j]; }) : kKeyNone;
        uint32_t n_pairs = 0, n_s = 0, n_l = 0, n_f = 0, n_strip = 0, spos = 0;

        // A batch of (ray lane, parent) pairs, two lanes per pair, each lane four of the parent's eight children (its own 64 contiguous bytes of
        // `bounds`): a child survives when the line meets its bound (scan_tile's arithmetic, margin m c), unless the bound lies behind the ray's origin
        // (centre behind, origin outside: both roots negative) or — BEYOND — is entered only beyond the ray's best hit so far.  Survivors go to `out`.
        // the survivors of the four children a lane looked at go to `out` (which holds a whole batch's worst case)
        auto push4 = [&](const bool (&keep)[MPL], uint32_t src, const uint32_t (&value)[MPL], uint32_t* out, uint32_t& n_out) {
#pragma unroll
            for (uint32_t m = 0; m < MPL; m++) {
                const unsigned long long km = __ballot(keep[m]);
                if (keep[m]) out[n_out + prefix_count(km)] = (src << kPairLaneShift) | value[m];
                n_out += (uint32_t)__popcll(km);
            }
        };
        static_assert(MPL == 4, "push4");
        auto expand = [&](uint32_t pair, bool valid, uint32_t part, const float4* __restrict__ bounds, uint32_t n_parents, auto beyond_tag,
                          uint32_t* out, uint32_t& n_out) {
            constexpr bool BEYOND = decltype(beyond_tag)::value && !REF;
            const uint32_t src = pair >> kPairLaneShift, g = pair & ((1u << kPairLaneShift) - 1u);
            const int sl = (int)src;
            const float sox = __shfl(ray.ox, sl), soy = __shfl(ray.oy, sl), soz = __shfl(ray.oz, sl);
            const float sdx = __shfl(ux, sl), sdy = __shfl(uy, sl), sdz = __shfl(uz, sl);
            bound_tests += (unsigned long long)__popcll(__ballot(valid)) * MPL;
            const bool ok = valid && g < n_parents;
            const uint32_t c0 = ok ? g * kLevFan + part * MPL : 0u;
            const float tb = BEYOND ? __uint_as_float(reinterpret_cast<const uint32_t*>(keys)[2u * src + 1u]) : __builtin_inff();
            float4 b[MPL];
#pragma unroll
            for (uint32_t m = 0; m < MPL; m++) b[m] = bounds[c0 + m];
            bool keep[MPL];
            uint32_t value[MPL];
#pragma unroll
            for (uint32_t m = 0; m < MPL; m++) {
                const float cx = b[m].x - sox, cy = b[m].y - soy, cz = b[m].z - soz;
                const float h = fma_(cz, sdz, fma_(cy, sdy, cx * sdx));
                const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -b[m].w)));
                const float disc = fma_(1e-4f, c, fma_(h, h, -c));
                bool drop = (h < 0.0f) & (c > 2e-3f * b[m].w);
                if constexpr (BEYOND) drop |= (tb < h) & (fma_(tb, fma_(-2.0f, h, tb), c) > 1e-3f * (c + b[m].w));
                keep[m] = ok && (((__float_as_uint(disc) >> 31) == 0u && !drop) || b[m].w >= 3e38f);      // (a bound with an unbounded member: always)
                value[m] = c0 + m;
            }
            push4(keep, src, value, out, n_out);
        };
        // 64 decoded (ray lane, top row) pairs on their way to the strip: one lane per pair looks at the top row's own bound
        auto set_aside = [&](uint32_t v, bool valid, const float4* __restrict__ topb, auto beyond_tag) {
            const uint32_t src = v >> kPairLaneShift, g = v & ((1u << kPairLaneShift) - 1u);
            const int sl = (int)src;
            const float sox = __shfl(ray.ox, sl), soy = __shfl(ray.oy, sl), soz = __shfl(ray.oz, sl);
            const float sdx = __shfl(ux, sl), sdy = __shfl(uy, sl), sdz = __shfl(uz, sl);
            if (__ballot(valid) == 0ull) return;
            const float4 b = topb[valid ? g : 0u];
            const float cx = b.x - sox, cy = b.y - soy, cz = b.z - soz;
            const float h = fma_(cz, sdz, fma_(cy, sdy, cx * sdx));
            const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -b.w)));
            bool drop = (h < 0.0f) & (c > 2e-3f * b.w);
            if constexpr (decltype(beyond_tag)::value && !REF) {
                const float tb = __uint_as_float(reinterpret_cast<const uint32_t*>(keys)[2u * src + 1u]);
                drop |= (tb < h) & (fma_(tb, fma_(-2.0f, h, tb), c) > 1e-3f * (c + b.w));
            }
            bound_tests += (unsigned long long)__popcll(__ballot(valid));
            const bool keep = valid && !drop;
            const unsigned long long km = __ballot(keep);
            if (keep) strip[n_strip + prefix_count(km)] = v;
            n_strip += (uint32_t)__popcll(km);
        };
        // The stages below `members_fn` / `exact_fn` differ between faces and spheres; the pump is the same.
        // rowb / n_top: LEVELS == 4, the rows' bounds and the number of super-rows; leafb / n_rows: the leaves' bounds and the number of rows
        auto pump = [&](bool final, const float4* __restrict__ rowb, uint32_t n_top, const float4* __restrict__ leafb, uint32_t n_rows, auto beyond_tag,
                        auto&& members_fn, auto&& exact_fn) {
            auto exact_batch = [&]() {
                const uint32_t take = min(n_f, 64u);
                n_f -= take;
                const bool v = lane < take;
                exact_fn(v ? fpairs[n_f + lane] : 0u, v);
                __builtin_amdgcn_wave_barrier();
            };
            auto members_batch = [&]() {
                const uint32_t take = min(n_l, PER);
                n_l -= take;
                const bool v = lane / LPP < take;
                members_fn(v ? lpairs[n_l + lane / LPP] : 0u, v, lane % LPP);
                __builtin_amdgcn_wave_barrier();
            };
            auto rows_batch = [&]() {                                               // LEVELS == 4: (ray lane, row) -> the row's leaves
                const uint32_t take = min(n_s, PER);
                n_s -= take;
                const bool v = lane / LPP < take;
                expand(v ? spairs[n_s + lane / LPP] : 0u, v, lane % LPP, leafb, n_rows, beyond_tag, lpairs, n_l);
                __builtin_amdgcn_wave_barrier();
            };
            if (n_strip != 0u) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the strip was written by other lanes of this wave
            bool have_next = false;
            uint32_t next_pr = 0u;
            for (;;) {
                const bool dry = spos >= n_strip;                                  // nothing left above the lists
                if (n_f >= 64u || (final && n_f != 0u && n_l == 0u && n_s == 0u && dry)) exact_batch();
                else if (n_l >= PER || (final && n_l != 0u && n_s == 0u && dry)) members_batch();
                else if (LEVELS == 4 && (n_s >= PER || (final && n_s != 0u && dry))) { if constexpr (LEVELS == 4) rows_batch(); }
                else if (!dry) {
                    const uint32_t take = min(n_strip - spos, PER);
                    const bool v = lane / LPP < take;
                    if (!have_next) next_pr = strip[min(spos + lane / LPP, n_strip - 1u)];
                    const uint32_t pr = next_pr;
                    spos += take;
                    have_next = spos < n_strip;                                    // the following batch's pairs: in flight while this one is tested
                    if (have_next) next_pr = strip[min(spos + lane / LPP, n_strip - 1u)];
                    if constexpr (LEVELS == 4) expand(pr, v, lane % LPP, rowb, n_top, beyond_tag, spairs, n_s);
                    else expand(pr, v, lane % LPP, leafb, n_rows, beyond_tag, lpairs, n_l);
                    __builtin_amdgcn_wave_barrier();
                } else break;
            }
            n_strip = 0u; spos = 0u;                                               // the strip is consumed (lists below their thresholds may remain unless `final`)
        };
        // One pass: scan the top rows, push, set aside; pump when the strip fills and at the end.
        auto pass = [&](const u32x4* __restrict__ frags, uint32_t frag_block0, uint32_t n_top, const float4* __restrict__ topb, float ccx, float ccy, float ccz,
                        auto beyond_tag, auto&& run_pump) {
            if (n_top == 0u) return;                                            // (every sphere on the direct list: no rows, no bounds)
            RayOperands32 R32;                                                  // coordinates about the filter's centre
            build_ray_operands32(ray.ox - ccx, ray.oy - ccy, ray.oz - ccz, ux, uy, uz, alive, R32);
            constexpr uint32_t kTile = kTB * kLevTileLoads / 128u;                 // row blocks per tile
            const uint32_t total_blocks = (n_top + 31u) / 32u;
            for (uint32_t b0 = 0; b0 < total_blocks; b0 += RES ? total_blocks : kTile) {
                const uint32_t nb = RES ? total_blocks : min(kTile, total_blocks - b0);
                const u32x4* tile = RES ? s_frag + (size_t)frag_block0 * 128u : s_frag;
                if constexpr (!RES) {
                    fill_tile<kTB, kLevTileLoads>(s_frag, frags + (size_t)b0 * 128u, nb * 128u, tid);
                    if (live == 0ull) continue;                                 // a wave without rays (the tail of a launch) only keeps the barriers
                }
                for (uint32_t h0 = 0; h0 < nb; h0 += BM) {
                    const uint32_t hb = min(BM, nb - h0);
                    const uint32_t nz = mfma32k_scan_tile<kTB, !RES>(tile + (size_t)h0 * 128u, hb, R32, s_bm + tid, lane, h0, 3);
                    push_pairs16_spill<kTB, RT3_DECODE_FP6 != 0>(nz, hb, s_bm + tid, (b0 + h0) * 32u, lane, pairs, n_pairs, [&](uint32_t v) {
                        set_aside(v, true, topb, beyond_tag);
                        if (n_strip + 64u > kStripPairs) run_pump(false);         // (rare: 128 candidate top rows per ray)
                    });
                    mfmas += hb * 8ull;
                }
            }
            { const bool lv = lane < n_pairs; set_aside(lv ? pair_decode<RT3_DECODE_FP6 != 0>(pairs[lane]) : 0u, lv, topb, beyond_tag); n_pairs = 0u; }
            run_pump(true);
        };

        if constexpr (HAS_TRI) {
            auto face_test = [&](uint32_t pair, bool valid) {                  // the reference's plane + three-edge test of (ray lane, face position) pairs
                const uint32_t src = pair >> kPairLaneShift, pos = pair & ((1u << kPairLaneShift) - 1u);
                const LaneRay r = fetch_ray<REF>(ray, src);
                exact += (unsigned long long)__popcll(__ballot(valid));
                if (!valid) return;
                const float4* f = A.tri_rec + (size_t)pos * 4;                    // the record in group order: a leaf's faces share cache lines
                const float4 n = f[0], p1 = f[1], p2 = f[2], p3 = f[3];
                const float t_hi = __uint_as_float((uint32_t)(keys[src] >> 32));   // the ray's best t so far: farther faces need no edge tests
                float t;
                if (!face_t(n, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.literal, t) || !(t >= A.t_min && t <= t_hi)) return;
                if (!face_inside(n, p1, p2, p3, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, t)) return;
                if (t < __builtin_inff()) atomicMin(&keys[src], hit_key(t, 0u, A.tri_perm[pos]));      // the key carries the face's own index
            };
            auto face_members = [&](uint32_t pair, bool valid, uint32_t part) {  // a leaf's faces: their own bounds (margin 1e-5 c, as the VALU kernels')
                const uint32_t src = pair >> kPairLaneShift, g = pair & ((1u << kPairLaneShift) - 1u);
                const int sl = (int)src;
                const float sox = __shfl(ray.ox, sl), soy = __shfl(ray.oy, sl), soz = __shfl(ray.oz, sl);
                const float sdx = __shfl(ux, sl), sdy = __shfl(uy, sl), sdz = __shfl(uz, sl);
                bound_tests += (unsigned long long)__popcll(__ballot(valid)) * MPL;
                const bool ok = valid && g < A.n_tri_leaves;
                const uint32_t p0 = ok ? g * kLevFan + part * MPL : 0u;
                float4 b[MPL];
#pragma unroll
                for (uint32_t m = 0; m < MPL; m++) b[m] = A.tri_grp[p0 + m];
                bool keep[MPL];
                uint32_t value[MPL];
#pragma unroll
                for (uint32_t m = 0; m < MPL; m++) {
                    const float cx = b[m].x - sox, cy = b[m].y - soy, cz = b[m].z - soz;
                    const float h = fma_(cz, sdz, fma_(cy, sdy, cx * sdx));
                    const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -b[m].w)));
                    const float disc = fma_(1e-5f, c, fma_(h, h, -c));
                    const bool behind = (h < 0.0f) & (c > 2e-3f * b[m].w);
                    // a face without a bounded hit region (r^2 = 3e38, possibly a non-finite centre) goes to the exact test whatever the arithmetic made of it
                    keep[m] = ok && (((__float_as_uint(disc) >> 31) == 0u && !behind) || b[m].w >= 3e38f);
                    value[m] = p0 + m;
                }
                push4(keep, src, value, fpairs, n_f);
            };
            auto run = [&](bool final) { pump(final, A.tri_rowb, A.n_tri_top, A.tri_leaf, A.n_tri_rows, std::true_type(), face_members, face_test); };
            pass(tri_frags, 0u, A.n_tri_top, A.tri_topb, A.tcx, A.tcy, A.tcz, std::true_type(), run);
        }
        if constexpr (HAS_SPH) {
            auto sphere_exact = [&](uint32_t pair, bool valid) {                // (ray lane, member position) pairs that passed the candidate rule
                const uint32_t src = pair >> kPairLaneShift, pos = pair & ((1u << kPairLaneShift) - 1u);
                const LaneRay r = fetch_ray<false>(ray, src);
                if (!valid) return;
                float t;
                if (sphere_root(A.sph_grp[pos], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, A.t_min, t) && t < __builtin_inff())
                    atomicMin(&keys[src], hit_key(t, 1u, A.sph_perm[pos]));
            };
            auto sphere_members = [&](uint32_t pair, bool valid, uint32_t part) {   // a leaf's spheres: sphere_root's own candidate rule, in its own arithmetic
                const uint32_t src = pair >> kPairLaneShift, g = pair & ((1u << kPairLaneShift) - 1u);
                const LaneRay r = fetch_ray<false>(ray, src);
                exact += (unsigned long long)__popcll(__ballot(valid)) * MPL;
                const bool ok = valid && g < A.n_sph_leaves;
                const uint32_t p0 = ok ? g * kLevFan + part * MPL : 0u;
                float4 sm[MPL];
#pragma unroll
                for (uint32_t m = 0; m < MPL; m++) sm[m] = A.sph_grp[p0 + m];
                bool keep[MPL];
                uint32_t value[MPL];
#pragma unroll
                for (uint32_t m = 0; m < MPL; m++) {
                    const float cx = sm[m].x - r.ox, cy = sm[m].y - r.oy, cz = sm[m].z - r.oz;
                    const float h = fma_(cz, r.dz, fma_(cy, r.dy, cx * r.dx));
                    const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -sm[m].w)));
                    const float disc = fma_(h, h, -c);
                    keep[m] = ok && ((c < 0.0f) | ((disc > 0.0f) & (h > 0.0f)));
                    value[m] = p0 + m;
                }
                push4(keep, src, value, fpairs, n_f);
            };
            auto run = [&](bool final) { pump(final, A.sph_rowb, A.n_sph_top, A.sph_leaf, A.n_sph_rows, std::false_type(), sphere_members, sphere_exact); };
            pass(sph_frags, tri_blocks, A.n_sph_top, A.sph_topb, A.fcx, A.fcy, A.fcz, std::false_type(), run);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t kind, ibest;
        float tbest;
        key_decode(keys[lane], kind, ibest, tbest);
        shade_lane<HAS_TRI, HAS_SPH, REF>(A, P, alive, kind, ibest, tbest, A.sph, A.sph_invr, A.sph_mat, A.sph_kind);
    }
    if (lane == 0 && casts != 0) { atomicAdd(A.cast_counter, casts); atomicAdd(A.cast_counter + 1, mfmas); atomicAdd(A.cast_counter + 2, exact); atomicAdd(A.cast_counter + 3, bound_tests); }
}

}  // namespace
