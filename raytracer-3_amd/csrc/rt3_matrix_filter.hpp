// rt3_matrix_filter.hpp — the candidate filter on the matrix cores and its kernels: k_trace_mfma32 (scenes of <= 512 spheres: the bench
// kernel), k_trace_mfma_tiled (every other scene up to 112 000 primitives; beyond: k_trace_levels, rt3_level_filter.hpp), k_mode_r_mfma (Mode R),
// k_trace_mfma (round 1's K = 64 kernel, the A/B reference).
// Order of the file: the K = 64 form on v_mfma_f32_32x32x16_bf16 (the derivation; k_trace_mfma only), its 16x16x32 variant (A/B reference
// for faces), the K = 32 form on v_mfma_f32_16x16x32_bf16 that every default kernel runs, the pair list, the kernels.
// Part of rt3_device.hip (one translation unit, gfx950 only); included from there, in this order.
#pragma once

namespace {

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
#ifndef RT3_FACE_K32
#define RT3_FACE_K32 1                                               // 1: faces go through the K = 32 form like the spheres (fragments: k_commit_mesh);
#endif                                                              // 0: the K = 64 16x16x32 form (A/B reference; tools/filter_probe measures both)

// Ray-side (B operand) fragments of the 64 rays of a wave: [column set (rays 0..31 / 32..63)][operands 0 and 1, operand 2, operand 3].
struct RayOperands { u32x4 b[2][3]; };
// v_cvt_pk_bf16_f32 (round to nearest even, two floats -> one dword).  HARDWARE NOTE (MI355X, ROCm 7.2; evidence: profiles/
// r02_cvt_hazard.md, probe: tools/cvt_hazard_probe.hip): a vector-ALU instruction issued in the slot right behind the conversion that
// reads its result can get the register's OLD content — about 3 in 10^6 ray casts lost a candidate, differently in every run.  hipcc
// emits this opcode for a plain float2 -> bf16x2 conversion and schedules the dependent shift straight behind it (its hazard
// recognizer knows the 2 wait states of v_permlane32_swap, which are never short here, and the cvt_scale forwarding rule, but has no
// rule for this opcode): that build fails tests/test_gpu_repeatability.py, while every build in which one instruction or one wait
// state separates the two passes.  So the conversion is issued through inline asm with ONE trailing wait state (round 1 used 2 + 4).
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
#ifdef RT3_PK_BF16_COMPILER                                         // the build that FAILS tests/test_gpu_repeatability.py: hipcc's own conversion (experiments only:
    typedef float f32x2_ __attribute__((ext_vector_type(2)));       // tools/cvt_isa_check.py compares its ISA with the product's)
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){ lo, hi }, bf16x2_));
#else
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 0" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#endif
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
// STRIDE: words between the candidate words of consecutive row blocks (kMB: one column per thread of the workgroup; 64: wave-private).
template <uint32_t STRIDE = kMB>
__device__ __forceinline__ uint32_t mfma_scan_tile(const u32x4* s_frag, uint32_t n_blocks, const RayOperands& R, uint32_t* bm, uint32_t lane) {
    uint32_t nz = 0;                                                // block blk -> bit n_blocks - 1 - blk
    if (n_blocks == 0) return nz;
    const u32x4* fr = s_frag + lane;
    // the operand fragments of a row block are fetched from LDS while the vector ALU decodes the previous block (their
    // registers are free as soon as that block's last MFMA has issued): the LDS latency is off the critical path
    u32x4 a0 = fr[0], a1 = fr[64], a2 = fr[128], a3 = fr[192];
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 4) {                 // four row blocks per trip: their LDS offsets are immediates
        uint32_t* bm0 = bm + b0 * STRIDE;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if (b0 + u >= n_blocks) break;
            const f32x16 zero = { 0 };
            f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, R.b[0][0]), zero, 0, 0, 0);
            f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, R.b[1][0]), zero, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, R.b[0][0]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, R.b[1][0]), d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, R.b[0][1]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, R.b[1][1]), d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3), __builtin_bit_cast(bf16x8, R.b[0][2]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3), __builtin_bit_cast(bf16x8, R.b[1][2]), d1, 0, 0, 0);
            if (b0 + u + 1 < n_blocks) {                            // next block's fragments, in flight during the decode below
                const u32x4* fn = fr + (size_t)(b0 + u + 1) * 256;
                a0 = fn[0]; a1 = fn[64]; a2 = fn[128]; a3 = fn[192];
            }
            uint32_t n0 = 0xFFFFFFFFu, n1 = 0xFFFFFFFFu;           // sign bits: register g -> bit 15 - g
#pragma unroll
            for (int g = 0; g < 16; g++) n0 = __builtin_amdgcn_alignbit(n0, __float_as_uint(d0[g]), 31);
#pragma unroll
            for (int g = 0; g < 16; g++) n1 = __builtin_amdgcn_alignbit(n1, __float_as_uint(d1[g]), 31);
            // lower lanes: own set-0 signs (rows of half 0) + the partner's set-0 signs (rows of half 1); upper lanes: set 1
            const auto sw = __builtin_amdgcn_permlane32_swap(n0, n1, false, false);
            const uint32_t w = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);
            bm0[u * STRIDE] = w;
            // nz = 2 nz + (w != ~0): compare into VCC, add with carry (the two wait states between a VALU write of VCC and a
            // VALU read of it are what hipcc itself inserts on gfx950)
            asm("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(nz) : "v"(w) : "vcc");
        }
    }
    return nz;
}

// Exact tests of the parked candidates of this lane's ray, in ascending sphere order.
struct CandIter { uint32_t nz, bits, blk, nb; };
template <uint32_t STRIDE = kMB>
__device__ __forceinline__ bool cand_next(CandIter& it, const uint32_t* bm, uint32_t& row) {
    if (it.bits == 0u && it.nz != 0u) {                             // (one divergent region; the rest is straight-line for every lane)
        const uint32_t hb = 31u - (uint32_t)__builtin_clz(it.nz);
        it.blk = it.nb - 1u - hb;
        it.nz ^= 1u << hb;
        it.bits = ~bm[it.blk * STRIDE];                             // non-zero: the scan set this word's nz bit
    }
    const bool have = it.bits != 0u;
    row = it.blk * 32u + ((uint32_t)__ffs((int)it.bits) - 1u);      // (bits == 0: a value nobody uses)
    it.bits &= it.bits - 1u;
    return have;
}
template <class Eval>
__device__ __forceinline__ void mfma_flush(uint32_t nz, uint32_t n_blocks, const uint32_t* bm, Eval&& eval) {
    CandIter it = { nz, 0u, 0u, n_blocks };
    uint32_t row;
    while (cand_next(it, bm, row)) eval(row);
}
// ------------------------------------------------------------------------------------------------------
// The same K = 64 filter on v_mfma_f32_16x16x32_bf16 (round 2's first step away from 32x32x16; -DRT3_FACE_K32=0 still selects it for faces)
// ------------------------------------------------------------------------------------------------------
// Under this load the chip is clock-limited, and it holds a higher clock on the 16x16x32 shape: tools/ubench_mfma_shape.hip — the
// scan's instruction mix (4 ds_read_b128 + K = 64 of MFMA + 32 v_alignbit per 32 rows x 64 rays, four waves per SIMD, random operands) —
// sustains 1530 TFLOP/s with 8 x 32x32x16 and 1720 with 16 x 16x16x32 (MI355X_MICROARCH.md, DVFS give-back, item 7).  Same products,
// other K order: the 64 K-slots are six groups of ten, (ray part of terms 0..8, c) x (sphere part of terms 0..8, c'):
//     A  (H x, 1)   x (H y, H K)      B  (H x, 1)   x (M y, M K)      C  (H x, 1)   x (L y, L K)
//     D  (M x, H E) x (H y, 1)        E  (M x, M E) x (M y, 1)        F  (L x, L E) x (H y, 1)        + four empty slots
// so the ray side of a group is five dwords the three-way split produces anyway: (ph0..ph3, h8), (pm0..pm3, m8a | m8b), (pl0..pl3, l8).
// Operand (h, q) of a row block holds rows 16 h .. 16 h + 15 (= spheres 32 blk + 16 h + c, natural order), K-slots 32 q .. 32 q + 31;
// lane (g, c) = 16 g + c supplies row / column c, K-slots 32 q + 8 g .. + 7, and receives rows 4 g .. 4 g + 3 of column c.
// A lane therefore ends a row block with 8 results (h, j) for each of FOUR rays 16 G + c: its candidate word is bit 8 G + 4 h + j <->
// (ray lane 16 G + c, sphere 16 h + 4 g + j).  It is not the word of the lane's own ray, and need not be: the pair list takes (ray lane,
// primitive) pairs from whichever lane found them.
// (Folding the 16 results of a half block with 8 v_max3_i32 first and decoding only when some lane's maximum has its sign bit clear —
// 18 % of the half blocks on the 100 000-sphere scene — was built and measured x1.019 / x1.024 slower, although a bare loop gains 8 %.)
struct RayOperands16 { u32x4 b[4][2]; };                            // [ray group G][q]: K-slice of this lane's group for ray 16 G + c
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Sphere side: out[q][g][dword] = K-slots 32 q + 8 g .. + 7 of one row (kj, never / always: as bound_frag_row).
__host__ __device__ inline void bound_frag16_row(float cx, float cy, float cz, float kj, uint32_t out[2][4][4]) {
    uint32_t y[9][3], k[3];                                         // [term][part]
    const double x = cx, yy = cy, z = cz;
    split3((float)(x * x), y[0]); split3((float)(yy * yy), y[1]); split3((float)(z * z), y[2]);
    split3((float)(x * yy), y[3]); split3((float)(x * z), y[4]); split3((float)(yy * z), y[5]);
    split3(cx, y[6]); split3(cy, y[7]); split3(cz, y[8]);
    split3(kj, k);
    const uint32_t one = 0x3F80u;
    uint32_t slot[64];
    const int part_of_group[6] = { 0, 1, 2, 0, 1, 0 };              // sphere part of groups A..F
    for (int grp = 0; grp < 6; grp++) {
        for (int t = 0; t < 9; t++) slot[10 * grp + t] = y[t][part_of_group[grp]];
        slot[10 * grp + 9] = grp < 3 ? k[grp] : one;
    }
    for (int i = 60; i < 64; i++) slot[i] = 0u;
    for (int q = 0; q < 2; q++)
        for (int g = 0; g < 4; g++)
            for (int d = 0; d < 4; d++) out[q][g][d] = slot[32 * q + 8 * g + 2 * d] | (slot[32 * q + 8 * g + 2 * d + 1] << 16);
}
// where the fragment of (row block, row b of it, q, lane group g) lives: [blk][operand 2 h + q][lane 16 g + c]
__host__ __device__ constexpr size_t frag16_index(uint32_t blk, uint32_t b, uint32_t q, uint32_t g) {
    return ((size_t)blk * 4 + 2u * (b >> 4) + q) * 64 + 16u * g + (b & 15u);
}

// v_permlane16_swap(x, y): x's odd 16-lane rows <-> y's even rows
__device__ __forceinline__ void swap16(uint32_t x, uint32_t y, uint32_t& nx, uint32_t& ny) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    nx = r[0]; ny = r[1];
}
// Ray side.  Every lane splits the ten factors of ITS ray as build_ray_operands does, which gives the 32 dwords D of its K vector;
// lane (g, c) then needs dwords 16 q + 4 g .. + 3 of the rays of lanes (G, c), G = 0..3: a 4 x 4 transpose of 8-dword pieces across the
// four 16-lane rows, done in two stages (half-waves with v_permlane32_swap, then rows with v_permlane16_swap) after which register
// (source group G = 2 s + p) is the same register in every lane.
__device__ __forceinline__ void build_ray_operands16(float ox, float oy, float oz, float dx, float dy, float dz, bool alive, RayOperands16& R) {
    const float od = dotf(ox, oy, oz, dx, dy, dz), oo = dotf(ox, oy, oz, ox, oy, oz);
    float x[9] = { dx * dx, dy * dy, dz * dz, 2.0f * dx * dy, 2.0f * dx * dz, 2.0f * dy * dz,
                   2.0f * (ox - od * dx), 2.0f * (oy - od * dy), 2.0f * (oz - od * dz) };
    float e = alive ? od * od - oo * (1.0f - kFilterEps) : -3e30f;      // a dead lane's column can never produce a candidate
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
    // the K vector in dwords: groups A, B, C = (ph0..3, h8) three times, D = (pm0..3, m8a), E = (pm0..3, m8b), F = (pl0..3, l8), two empty
    const uint32_t D[32] = { ph[0], ph[1], ph[2], ph[3], h8, ph[0], ph[1], ph[2], ph[3], h8, ph[0], ph[1], ph[2], ph[3], h8,
                             pm[0], pm[1], pm[2], pm[3], m8a, pm[0], pm[1], pm[2], pm[3], m8b, pl[0], pl[1], pl[2], pl[3], l8, 0u, 0u };
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // what this lane sends to lane group g: D[16 q + 4 g + i].  Stage 1 delivers across the half-waves ...
            uint32_t x0, y0, x1, y1;
            swap32(D[16 * q + i], D[16 * q + 8 + i], x0, y0);           // destinations g = 0 | 2
            swap32(D[16 * q + 4 + i], D[16 * q + 12 + i], x1, y1);      // destinations g = 1 | 3
            // ... now x* = from the lower half-wave, y* = from the upper one; stage 2 delivers across the rows of a half-wave
            uint32_t s0e, s0o, s1e, s1o;
            swap16(x0, x1, s0e, s0o);                                   // source half 0: from its even row (G = 0), from its odd row (G = 1)
            swap16(y0, y1, s1e, s1o);                                   // source half 1: G = 2, G = 3
            R.b[0][q][i] = s0e; R.b[1][q][i] = s0o; R.b[2][q][i] = s1e; R.b[3][q][i] = s1o;
        }
}

// The scan of one LDS-resident tile of up to 16 row blocks with the 16x16x32 shape.  Candidate word `blk` of this LANE (see above: four
// rays x eight rows) goes to bm[blk * STRIDE], bit CLEAR <-> candidate; the return value has bit (n_blocks - 1 - blk) set when that word
// holds any candidate.  Per row block: 4 ds_read_b128, 16 MFMAs, 32 v_alignbit.
// Issue priority of this wave (0..3; `p` must be wave-uniform: s_setprio ignores EXEC).
// The tiled kernels use it to keep the four waves of a SIMD level between two tile barriers.  Instruction issue goes by priority, then
// age (MI355X_MICROARCH.md), so at equal priority the oldest wave runs ahead, the others take the slots it leaves, they finish a tile one
// after the other and the youngest ends it alone on its SIMD at the pace of its own dependencies — while in this vector-ALU-bound scan
// four waves are just enough to fill the issue slots.  With a priority that FALLS with the wave's progress through the tile (four levels:
// quarters of the tile) a wave that is behind always outranks one that is ahead: they arrive at the barrier together.  Measured in one
// process (profiles/r02_ab_progress_prio.log): 100 000 spheres x0.938, 47 106 faces x0.969, Mode R x0.964; a rotating priority
// ((rank + quarter) & 3) instead: x0.961 / x1.019 / x1.009.
__device__ __forceinline__ void set_prio(uint32_t p) {
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#ifndef RT3_PROGRESS_PRIO
#define RT3_PROGRESS_PRIO 1
#endif
template <uint32_t STRIDE = kMB>
__device__ __forceinline__ uint32_t mfma16_scan_tile(const u32x4* s_frag, uint32_t n_blocks, const RayOperands16& R, uint32_t* bm, uint32_t lane,
                                                     uint32_t prio_base = 0, uint32_t prio_shift = 2) {
    uint32_t nz = 0;
    if (n_blocks == 0) return nz;
    const u32x4* fr = s_frag + lane;
    u32x4 a00 = fr[0], a01 = fr[64], a10 = fr[128], a11 = fr[192];  // [h][q]
    auto mm = [](const u32x4& a, const u32x4& b, const f32x4v& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 4) {
        uint32_t* bm0 = bm + b0 * STRIDE;
#if RT3_PROGRESS_PRIO
        set_prio(3u - min(3u, (prio_base + b0) >> prio_shift));     // the further into the tile, the lower: laggards catch up
#endif
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if (b0 + u >= n_blocks) break;
            const f32x4v zero = { 0.0f, 0.0f, 0.0f, 0.0f };
            uint32_t n = 0xFFFFFFFFu;
            // ray groups 3, 2 first, then 1, 0: the sign bits are shifted in from the top, bit 8 G + 4 h + j
#pragma unroll
            for (int gp = 1; gp >= 0; gp--) {
                const int G1 = 2 * gp + 1, G0 = 2 * gp;
                f32x4v d11 = mm(a10, R.b[G1][0], zero), d01 = mm(a00, R.b[G1][0], zero), d10 = mm(a10, R.b[G0][0], zero), d00 = mm(a00, R.b[G0][0], zero);
                d11 = mm(a11, R.b[G1][1], d11); d01 = mm(a01, R.b[G1][1], d01); d10 = mm(a11, R.b[G0][1], d10); d00 = mm(a01, R.b[G0][1], d00);
                if (gp == 0 && b0 + u + 1 < n_blocks) {                 // next block's fragments, in flight during the decode below
                    const u32x4* fn = fr + (size_t)(b0 + u + 1) * 256;
                    a00 = fn[0]; a01 = fn[64]; a10 = fn[128]; a11 = fn[192];
                }
#pragma unroll
                for (int j = 3; j >= 0; j--) n = __builtin_amdgcn_alignbit(n, __float_as_uint(d11[j]), 31);
#pragma unroll
                for (int j = 3; j >= 0; j--) n = __builtin_amdgcn_alignbit(n, __float_as_uint(d01[j]), 31);
#pragma unroll
                for (int j = 3; j >= 0; j--) n = __builtin_amdgcn_alignbit(n, __float_as_uint(d10[j]), 31);
#pragma unroll
                for (int j = 3; j >= 0; j--) n = __builtin_amdgcn_alignbit(n, __float_as_uint(d00[j]), 31);
            }
            bm0[u * STRIDE] = n;
            asm("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(nz) : "v"(n) : "vcc");
        }
    }
    return nz;
}

// ------------------------------------------------------------------------------------------------------
// The K = 32 form for analytic spheres: half the matrix work and half the operand bytes per test, a wider margin
// ------------------------------------------------------------------------------------------------------
// The tiled kernels run at the rate the chip sustains for their instruction mix, so what a test costs is its share of
// MFMAs, LDS operand bytes and decode.  For spheres the exact test is cheap (one 16-byte record, ~45 instructions, 64 lanes at a time through
// the pair list), so the filter may be coarser: only the products HH, MH, HM of the split factors (sphere part x ray part), K = 32 —
// ONE v_mfma_f32_16x16x32_bf16 per 16 x 16 tile, two ds_read_b128 per row block — with eps = 2.2e-4 covering what is dropped (MM, LH, HL:
// 3 * 2^-16 per term; bound in DESIGN.md 5.2c) and the coordinates taken about the spheres' centroid, which keeps |C|^2 + |o|^2, and with it
// the margin, small.  K-slots: A (H x, 1) x (H y, H K) | B (H x, 1) x (M y, M K rounded UP) | D (M x, H E) x (H y, 1) | (M E, L E) x (1, 1).
// Faces take the same form: their exact test is dearer (a 64-byte gather, ~150 instructions) and K = 32 doubles their candidates (20 -> 39 per
// ray cast in the 47 106-face box), but the K = 64 face scan was bound by the matrix pipe itself: x0.82 on that scene, x0.76 on Mode R
// (profiles/r02_ab_face_k32.log).  -DRT3_FACE_K32=0 keeps the K = 64 form of this shape (mfma16_scan_tile) as the A/B reference.
constexpr float kFilterEps32 = 2.2e-4f;
struct RayOperands32 { u32x4 b[4]; };                               // [ray group G]: K-slots 8 g .. 8 g + 7 of ray 16 G + c
__host__ __device__ inline float filter_kj32(double c2, double r2) { return (float)((r2 - c2) + (double)kFilterEps32 * (c2 + r2)); }
__host__ __device__ inline uint32_t bf16_ceil(float x) {           // smallest bf16 >= x (finite x)
    uint32_t u = __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u;     // towards zero
    if (x > 0.0f && __builtin_bit_cast(float, u) < x) u += 0x10000u;
    return u >> 16;
}
// Sphere side: out[g][dword] = K-slots 8 g .. 8 g + 7 of one row; (cx, cy, cz) are already relative to the filter centre.
__host__ __device__ inline void bound_frag32_row(float cx, float cy, float cz, float kj, uint32_t out[4][4]) {
    uint32_t y[9][3], k[2];
    const double x = cx, yy = cy, z = cz;
    split3((float)(x * x), y[0]); split3((float)(yy * yy), y[1]); split3((float)(z * z), y[2]);
    split3((float)(x * yy), y[3]); split3((float)(x * z), y[4]); split3((float)(yy * z), y[5]);
    split3(cx, y[6]); split3(cy, y[7]); split3(cz, y[8]);
    k[0] = bf16_rn(kj);
    k[1] = bf16_ceil(kj - bf16_up(k[0]));                           // H K + M K >= kj: what the dropped low part would add stays on the safe side
    const uint32_t one = 0x3F80u;
    uint32_t slot[32];
    for (int t = 0; t < 9; t++) { slot[t] = y[t][0]; slot[10 + t] = y[t][1]; slot[20 + t] = y[t][0]; }
    slot[9] = k[0]; slot[19] = k[1]; slot[29] = one; slot[30] = one; slot[31] = one;
    for (int g = 0; g < 4; g++)
        for (int d = 0; d < 4; d++) out[g][d] = slot[8 * g + 2 * d] | (slot[8 * g + 2 * d + 1] << 16);
}
__host__ __device__ constexpr size_t frag32_index(uint32_t blk, uint32_t b, uint32_t g) {       // [blk][h][16 g + c]
    return ((size_t)blk * 2 + (b >> 4)) * 64 + 16u * g + (b & 15u);
}
// Ray side (o relative to the filter centre).  16 dwords per ray: (ph0..3, h8) twice, (pm0..3, m8a), (M E, L E); lane (g, c) needs
// dwords 4 g .. 4 g + 3 of the rays of lanes (G, c): the two-stage transpose of build_ray_operands16 on 4-dword pieces.
__device__ __forceinline__ void build_ray_operands32(float ox, float oy, float oz, float dx, float dy, float dz, bool alive, RayOperands32& R) {
    const float od = dotf(ox, oy, oz, dx, dy, dz), oo = dotf(ox, oy, oz, ox, oy, oz);
    float x[9] = { dx * dx, dy * dy, dz * dz, 2.0f * dx * dy, 2.0f * dx * dz, 2.0f * dy * dz,
                   2.0f * (ox - od * dx), 2.0f * (oy - od * dy), 2.0f * (oz - od * dz) };
    float e = alive ? od * od - oo * (1.0f - kFilterEps32) : -3e30f;
    uint32_t ph[4], pm[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { ph[i] = pk_bf16(x[2 * i], x[2 * i + 1]); x[2 * i] -= pk_lo(ph[i]); x[2 * i + 1] -= pk_hi(ph[i]); }
    const uint32_t h8 = pk_bf16(x[8], 1.0f);
    x[8] -= pk_lo(h8);
#pragma unroll
    for (int i = 0; i < 4; i++) pm[i] = pk_bf16(x[2 * i], x[2 * i + 1]);
    const uint32_t m8a = pk_bf16(x[8], e);                          // (M x_8, H E)
    e -= pk_hi(m8a);
    const uint32_t em = pk_bf16(e, 0.0f);                           // M E
    const uint32_t eml = pk_bf16(e, e - pk_lo(em));                 // (M E, L E)
    const uint32_t D[16] = { ph[0], ph[1], ph[2], ph[3], h8, ph[0], ph[1], ph[2], ph[3], h8, pm[0], pm[1], pm[2], pm[3], m8a, eml };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x0, y0, x1, y1, s0e, s0o, s1e, s1o;
        swap32(D[i], D[8 + i], x0, y0);                             // what goes to lane groups 0 | 2
        swap32(D[4 + i], D[12 + i], x1, y1);                        // 1 | 3
        swap16(x0, x1, s0e, s0o);
        swap16(y0, y1, s1e, s1o);
        R.b[0][i] = s0e; R.b[1][i] = s0o; R.b[2][i] = s1e; R.b[3][i] = s1o;
    }
}
// The scan: per row block 2 ds_read_b128, 8 MFMAs and the decode of their 32 results per lane into one candidate word (bit clear = candidate:
// the sign bit of the result).
// RT3_DECODE_FP6 = 1 (r3): ONE v_cvt_scalef32_2xpk16_fp6_f32 reads all 32 accumulators (2 x 16 registers) and writes 32 six-bit floats, each with the
// sign of its input — zeros, denormals, underflow, overflow, infinities and NaNs of both signs included (tools/ubench_fp6_decode.hip checks every class
// on the hardware).  It occupies the vector ALU AND the matrix pipe for 63-65 cycles (8 MFMAs + the conversion: 189 cycles per block with four waves
// per SIMD, the conversion alone 65), where 32 v_alignbit_b32 take 117 in this kernel's instruction mix (158 on their own).  Input a[i] lands in field
// 2 i, b[i] in field 2 i + 1, the sign of field f at bit 6 f + 5 of the 192: in the three words of a half the sign bits sit on DISJOINT odd positions
// (5, 11, .. | 3, 9, .. | 1, 7, ..), so two v_bfi_b32 merge them, and one shift + v_bfi_b32 puts the second half on the even positions: 6 more
// instructions.  Bit p of the word then belongs to field f = ((11 (p >> 1) + 10) & 15) + 16 (1 - (p & 1)) (3 f + 2 = p >> 1 mod 16), that is row half
// h = f & 1, ray group G = f >> 3, accumulator j = (f >> 1) & 3 with a = (d0G), b = (d1G) below: cand_bit().  Measured in one process against the
// v_alignbit build, identical frames (profiles/r03_ab_fp6_decode.log): headline kernel x0.95.  (Merging block b's words behind the MFMAs of block b + 1
// changes nothing — x0.9495 against x0.9516 — the loop is bound by what the units are busy with, not by the wave's own latencies: not kept.)
// RT3_DECODE_FP6 = 0: 32 v_alignbit_b32, bit 8 G + 4 h + j (the words of mfma16_scan_tile).
#ifndef RT3_DECODE_FP6
#define RT3_DECODE_FP6 1
#endif
#ifndef RT3_PRUNE_BEHIND
#define RT3_PRUNE_BEHIND 1
#endif
#ifndef RT3_PRUNE_BEYOND
#define RT3_PRUNE_BEYOND 1
#endif
#ifndef RT3_PRUNE_ROWS
#define RT3_PRUNE_ROWS 1
#endif
#ifndef RT3_DECODE_MIX
#define RT3_DECODE_MIX 0
#endif
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x6v __attribute__((ext_vector_type(6)));
__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) {   // (a & mask) | (b & ~mask): ONE v_bfi_b32, mask in an SGPR
    uint32_t d;                                                     // (left to itself the compiler prefers v_and_b32 with literals + v_or3_b32: 9 instead of 6)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
    return d;
}
// (G, h, j) of candidate bit `bit` of a word: ray group, row half, accumulator index
template <bool FP6>
__device__ __forceinline__ void cand_bit(uint32_t bit, uint32_t& G, uint32_t& h, uint32_t& j) {
    if constexpr (FP6) {
        const uint32_t f = ((11u * (bit >> 1) + 10u) & 15u) | ((~bit & 1u) << 4);
        h = f & 1u; G = f >> 3; j = (f >> 1) & 3u;
    } else { G = bit >> 3; h = (bit >> 2) & 1u; j = bit & 3u; }
}
template <uint32_t STRIDE = kMB, bool PRIO = true>                  // PRIO: progress priority (kernels whose waves meet at tile barriers)
__device__ __forceinline__ uint32_t mfma32k_scan_tile(const u32x4* s_frag, uint32_t n_blocks, const RayOperands32& R, uint32_t* bm, uint32_t lane,
                                                      uint32_t prio_base = 0, uint32_t prio_shift = 3) {
    uint32_t nz = 0;
    if (n_blocks == 0) return nz;
    const u32x4* fr = s_frag + lane;
    u32x4 a0 = fr[0], a1 = fr[64];                                  // rows 0..15 | 16..31 of the block
    auto mm = [](const u32x4& a, const u32x4& b) {
        const f32x4v zero = { 0.0f, 0.0f, 0.0f, 0.0f };
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), zero, 0, 0, 0);
    };
    auto note = [&](uint32_t n, uint32_t* where) {                  // stores a candidate word, counts the words with a candidate
        *where = n;
        asm("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(nz) : "v"(n) : "vcc");
    };
#if RT3_DECODE_FP6
    auto merge = [](const u32x6v& r) {
        const uint32_t t = bfi32(0x28A28A28u, bfi32(0x20820820u, r[0], r[1]), r[2]);      // odd bits: the signs of fields 0..15
        const uint32_t v = bfi32(0x28A28A28u, bfi32(0x20820820u, r[3], r[4]), r[5]);      //           ... of fields 16..31
        return bfi32(0xAAAAAAAAu, t, v >> 1);
    };
#endif
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 4) {
        uint32_t* bm0 = bm + b0 * STRIDE;
#if RT3_PROGRESS_PRIO
        if constexpr (PRIO) set_prio(3u - min(3u, (prio_base + b0) >> prio_shift));     // the further into the tile, the lower: laggards catch up
#endif
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            if (b0 + u >= n_blocks) break;
#if RT3_DECODE_FP6
            const f32x4v d00 = mm(a0, R.b[0]), d01 = mm(a0, R.b[1]), d02 = mm(a0, R.b[2]), d03 = mm(a0, R.b[3]);
            const f32x4v d10 = mm(a1, R.b[0]), d11 = mm(a1, R.b[1]), d12 = mm(a1, R.b[2]), d13 = mm(a1, R.b[3]);
            if (b0 + u + 1 < n_blocks) { const u32x4* fn = fr + (size_t)(b0 + u + 1) * 128; a0 = fn[0]; a1 = fn[64]; }
            const f32x16v lo = { d00[0], d00[1], d00[2], d00[3], d01[0], d01[1], d01[2], d01[3], d02[0], d02[1], d02[2], d02[3], d03[0], d03[1], d03[2], d03[3] };
            const f32x16v hi = { d10[0], d10[1], d10[2], d10[3], d11[0], d11[1], d11[2], d11[3], d12[0], d12[1], d12[2], d12[3], d13[0], d13[1], d13[2], d13[3] };
#if RT3_DECODE_MIX
            if (u % RT3_DECODE_MIX == RT3_DECODE_MIX - 1) {             // (experiment) every RT3_DECODE_MIX-th block on the vector ALU: the same word, bit by bit
                uint32_t n = 0xFFFFFFFFu;
#pragma unroll
                for (int p = 31; p >= 0; p--) {
                    const uint32_t f = ((11u * ((uint32_t)p >> 1) + 10u) & 15u) | ((~(uint32_t)p & 1u) << 4);
                    const uint32_t i = f >> 1;
                    n = __builtin_amdgcn_alignbit(n, __float_as_uint((f & 1u) ? hi[i] : lo[i]), 31);
                }
                note(n, bm0 + u * STRIDE);
            } else
#endif
            note(merge(__builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo, hi, 1.0f)), bm0 + u * STRIDE);
#else
            const f32x4v d13 = mm(a1, R.b[3]), d03 = mm(a0, R.b[3]), d12 = mm(a1, R.b[2]), d02 = mm(a0, R.b[2]);
            const f32x4v d11 = mm(a1, R.b[1]), d01 = mm(a0, R.b[1]), d10 = mm(a1, R.b[0]), d00 = mm(a0, R.b[0]);
            if (b0 + u + 1 < n_blocks) { const u32x4* fn = fr + (size_t)(b0 + u + 1) * 128; a0 = fn[0]; a1 = fn[64]; }
            uint32_t n = 0xFFFFFFFFu;                               // bit 8 G + 4 h + j, shifted in from the top
            const f32x4v* order[8] = { &d13, &d03, &d12, &d02, &d11, &d01, &d10, &d00 };
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int j = 3; j >= 0; j--) n = __builtin_amdgcn_alignbit(n, __float_as_uint((*order[k])[j]), 31);
            note(n, bm0 + u * STRIDE);
#endif
        }
    }
    return nz;
}

// ------------------------------------------------------------------------------------------------------
// Exact tests at full lane utilisation: the candidates of a wave's 64 rays are compacted into a (ray lane, primitive) pair list
// ------------------------------------------------------------------------------------------------------
// A ray has few candidates but they cluster (a ray skimming a tessellated wall meets dozens of bounding spheres in one tile while its
// 63 neighbours meet none), so a loop in which every lane walks its OWN candidates runs at the pace of the unluckiest lane: 5.5 trips per
// 512-row tile on the 47k-face scene for 0.33 candidates per lane.  Instead each lane pushes its candidates (ballot + prefix count)
// into a wave-private list in LDS, and in a TEST PHASE all 64 lanes run one exact test each — lane k takes pair k, fetches the ray of the
// owning lane with ds_bpermute and the primitive from global memory — and a hit is folded into the owning ray's record by
// one 64-bit ds_min: key = (t bits, sphere?, primitive index, sign of t), so the minimum IS the sequential loops' answer (nearest t;
// on equal t faces before spheres, then the lower index) whatever order the pairs are tested in.  t >= 0 always (t_min >= 0 is checked
// by the host), so its bit pattern with the sign cleared orders like the number; the sign of a -0.0 rides in the lowest bit.
// (Running the tests of all 16 waves together, in a phase of the whole workgroup behind a tile's first barrier, instead of whenever a wave
// has 64 pairs, was built and measured neutral — x1.001 / x0.990 / x0.993 on the three tiled workloads — and taken out again.)
constexpr uint32_t kPairCap = 128;                                  // per wave: < 64 left over + one round of <= 64 new pairs
constexpr uint32_t kPairLaneShift = 26;                             // pair = lane << 26 | primitive row (< 2^26)
constexpr unsigned long long kKeyNone = 0x7F800000FFFFFFFFull;      // t = +inf: what a real hit (t < inf) always beats
__device__ __forceinline__ unsigned long long hit_key(float t, uint32_t is_sphere, uint32_t idx) {
    const uint32_t tb = __float_as_uint(t);
    return ((unsigned long long)(tb & 0x7FFFFFFFu) << 32) | (is_sphere << 31) | (idx << 1) | (tb >> 31);
}
__device__ __forceinline__ void key_decode(unsigned long long key, uint32_t& kind, uint32_t& idx, float& t) {
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    kind = key == kKeyNone ? 0u : 1u + (lo >> 31);
    idx = (lo & 0x7FFFFFFFu) >> 1;
    t = __uint_as_float(hi | (lo << 31));
}
// The direct spheres (TraceArgs::direct): every lane tests its own ray; returns the key the ray's record starts from.
template <class SphereAt>
__device__ __forceinline__ unsigned long long direct_tests(const TraceArgs& A, float ox, float oy, float oz, float dx, float dy, float dz, SphereAt&& sphere_at) {
    unsigned long long key = kKeyNone;
    for (uint32_t i = 0; i < A.n_direct; i++) {
        const uint32_t j = A.direct[i];
        float t;
        if (sphere_root(sphere_at(j), ox, oy, oz, dx, dy, dz, A.t_min, t) && t < __builtin_inff()) {
            const unsigned long long k = hit_key(t, 1u, j);
            key = k < key ? k : key;
        }
    }
    return key;
}
// Rays of the wave as the exact tests fetch them from the owning lane.
struct LaneRay { float ox, oy, oz, dx, dy, dz; bool literal; };
template <bool REF>
__device__ __forceinline__ LaneRay fetch_ray(const LaneRay& mine, uint32_t src) {
    const int s = (int)src;
    LaneRay r;
    r.ox = __shfl(mine.ox, s); r.oy = __shfl(mine.oy, s); r.oz = __shfl(mine.oz, s);
    r.dx = __shfl(mine.dx, s); r.dy = __shfl(mine.dy, s); r.dz = __shfl(mine.dz, s);
    r.literal = REF ? __shfl((int)mine.literal, s) != 0 : false;
    return r;
}
// Runs `test(pair, valid)` — ALL lanes call it, `valid` says whether the lane holds a pair — on everything the list holds.
template <class Test>
__device__ __forceinline__ void test_all(uint32_t lane, const uint32_t* pairs, uint32_t& n_pairs, Test&& test) {
    while (n_pairs >= 64u) {
        n_pairs -= 64u;
        test(pairs[n_pairs + lane], true);
        __builtin_amdgcn_wave_barrier();
    }
    if (n_pairs != 0u) {
        const bool valid = lane < n_pairs;
        test(valid ? pairs[lane] : 0u, valid);
        __builtin_amdgcn_wave_barrier();
        n_pairs = 0u;
    }
}
// Pushes this lane's candidates of the tile just scanned (rows row0 ..) into the wave's list and tests 64 pairs whenever that many are
// there; n_pairs (wave-uniform) carries the remainder to the next tile.  The words are mfma16_scan_tile's: bit 8 G + 4 h + j of word blk
// is (ray lane 16 G + c, row 32 blk + 16 h + 4 g + j), with (g, c) the pushing lane's own position.
// The list holds RAW pairs (pushing lane << 26 | row0 + 32 blk + bit) and pair_decode() turns 64 of them into (ray lane << 26 | row) when they are
// tested or set aside — the ten instructions of that arithmetic then run once per 64 pairs instead of once per trip of this loop, which goes at the pace of
// the lane with the most candidates (7 trips for 2.45 candidates per lane on the bench scene).
template <bool FP6>
__device__ __forceinline__ uint32_t pair_decode(uint32_t raw) {
    const uint32_t from = raw >> kPairLaneShift;
    uint32_t G, h, j;
    cand_bit<FP6>(raw & 31u, G, h, j);
    return (((G << 4) + (from & 15u)) << kPairLaneShift) | ((raw & ((1u << kPairLaneShift) - 32u)) + (h << 4) + (from >> 4) * 4u + j);
}
template <uint32_t STRIDE = kMB, bool FP6 = false, class Test>
__device__ __forceinline__ void push_pairs16(uint32_t nz, uint32_t n_blocks, const uint32_t* bm, uint32_t row0, uint32_t lane, uint32_t* pairs,
                                             uint32_t& n_pairs, Test&& test) {
    CandIter it = { nz, 0u, 0u, n_blocks };
    for (;;) {
        uint32_t pos = 0;                                           // 32 blk + bit
        const bool have = cand_next<STRIDE>(it, bm, pos);
        const unsigned long long m = __ballot(have);
        if (m == 0ull) break;
        const uint32_t pair = (lane << kPairLaneShift) | (row0 + pos);
        if (have) pairs[n_pairs + prefix_count(m)] = pair;
        n_pairs += (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (n_pairs >= 64u) {
            n_pairs -= 64u;
            const uint32_t v = pairs[n_pairs + lane];
            test(pair_decode<FP6>(v), true);
            __builtin_amdgcn_wave_barrier();
        }
    }
}
// What is left in a RAW list (see test_all).
template <bool FP6, class Test>
__device__ __forceinline__ void test_all_raw(uint32_t lane, const uint32_t* pairs, uint32_t& n_pairs, Test&& test) {
    if (n_pairs != 0u) {
        const bool valid = lane < n_pairs;
        test(valid ? pair_decode<FP6>(pairs[lane]) : 0u, valid);
        __builtin_amdgcn_wave_barrier();
        n_pairs = 0u;
    }
}

// Two-level filter: a row of the scan is a GROUP of primitives, a candidate (ray lane, group row) pair expands into the members'
// tests.  The pairs of a pass are not tested between two tile barriers — there the 16 waves of the workgroup would wait for whichever has the most
// candidates in THIS tile, and a batch of member tests is mostly latency — but set aside, 64 at a time (one coalesced 256-byte store), in the
// wave's own strip of global memory and tested when the pass has seen every tile: one run of full 64-lane batches with no barrier in it
// (wave time at the tile barriers on the 100 000-sphere scene: 43 % -> 13 %).  In a batch every LANE owns one pair: it fetches the ray of the
// owning lane once (ds_bpermute) and walks the group's members itself — their records are the lane's own 128 contiguous bytes — so the
// per-pair overhead is paid once per GROUP member tests.  A strip holds kStripPairs pairs; a wave that fills it drains it on the spot.
constexpr uint32_t kStripPairs = 8192;                              // per wave: 32 KiB (x 16 waves x 256 workgroups = 128 MiB)
template <uint32_t STRIDE = kMB, bool FP6 = false, class Spill>
__device__ __forceinline__ void push_pairs16_spill(uint32_t nz, uint32_t n_blocks, const uint32_t* bm, uint32_t row0, uint32_t lane, uint32_t* pairs,
                                                   uint32_t& n_pairs, Spill&& spill) {
    push_pairs16<STRIDE, FP6>(nz, n_blocks, bm, row0, lane, pairs, n_pairs, [&](uint32_t v, bool) { spill(v); });   // (v: decoded by push_pairs16)
}
// Runs `group(pair, valid, part)` — ALL lanes call it — on the n_strip pairs of the wave's strip.  LPP lanes share a pair: lane k of a batch takes
// pair k / LPP and the members [part, part + 1) * GROUP / LPP of its group, part = k % LPP.  What LPP trades (100 000-sphere scene, per pair):
//   1 lane  per pair: the ray is fetched once per 8 member tests (3 wave-instructions per pair), but every load instruction touches 64 different
//                     cache lines (8 tag look-ups per pair): the texture path, not the ALU, sets the pace (measured: no faster than 8 lanes)
//   8 lanes per pair: one look-up per pair (the 8 lanes read one 128-byte line), 8 wave-instructions per pair: the vector ALU sets the pace
//   2 / 4 lanes per pair: 4 / 2 look-ups and 3.2 / 3.7 wave-instructions per pair
// Measured in one process (profiles/r03_ab_lanes_per_pair.log), 100 000 spheres | 47 106 faces: 1 lane 46.6 | 21.7 ms, 2 lanes 44.5 | 19.3 (the default,
// RT3_LANES_PER_PAIR), 4 lanes 46.7 | 20.1, 8 lanes 53.9 | 21.0.
#ifndef RT3_LANES_PER_PAIR
#define RT3_LANES_PER_PAIR 2
#endif
constexpr uint32_t kLanesPerPair = RT3_LANES_PER_PAIR;
template <uint32_t LPP, class GroupFn>
__device__ __forceinline__ void drain_strip(uint32_t lane, const uint32_t* strip, uint32_t n_strip, GroupFn&& group) {
    constexpr uint32_t PER = 64u / LPP;                             // pairs per batch
    if (n_strip == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");          // the strip was written by other lanes of this wave
    uint32_t next = strip[min(lane, n_strip - 1u)];
    for (uint32_t base = 0; base < n_strip; base += 64u) {
        const uint32_t chunk = next;
        if (base + 64u < n_strip) next = strip[min(base + 64u + lane, n_strip - 1u)];      // in flight during this chunk's batches
#pragma unroll
        for (uint32_t k0 = 0; k0 < 64u; k0 += PER) {
            if (base + k0 >= n_strip) break;                        // (wave-uniform)
            const uint32_t k = k0 + lane / LPP;
            group(LPP == 1 ? chunk : (uint32_t)__shfl((int)chunk, (int)k), base + k < n_strip, lane % LPP);
            __builtin_amdgcn_wave_barrier();
        }
    }
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
    bool prof_dry_seen = false;
    if (lane == 0) { const unsigned long long t0 = wall_clock64(); atomicMin(A.cast_counter + 8, t0); atomicMax(A.cast_counter + 11, t0); }
#endif

#ifdef RT3_PROFILE_PHASES                                                       // wave time (s_memtime) per part of a trip: tools/README.md
    unsigned long long ph_refill = 0, ph_operands = 0, ph_scan = 0, ph_flush = 0, ph_shade = 0, ph_mark = clock64();
#define RT3_SPHASE(acc) { const unsigned long long now_ = clock64(); acc += now_ - ph_mark; ph_mark = now_; }
#else
#define RT3_SPHASE(acc)
#endif
    for (;;) {
        refill_from_stock(A, lane, alive, P, Q, chunk_next, chunk_end, exhausted);
        RT3_SPHASE(ph_refill)
#ifdef RT3_PROFILE
        if (exhausted && !prof_dry_seen) {
            prof_dry_seen = true;
            if (lane == 0) { const unsigned long long t = wall_clock64(); atomicMin(A.cast_counter + 9, t); atomicMax(A.cast_counter + 10, t); }
        }
#endif
        const unsigned long long live = __ballot(alive);
        if (live == 0ull) break;
        casts += (unsigned long long)__popcll(live);
        iters++;
        const float ox = P.ox, oy = P.oy, oz = P.oz, dx = P.dx, dy = P.dy, dz = P.dz;
        RayOperands R;
        build_ray_operands(ox - A.fcx, oy - A.fcy, oz - A.fcz, dx, dy, dz, alive, R);      // the fragments are about (fcx, fcy, fcz)
        RT3_SPHASE(ph_operands)

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
            const float t_near = h - sq;
            const float t = t_near > A.t_min ? t_near : h + sq;
            // the candidates of a ray come in ascending sphere index (mfma_flush), so "the lower index wins ties" is strict <
            const bool better = (t > A.t_min) & (t < tbest);
            tbest = better ? t : tbest;
            ibest = better ? j : ibest;
        };
        const uint32_t nz = mfma_scan_tile(s_frag, n_blocks, R, s_bm + tid, lane);
        RT3_SPHASE(ph_scan)
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
        {                                                                       // the direct spheres (not in the filter): (t, index) order as the pair-list kernels' key
            uint32_t dk, di; float dt;
            key_decode(direct_tests(A, ox, oy, oz, dx, dy, dz, [&](uint32_t j) { return s_sph[j]; }), dk, di, dt);
            const bool better = dk != 0u && (dt < tbest || (dt == tbest && di < ibest));
            tbest = better ? dt : tbest;
            ibest = better ? di : ibest;
        }
        RT3_SPHASE(ph_flush)
        kind = tbest < __builtin_inff() ? 2u : 0u;
        shade_lane<false, true>(A, P, alive, kind, ibest, tbest, s_sph, s_invr, s_mat, s_kind);
        RT3_SPHASE(ph_shade)
    }
#ifdef RT3_PROFILE_PHASES
    if (lane == 0) {
        atomicAdd(A.cast_counter + 11, ph_refill); atomicAdd(A.cast_counter + 12, ph_operands); atomicAdd(A.cast_counter + 13, ph_scan);
        atomicAdd(A.cast_counter + 14, ph_flush); atomicAdd(A.cast_counter + 15, ph_shade);
    }
#endif
#undef RT3_SPHASE
    if (lane == 0 && casts != 0) { atomicAdd(A.cast_counter, casts); atomicAdd(A.cast_counter + 1, iters * n_blocks * 8ull); }
#ifdef RT3_PROFILE
    if (lane == 0) {
        atomicAdd(A.cast_counter + 2, prof_flush_iters); atomicAdd(A.cast_counter + 3, prof_cands); atomicAdd(A.cast_counter + 4, iters); atomicAdd(A.cast_counter + 5, prof_refills);
        const unsigned long long now = wall_clock64();              // 100 MHz: when the first and the last wave ended, when the queue ran dry
        atomicMin(A.cast_counter + 6, now); atomicMax(A.cast_counter + 7, now);
    }
#endif
}

// The same scene class (<= 512 spheres, everything in LDS, no barrier after the prologue) on the K = 32 form: ONE v_mfma_f32_16x16x32_bf16
// per 16 x 16 tests, 64 B of fragments per sphere, candidates through the pair list, exact tests on the LDS mirror of the spheres.
// The vector-ALU instruction count per ray cast is that of k_trace_mfma (the K = 32 margin brings more candidates, the pair list tests them
// at full lane utilisation, the ray operands cost half), but the matrix pipe does half the work and the chip, which throttles under
// k_trace_mfma's load (2.0-2.2 GHz), holds 2.3-2.4 GHz here — and a kernel bound by vector-ALU issue runs at the clock.
__global__ __launch_bounds__(kMB) void k_trace_mfma32(const TraceArgs A, const u32x4* __restrict__ frags, uint32_t n_blocks) {
    extern __shared__ u32x4 lds_dyn[];
    u32x4* s_frag = lds_dyn;                                                   // [n_blocks][2][64]
    float4* s_sph = reinterpret_cast<float4*>(s_frag + (size_t)n_blocks * 128);   // [n_blocks * 32] (cx, cy, cz, r^2) for the exact test
    float4* s_mat = s_sph + (size_t)n_blocks * 32;                               // materials, kinds, 1/r: read at every hit
    float* s_invr = reinterpret_cast<float*>(s_mat + (size_t)n_blocks * 32);
    uint32_t* s_kind = reinterpret_cast<uint32_t*>(s_invr + (size_t)n_blocks * 32);
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_kind + (size_t)n_blocks * 32); // [16][kMB] candidate words
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_bm + 16 * kMB);   // [kMB] nearest hit of every lane's ray
    uint32_t* s_pairs = reinterpret_cast<uint32_t*>(s_key + kMB);              // [16 waves][kPairCap]
    const uint32_t tid = threadIdx.x, lane = lane_id();
    uint32_t* pairs = s_pairs + (tid / 64u) * kPairCap;
    unsigned long long* keys = s_key + (tid & ~63u);
    for (uint32_t k = tid; k < n_blocks * 128; k += kMB) s_frag[k] = frags[k];
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
    unsigned long long casts = 0, iters = 0, exact = 0;
#ifdef RT3_PROFILE_PHASES                                                       // wave time (s_memtime) per part of a trip: tools/README.md
    unsigned long long ph_refill = 0, ph_operands = 0, ph_scan = 0, ph_flush = 0, ph_shade = 0, ph_mark = clock64();
#define RT3_SPHASE(acc) { const unsigned long long now_ = clock64(); acc += now_ - ph_mark; ph_mark = now_; }
#else
#define RT3_SPHASE(acc)
#endif

    for (;;) {
        refill_from_stock(A, lane, alive, P, Q, chunk_next, chunk_end, exhausted);
        RT3_SPHASE(ph_refill)
        const unsigned long long live = __ballot(alive);
        if (live == 0ull) break;
        casts += (unsigned long long)__popcll(live);
        iters++;
        const LaneRay ray = { P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, false };
        RayOperands32 R;
        build_ray_operands32(ray.ox - A.fcx, ray.oy - A.fcy, ray.oz - A.fcz, ray.dx, ray.dy, ray.dz, alive, R);
        keys[lane] = direct_tests(A, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, [&](uint32_t j) { return s_sph[j]; });
        uint32_t n_pairs = 0;
        auto test = [&](uint32_t pair, bool valid) {
            const uint32_t src = pair >> kPairLaneShift, j = pair & ((1u << kPairLaneShift) - 1u);
            const LaneRay r = fetch_ray<false>(ray, src);
            exact += (unsigned long long)__popcll(__ballot(valid));
            if (!valid) return;                                                 // (padding rows are never candidates; s_sph covers every row)
            float t;
            if (sphere_root(s_sph[j], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, A.t_min, t) && t < __builtin_inff()) atomicMin(&keys[src], hit_key(t, 1u, j));
        };
        RT3_SPHASE(ph_operands)
        const uint32_t nz = mfma32k_scan_tile<kMB, false>(s_frag, n_blocks, R, s_bm + tid, lane);
        RT3_SPHASE(ph_scan)
        push_pairs16<kMB, RT3_DECODE_FP6 != 0>(nz, n_blocks, s_bm + tid, 0u, lane, pairs, n_pairs, test);
        test_all_raw<RT3_DECODE_FP6 != 0>(lane, pairs, n_pairs, test);
        RT3_SPHASE(ph_flush)
        __builtin_amdgcn_wave_barrier();
        uint32_t kind, ibest;
        float tbest;
        key_decode(keys[lane], kind, ibest, tbest);
        shade_lane<false, true>(A, P, alive, kind, ibest, tbest, s_sph, s_invr, s_mat, s_kind);
        RT3_SPHASE(ph_shade)
    }
#ifdef RT3_PROFILE_PHASES
    if (lane == 0) {
        atomicAdd(A.cast_counter + 11, ph_refill); atomicAdd(A.cast_counter + 12, ph_operands); atomicAdd(A.cast_counter + 13, ph_scan);
        atomicAdd(A.cast_counter + 14, ph_flush); atomicAdd(A.cast_counter + 15, ph_shade);
    }
#endif
#undef RT3_SPHASE
    if (lane == 0 && casts != 0) { atomicAdd(A.cast_counter, casts); atomicAdd(A.cast_counter + 1, iters * n_blocks * 8ull); atomicAdd(A.cast_counter + 2, exact); }
}

// Any scene: faces (through their bounding spheres) and spheres, streamed through LDS in 64-KiB tiles of 1024 rows.  The 16 waves of the
// workgroup move through the tiles together (two barriers per tile); candidates go through the pair list above, the exact tests
// gather their records from global memory, all 64 lanes at a time.
// REF: RT3_FLAG_REFERENCE_PRIMARY (camera at the origin, no lens: checked by the host).
#ifndef RT3_TILED_TB
#define RT3_TILED_TB 1024
#endif
#ifndef RT3_TILE_LOADS
#define RT3_TILE_LOADS 4
#endif
#ifndef RT3_BM_BLOCKS
#define RT3_BM_BLOCKS 12
#endif
constexpr uint32_t kTB = RT3_TILED_TB;                              // threads per workgroup of k_trace_mfma_tiled
constexpr uint32_t kTileLoads = RT3_TILE_LOADS;                     // 16-byte vectors per thread and tile: the fragment tile is kTB * kTileLoads * 16 bytes
constexpr uint32_t kBmBlocks = RT3_BM_BLOCKS;                       // row blocks scanned before their candidate words are pushed (words: kBmBlocks * kTB * 4 bytes)
constexpr size_t kTiledLdsBytes = (size_t)16 * 4096 + kBitmapBytes + (size_t)kMB * 8 + (size_t)(kMB / 64) * kPairCap * 4;     // k_mode_r_mfma
constexpr size_t kTraceTiledLdsBytes = (size_t)kTB * (kTileLoads * 16 + kBmBlocks * 4 + 8) + (size_t)(kTB / 64) * kPairCap * 4 * 3;      // three pair lists per wave: 160 KiB in all
// One tile of fragments through the workgroup: [barrier] loads -> LDS stores [barrier].
template <uint32_t TB = kMB, uint32_t LOADS = 4>
__device__ __forceinline__ void fill_tile(u32x4* s_frag, const u32x4* __restrict__ src, uint32_t n_vec, uint32_t tid) {
    __syncthreads();                                                // every wave is done with the previous tile
    u32x4 v[LOADS];
#pragma unroll
    for (uint32_t i = 0; i < LOADS; i++) v[i] = src[min(tid + i * TB, n_vec - 1u)];   // unconditional: all loads in flight, no branches
#pragma unroll
    for (uint32_t i = 0; i < LOADS; i++) { const uint32_t k = tid + i * TB; if (k < n_vec) s_frag[k] = v[i]; }
    __syncthreads();
}
// GT / GS: group size of the faces' / spheres' rows (1: one row per primitive, the flat filter; > 1: the two-level filter of DESIGN.md 5.2e —
// a row is the bounding sphere of G primitives, a candidate row expands into G member tests: for spheres the exact test itself, for faces
// first the f32 test of the member's own bounding sphere (the vector-ALU kernels' scan arithmetic, 16-byte record) whose survivors are
// compacted into a second pair list and get the reference's plane + three-edge test, 64 at a time).  Both must divide 64.
#ifndef RT3_GROUP_TRI
#define RT3_GROUP_TRI 8
#endif
#ifndef RT3_GROUP_SPH
#define RT3_GROUP_SPH 8
#endif
#ifndef RT3_SUPER
#define RT3_SUPER 8                                                 // leaf groups per row of the matrix filter (1: two levels, rows = leaf groups)
#endif
constexpr uint32_t kGroupTri = RT3_GROUP_TRI, kGroupSph = RT3_GROUP_SPH, kSuper = RT3_SUPER;
// SUP > 1 (with GT, GS > 1): three levels — a row bounds SUP consecutive leaf groups; a candidate row's ray is first tested, in f32, against the SUP
// leaves' bounding spheres (A.*_leaf; scan_tile's arithmetic with a margin of 1e-4 c, which covers what the exact sphere test's own rounding lets
// through: DESIGN.md 5.2e), the surviving (ray lane, leaf) pairs go through a list of their own to the members' tests described above.
// RES (with three levels): ALL rows of the scene stay in LDS — 64 B per row of 64 primitives: 100 KiB for the 100 000-sphere scene — loaded once; no tile
// fill and no barrier after the prologue, the 16 waves of the workgroup run free as k_trace_mfma32's do (each leaves when its own paths are done).
// The host picks it when the rows fit (kResidentBlocks); the candidate words then cover kBmBlocksRes row blocks per push.
constexpr uint32_t kBmBlocksRes = 4;
constexpr uint32_t kResidentBlocks = (160u * 1024u - kTB * 8u - (kTB / 64u) * kPairCap * 4u * 3u - kBmBlocksRes * kTB * 4u) / 2048u - 1u;   // row blocks of 2 KiB: 55
// (one block of headroom: a kernel with any static LDS beside the dynamic request — __syncthreads_or's word, say — is refused at exactly 160 KiB;
// this variant has none and did launch with 56, profiles/README.md)
template <bool HAS_TRI, bool HAS_SPH, bool REF, uint32_t GT = 1, uint32_t GS = 1, uint32_t SUP = 1, bool RES = false>
__global__ __launch_bounds__(kTB) void k_trace_mfma_tiled(const TraceArgs A, const u32x4* __restrict__ tri_frags, const u32x4* __restrict__ sph_frags) {
    static_assert(64 % GT == 0 && 64 % GS == 0, "group sizes must divide the wave");
    static_assert(!RES || (SUP > 1 && RT3_FACE_K32), "resident rows: three-level filter only");
    constexpr uint32_t BM = RES ? kBmBlocksRes : kBmBlocks;                     // row blocks per push
    extern __shared__ u32x4 lds_dyn[];
    const uint32_t res_tri_blocks = HAS_TRI ? (A.n_tri_rows + 31u) / 32u : 0u, res_sph_blocks = HAS_SPH ? (A.n_sph_rows + 31u) / 32u : 0u;
    u32x4* s_frag = lds_dyn;                                                   // tiles: kTB * kTileLoads vectors ([16][4][64] at 1024 threads x 4); RES: every row block
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_frag + (RES ? (size_t)(res_tri_blocks + res_sph_blocks) * 128u : (size_t)kTB * kTileLoads));   // [BM][kTB] candidate words
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_bm + BM * kTB);   // [kTB] nearest hit of every lane's ray
    uint32_t* s_pairs = reinterpret_cast<uint32_t*>(s_key + kTB);              // [waves][2][kPairCap]
    const uint32_t tid = threadIdx.x, lane = lane_id();
    uint32_t* pairs = s_pairs + (tid / 64u) * (3 * kPairCap);                  // (ray lane, row) pairs from the scan
    uint32_t* fpairs = pairs + kPairCap;                                       // GT > 1: (ray lane, face) pairs that passed their own bound
    uint32_t* lpairs = fpairs + kPairCap;                                      // SUP > 1: (ray lane, leaf group) pairs that passed the leaf's bound
    unsigned long long* keys = s_key + (tid & ~63u);                           // this wave's 64 records
    uint32_t* strip = A.pair_strips + ((size_t)blockIdx.x * (kTB / 64u) + tid / 64u) * kStripPairs;     // deferred member tests (GT, GS > 1)
    if constexpr (RES) {                                                        // the rows: faces' first, then the spheres'
        for (uint32_t k = tid; k < res_tri_blocks * 128u; k += kTB) s_frag[k] = tri_frags[k];
        for (uint32_t k = tid; k < res_sph_blocks * 128u; k += kTB) s_frag[res_tri_blocks * 128u + k] = sph_frags[k];
        __syncthreads();                                                        // the only barrier
    }

    Path P;
    P.ox = P.oy = P.oz = 0.0f; P.dx = P.dy = 0.0f; P.dz = 1.0f;
    P.tr = P.tg = P.tb = 0.0f; P.lr = P.lg = P.lb = 0.0f; P.slot = 0; P.base = 0; P.depth = 0;
    bool alive = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    unsigned long long casts = 0, mfmas = 0, exact = 0, bound_tests = 0;
#ifdef RT3_PROFILE                                                              // wall-clock ticks (100 MHz) of this wave per phase
    unsigned long long pt_fill = 0, pt_scan = 0, pt_push = 0, pt_test = 0, pt_rest = 0, pt_mark = wall_clock64();
#define RT3_PHASE(acc) { const unsigned long long now_ = wall_clock64(); acc += now_ - pt_mark; pt_mark = now_; }
#else
#define RT3_PHASE(acc)
#endif

    for (;;) {
        // (no ray stock here: a ray cast costs at least one tile scan, start_path is noise beside it, and the stock's 8 registers are needed)
        refill_lanes<REF>(A, lane, alive, P, chunk_next, chunk_end, exhausted);
        const unsigned long long live = __ballot(alive);
        if constexpr (RES) { if (live == 0ull) break; }                           // every wave for itself
        else if (!__syncthreads_or(live != 0ull ? 1 : 0)) break;                 // tiles: the workgroup ends together
        casts += (unsigned long long)__popcll(live);
        LaneRay ray = { P.ox, P.oy, P.oz, P.dx, P.dy, P.dz, REF && P.depth == 0 };
        float ux = ray.dx, uy = ray.dy, uz = ray.dz;                            // what the filter sees: always a unit direction
        if (REF && ray.literal) { const float inv = 1.0f / __builtin_sqrtf(dot3(ux, uy, uz, ux, uy, uz)); ux = ux * inv; uy = uy * inv; uz = uz * inv; }
        RayOperands16 R;                                                        // (faces in the K = 64 form: -DRT3_FACE_K32=0 only)
        RayOperands32 R32;                                                      // spheres: K = 32, coordinates about the filter centre
#if RT3_FACE_K32
        if (HAS_TRI) build_ray_operands32(ray.ox - A.tcx, ray.oy - A.tcy, ray.oz - A.tcz, ux, uy, uz, alive, R32);
#else
        if (HAS_TRI) build_ray_operands16(ray.ox - A.tcx, ray.oy - A.tcy, ray.oz - A.tcz, ux, uy, uz, alive, R);
#endif
        if (HAS_SPH && !HAS_TRI) build_ray_operands32(ray.ox - A.fcx, ray.oy - A.fcy, ray.oz - A.fcz, ux, uy, uz, alive, R32);
        keys[lane] = HAS_SPH ? direct_tests(A, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz, [&](uint32_t j) { return A.sph[j]; }) : kKeyNone;
        uint32_t n_pairs = 0, n_fpairs = 0, n_lpairs = 0, n_strip = 0;
        // SUP > 1: the middle level.  A batch of (ray lane, row) pairs, LPP lanes per pair, each lane SUP / LPP leaf bounds (its own contiguous bytes);
        // survivors are appended to lpairs and handed, 64 / LPP at a time, to `leaf_fn` (the members' tests of the two-level filter).
        auto super_stage = [&](const float4* __restrict__ leaf_bounds, uint32_t n_rows, auto lpp_tag, auto beyond_tag, auto&& leaf_fn) {
            return [&, leaf_bounds, n_rows](uint32_t pair, bool valid, uint32_t part) {
                constexpr bool BEYOND = decltype(beyond_tag)::value && !REF && RT3_PRUNE_BEYOND;   // (faces: x0.97; spheres: x1.01, off)
                constexpr uint32_t LPPL = decltype(lpp_tag)::value;                 // lanes per pair of the LEAF stage: its batches take 64 / LPPL pairs
                constexpr uint32_t LPPS = kLanesPerPair < SUP ? kLanesPerPair : SUP, MPLS = SUP / LPPS;
                const uint32_t src = pair >> kPairLaneShift, g = pair & ((1u << kPairLaneShift) - 1u);
                const int sl = (int)src;
                const float sox = __shfl(ray.ox, sl), soy = __shfl(ray.oy, sl), soz = __shfl(ray.oz, sl);
                const float sdx = __shfl(ux, sl), sdy = __shfl(uy, sl), sdz = __shfl(uz, sl);
                bound_tests += (unsigned long long)__popcll(__ballot(valid)) * MPLS;
                const bool ok = valid && g < n_rows;
                // the ray's best hit so far (its record in LDS; unit directions only: a literal reference ray counts t in its own units)
                const float tb = BEYOND ? __uint_as_float(reinterpret_cast<const uint32_t*>(keys)[2u * src + 1u]) : __builtin_inff();
                const uint32_t l0 = ok ? g * SUP + part * MPLS : 0u;
                float4 b[MPLS];
#pragma unroll
                for (uint32_t m = 0; m < MPLS; m++) b[m] = leaf_bounds[l0 + m];
#pragma unroll
                for (uint32_t m = 0; m < MPLS; m++) {
                    const float cx = b[m].x - sox, cy = b[m].y - soy, cz = b[m].z - soz;
                    const float h = fma_(cz, sdz, fma_(cy, sdy, cx * sdx));
                    const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -b[m].w)));
                    const float disc = fma_(1e-4f, c, fma_(h, h, -c));
#if RT3_PRUNE_BEHIND
                    // a leaf whose centre lies behind an origin that is outside it meets only the backward half of the line (both roots < 0)
                    bool behind = (h < 0.0f) & (c > 2e-3f * b[m].w);
                    // ... or that the ray only enters beyond its best hit so far: the point at t_best lies before the closest approach and outside
                    // the leaf by a margin (|p(t_best) - C|^2 - R^2 = t_best^2 - 2 h t_best + c > 10^-3 |C - o|^2).  What it saves depends on the
                    // order the pairs come in (47 106 faces: 7 % of the exact tests; 100 000 spheres: 11 % of the member tests, not worth 6 instructions)
                    if constexpr (BEYOND) behind |= (tb < h) & (fma_(tb, fma_(-2.0f, h, tb), c) > 1e-3f * (c + b[m].w));
                    const bool keep = ok && (((__float_as_uint(disc) >> 31) == 0u && !behind) || b[m].w >= 3e38f);
#else
                    const bool keep = ok && ((__float_as_uint(disc) >> 31) == 0u || b[m].w >= 3e38f);   // (a leaf with an unbounded member: always)
#endif
                    const unsigned long long km = __ballot(keep);
                    if (km == 0ull) continue;
                    if (keep) lpairs[n_lpairs + prefix_count(km)] = (src << kPairLaneShift) | (l0 + m);
                    n_lpairs += (uint32_t)__popcll(km);
                    __builtin_amdgcn_wave_barrier();
                    while (n_lpairs >= 64u / LPPL) {
                        n_lpairs -= 64u / LPPL;
                        leaf_fn(lpairs[n_lpairs + lane / LPPL], true, lane % LPPL);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            };
        };
        auto flush_leaves = [&](auto lpp_tag, auto&& leaf_fn) {                    // what is left in lpairs at the end of a pass
            constexpr uint32_t LPPL = decltype(lpp_tag)::value;
            if (n_lpairs != 0u) {
                const bool valid = lane / LPPL < n_lpairs;
                leaf_fn(valid ? lpairs[lane / LPPL] : 0u, valid, lane % LPPL);
                __builtin_amdgcn_wave_barrier();
                n_lpairs = 0u;
            }
        };

        // Sets (ray lane, row) pairs aside in the wave's strip — ALL lanes call it — after one more look at the ROW's own bound (rowb; three-level filter):
        // the matrix filter tested the line, a row that lies behind the ray's origin (centre behind, origin outside: both roots negative) or, for
        // faces, that the ray enters only beyond its best hit so far costs eight leaf tests for nothing.  One lane per pair: 64 at a time.
        auto set_aside = [&](uint32_t v, bool valid, const float4* __restrict__ rowb, auto beyond_tag) {
            bool keep = valid;
#if RT3_PRUNE_ROWS
            if (rowb != nullptr) {
                const uint32_t src = v >> kPairLaneShift, g = v & ((1u << kPairLaneShift) - 1u);
                const int sl = (int)src;
                const float sox = __shfl(ray.ox, sl), soy = __shfl(ray.oy, sl), soz = __shfl(ray.oz, sl);
                const float sdx = __shfl(ux, sl), sdy = __shfl(uy, sl), sdz = __shfl(uz, sl);
                const float4 b = rowb[valid ? g : 0u];
                const float cx = b.x - sox, cy = b.y - soy, cz = b.z - soz;
                const float h = fma_(cz, sdz, fma_(cy, sdy, cx * sdx));
                const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -b.w)));
                bool drop = (h < 0.0f) & (c > 2e-3f * b.w);
                if constexpr (decltype(beyond_tag)::value && !REF && RT3_PRUNE_BEYOND) {
                    const float tb = __uint_as_float(reinterpret_cast<const uint32_t*>(keys)[2u * src + 1u]);
                    drop |= (tb < h) & (fma_(tb, fma_(-2.0f, h, tb), c) > 1e-3f * (c + b.w));
                }
                keep = valid && !drop;
                bound_tests += (unsigned long long)__popcll(__ballot(valid));
            }
#endif
            const unsigned long long km = __ballot(keep);
            if (keep) strip[n_strip + prefix_count(km)] = v;
            n_strip += (uint32_t)__popcll(km);
        };

        auto pass = [&](const u32x4* __restrict__ frags, uint32_t n_rows, const float4* __restrict__ rowb, auto beyond_tag, auto k32, auto group_tag, auto&& test, auto&& finish) {
            constexpr bool K32 = decltype(k32)::value;                          // spheres: 2 operand fragments (2 KiB) per row block, else 4
            constexpr bool GROUPED = decltype(group_tag)::value > 1;            // `test` is then the deferred (pair, valid, part) form
            constexpr uint32_t LPPX = kLanesPerPair < decltype(group_tag)::value ? kLanesPerPair : decltype(group_tag)::value;
            constexpr uint32_t kVec = K32 ? 128u : 256u;
            constexpr uint32_t kTile = kTB * kTileLoads / kVec;                 // row blocks per tile (64 KiB at 1024 threads x 4 loads): 32 | 16
            const uint32_t total_blocks = (n_rows + 31u) / 32u;
            for (uint32_t b0 = 0; b0 < total_blocks; b0 += RES ? total_blocks : kTile) {
                const uint32_t nb = RES ? total_blocks : min(kTile, total_blocks - b0);
                const u32x4* tile = s_frag;
                if constexpr (RES) tile = s_frag + (frags == tri_frags ? 0u : res_tri_blocks * 128u);       // this pass's rows, resident
                else {
                    RT3_PHASE(pt_rest)
                    fill_tile<kTB, kTileLoads>(s_frag, frags + (size_t)b0 * kVec, nb * kVec, tid);
                    RT3_PHASE(pt_fill)
                    if (live == 0ull) continue;                                 // a wave without rays (the tail of a launch) only keeps the barriers
                }
                for (uint32_t h0 = 0; h0 < nb; h0 += BM) {                      // the candidate words hold BM row blocks: scan and push in parts
                    const uint32_t hb = min(BM, nb - h0);
                    uint32_t nz;
                    if constexpr (K32) nz = mfma32k_scan_tile<kTB, !RES>(tile + (size_t)h0 * kVec, hb, R32, s_bm + tid, lane, h0, 3);
                    else nz = mfma16_scan_tile<kTB>(tile + (size_t)h0 * kVec, hb, R, s_bm + tid, lane);
                    RT3_PHASE(pt_scan)
                    if constexpr (GROUPED) {
                        push_pairs16_spill<kTB, K32 && RT3_DECODE_FP6 != 0>(nz, hb, s_bm + tid, (b0 + h0) * 32u, lane, pairs, n_pairs, [&](uint32_t v) {
                            set_aside(v, true, rowb, beyond_tag);
                            if (n_strip + 64u > kStripPairs) { drain_strip<LPPX>(lane, strip, n_strip, test); n_strip = 0u; }    // (rare: 128 candidate rows per ray)
                        });
                    } else push_pairs16<kTB, K32 && RT3_DECODE_FP6 != 0>(nz, hb, s_bm + tid, (b0 + h0) * 32u, lane, pairs, n_pairs, test);
                    RT3_PHASE(pt_push)
                    mfmas += hb * (K32 ? 8ull : 16ull);
                }
            }
            finish();                                                           // what is left at the end of the pass
            RT3_PHASE(pt_test)
        };
        if (HAS_TRI) {
            // the reference's plane + three-edge test of (ray lane, face) pairs.  GT == 1: the pair names the face; GT > 1: its POSITION in group order —
            // the record is read from A.tri_rec (a leaf's 8 faces: 512 contiguous bytes) and the face's own index, which the key carries, only on a hit
            auto face_test = [&](uint32_t pair, bool valid) {
                const uint32_t src = pair >> kPairLaneShift, j = pair & ((1u << kPairLaneShift) - 1u);
                const LaneRay r = fetch_ray<REF>(ray, src);
                exact += (unsigned long long)__popcll(__ballot(valid));
                if (!valid || (GT == 1 && j >= A.n_tri)) return;
                const float4* f = (GT == 1 ? A.tri : A.tri_rec) + (size_t)j * 4;
                const float4 n = f[0], p1 = f[1], p2 = f[2], p3 = f[3];           // the whole 64-byte record at once: ONE trip to L2, not two
                const float t_hi = __uint_as_float((uint32_t)(keys[src] >> 32));   // the ray's best t so far: farther faces need no edge tests
                float t;
                if (!face_t(n, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.literal, t) || !(t >= A.t_min && t <= t_hi)) return;
                if (!face_inside(n, p1, p2, p3, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, t)) return;
                if (t < __builtin_inff()) atomicMin(&keys[src], hit_key(t, 0u, GT == 1 ? j : A.tri_perm[j]));
            };
            if constexpr (GT == 1) {
                pass(tri_frags, A.n_tri_rows, (const float4*)nullptr, std::false_type(), std::bool_constant<RT3_FACE_K32 != 0>(), std::integral_constant<uint32_t, 1>(), face_test,
                     [&]() { test_all_raw<RT3_FACE_K32 && RT3_DECODE_FP6>(lane, pairs, n_pairs, face_test); });
            } else {
                // member m of a candidate group: the member's own bounding sphere in f32 — scan_tile's arithmetic with its margin
                // (rt3_valu_scan.hpp; what k_trace and k_mode_r_fast filter with), on the unit direction the filter saw — then the list of survivors
                constexpr uint32_t LPP = kLanesPerPair < GT ? kLanesPerPair : GT, MPL = GT / LPP;   // lanes per pair, members per lane
                auto face_group = [&](uint32_t pair, bool valid, uint32_t part) {
                    const uint32_t src = pair >> kPairLaneShift, g = pair & ((1u << kPairLaneShift) - 1u);
                    const int sl = (int)src;
                    const float sox = __shfl(ray.ox, sl), soy = __shfl(ray.oy, sl), soz = __shfl(ray.oz, sl);
                    const float sdx = __shfl(ux, sl), sdy = __shfl(uy, sl), sdz = __shfl(uz, sl);
                    bound_tests += (unsigned long long)__popcll(__ballot(valid)) * MPL;
                    const uint32_t p0 = valid && g < A.n_tri_leaves ? g * GT + part * MPL : 0u;
                    float4 b[MPL];                                                  // this lane's members' bounds (group order): MPL x 16 contiguous bytes
#pragma unroll
                    for (uint32_t m = 0; m < MPL; m++) b[m] = A.tri_grp[p0 + m];
#pragma unroll
                    for (uint32_t m = 0; m < MPL; m++) {
                        const float cx = b[m].x - sox, cy = b[m].y - soy, cz = b[m].z - soz;
                        const float h = fma_(cz, sdz, fma_(cy, sdy, cx * sdx));
                        const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -b[m].w)));
                        const float disc = fma_(1e-5f, c, fma_(h, h, -c));
                        // the sign bit, as scan_tile reads it (padding: r^2 = -1e30); a face without a bounded hit region (r^2 = 3e38, possibly a
                        // non-finite centre) goes to the exact test whatever the arithmetic above made of it
#if RT3_PRUNE_BEHIND
                        const bool behind = (h < 0.0f) & (c > 2e-3f * b[m].w);
                        const bool keep = valid && g < A.n_tri_leaves && (((__float_as_uint(disc) >> 31) == 0u && !behind) || b[m].w >= 3e38f);
#else
                        const bool keep = valid && g < A.n_tri_leaves && ((__float_as_uint(disc) >> 31) == 0u || b[m].w >= 3e38f);
#endif
                        const unsigned long long km = __ballot(keep);
                        if (km == 0ull) continue;
                        if (keep) fpairs[n_fpairs + prefix_count(km)] = (src << kPairLaneShift) | (p0 + m);
                        n_fpairs += (uint32_t)__popcll(km);
                        __builtin_amdgcn_wave_barrier();
                        if (n_fpairs >= 64u) {
                            n_fpairs -= 64u;
                            face_test(fpairs[n_fpairs + lane], true);
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                };
                if constexpr (SUP == 1) {
                    pass(tri_frags, A.n_tri_rows, (const float4*)nullptr, std::false_type(), std::true_type(), std::integral_constant<uint32_t, GT>(), face_group,
                         [&]() {
                             { const bool lv = lane < n_pairs; set_aside(lv ? pair_decode<RT3_DECODE_FP6 != 0>(pairs[lane]) : 0u, lv, (const float4*)nullptr, std::false_type()); }
                             drain_strip<LPP>(lane, strip, n_strip, face_group);
                             n_strip = 0u; n_pairs = 0u;
                             test_all(lane, fpairs, n_fpairs, face_test);
                         });
                } else {
                    auto stage = super_stage(A.tri_leaf, A.n_tri_rows, std::integral_constant<uint32_t, LPP>(), std::true_type(), face_group);
                    pass(tri_frags, A.n_tri_rows, A.tri_rowb, std::true_type(), std::true_type(), std::integral_constant<uint32_t, SUP>(), stage,
                         [&]() {
                             { const bool lv = lane < n_pairs; set_aside(lv ? pair_decode<RT3_DECODE_FP6 != 0>(pairs[lane]) : 0u, lv, A.tri_rowb, std::true_type()); }
                             drain_strip<(kLanesPerPair < SUP ? kLanesPerPair : SUP)>(lane, strip, n_strip, stage);
                             n_strip = 0u; n_pairs = 0u;
                             flush_leaves(std::integral_constant<uint32_t, LPP>(), face_group);
                             test_all(lane, fpairs, n_fpairs, face_test);
                         });
                }
            }
        }
        if (HAS_SPH) {
            if (HAS_TRI) build_ray_operands32(ray.ox - A.fcx, ray.oy - A.fcy, ray.oz - A.fcz, ux, uy, uz, alive, R32);   // (after the faces' pass: their operands are dead)
            if constexpr (GS == 1) {
                auto test = [&](uint32_t pair, bool valid) {
                    const uint32_t src = pair >> kPairLaneShift, j = pair & ((1u << kPairLaneShift) - 1u);
                    const LaneRay r = fetch_ray<false>(ray, src);
                    exact += (unsigned long long)__popcll(__ballot(valid));
                    if (!valid || j >= A.n_sph) return;
                    float t;
                    if (sphere_root(A.sph[j], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, A.t_min, t) && t < __builtin_inff()) atomicMin(&keys[src], hit_key(t, 1u, j));
                };
                pass(sph_frags, A.n_sph_rows, (const float4*)nullptr, std::false_type(), std::true_type(), std::integral_constant<uint32_t, 1>(), test, [&]() { test_all_raw<RT3_DECODE_FP6 != 0>(lane, pairs, n_pairs, test); });
            } else {
                // member m of a candidate group: the exact test on the member's record (group order), keyed by the sphere's own index
                constexpr uint32_t LPP = kLanesPerPair < GS ? kLanesPerPair : GS, MPL = GS / LPP;   // lanes per pair, members per lane
                // the exact test, on (ray lane, member position) pairs that passed its first half below — 64 at a time: the square root, the root choice
                // and the key are ~40 instructions that 2 % of the member tests need, and a wave runs them whenever ONE of its 64 lanes does
                auto sphere_exact = [&](uint32_t pair, bool valid) {
                    const uint32_t src = pair >> kPairLaneShift, pos = pair & ((1u << kPairLaneShift) - 1u);
                    const LaneRay r = fetch_ray<false>(ray, src);
                    if (!valid) return;
                    float t;
                    if (sphere_root(A.sph_grp[pos], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, A.t_min, t) && t < __builtin_inff())
                        atomicMin(&keys[src], hit_key(t, 1u, A.sph_perm[pos]));
                };
                auto sphere_group = [&](uint32_t pair, bool valid, uint32_t part) {
                    const uint32_t src = pair >> kPairLaneShift, g = pair & ((1u << kPairLaneShift) - 1u);
                    const LaneRay r = fetch_ray<false>(ray, src);
                    exact += (unsigned long long)__popcll(__ballot(valid)) * MPL;
                    const bool ok = valid && g < A.n_sph_leaves;
                    const uint32_t p0 = ok ? g * GS + part * MPL : 0u;
                    float4 sm[MPL];                                                 // this lane's members' records (group order): MPL x 16 contiguous bytes
#pragma unroll
                    for (uint32_t m = 0; m < MPL; m++) sm[m] = A.sph_grp[p0 + m];
#pragma unroll
                    for (uint32_t m = 0; m < MPL; m++) {
                        // sphere_root's own candidate rule, in its own arithmetic (rt3_kernel_common.hpp): whoever passes is tested in full
                        const float cx = sm[m].x - r.ox, cy = sm[m].y - r.oy, cz = sm[m].z - r.oz;
                        const float h = fma_(cz, r.dz, fma_(cy, r.dy, cx * r.dx));
                        const float c = fma_(cz, cz, fma_(cy, cy, fma_(cx, cx, -sm[m].w)));
                        const float disc = fma_(h, h, -c);
                        const bool keep = ok && ((c < 0.0f) | ((disc > 0.0f) & (h > 0.0f)));
                        const unsigned long long km = __ballot(keep);
                        if (km == 0ull) continue;
                        if (keep) fpairs[n_fpairs + prefix_count(km)] = (src << kPairLaneShift) | (p0 + m);
                        n_fpairs += (uint32_t)__popcll(km);
                        __builtin_amdgcn_wave_barrier();
                        if (n_fpairs >= 64u) {
                            n_fpairs -= 64u;
                            sphere_exact(fpairs[n_fpairs + lane], true);
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                };
                if constexpr (SUP == 1) {
                    pass(sph_frags, A.n_sph_rows, (const float4*)nullptr, std::false_type(), std::true_type(), std::integral_constant<uint32_t, GS>(), sphere_group,
                         [&]() {
                             { const bool lv = lane < n_pairs; set_aside(lv ? pair_decode<RT3_DECODE_FP6 != 0>(pairs[lane]) : 0u, lv, (const float4*)nullptr, std::false_type()); }
                             drain_strip<LPP>(lane, strip, n_strip, sphere_group);
                             n_strip = 0u; n_pairs = 0u;
                             test_all(lane, fpairs, n_fpairs, sphere_exact);
                         });
                } else {
                    auto stage = super_stage(A.sph_leaf, A.n_sph_rows, std::integral_constant<uint32_t, LPP>(), std::false_type(), sphere_group);
                    pass(sph_frags, A.n_sph_rows, A.sph_rowb, std::false_type(), std::true_type(), std::integral_constant<uint32_t, SUP>(), stage,
                         [&]() {
                             { const bool lv = lane < n_pairs; set_aside(lv ? pair_decode<RT3_DECODE_FP6 != 0>(pairs[lane]) : 0u, lv, A.sph_rowb, std::false_type()); }
                             drain_strip<(kLanesPerPair < SUP ? kLanesPerPair : SUP)>(lane, strip, n_strip, stage);
                             n_strip = 0u; n_pairs = 0u;
                             flush_leaves(std::integral_constant<uint32_t, LPP>(), sphere_group);
                             test_all(lane, fpairs, n_fpairs, sphere_exact);
                         });
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t kind, ibest;
        float tbest;
        key_decode(keys[lane], kind, ibest, tbest);
        shade_lane<HAS_TRI, HAS_SPH, REF>(A, P, alive, kind, ibest, tbest, A.sph, A.sph_invr, A.sph_mat, A.sph_kind);
    }
    if (lane == 0 && casts != 0) { atomicAdd(A.cast_counter, casts); atomicAdd(A.cast_counter + 1, mfmas); atomicAdd(A.cast_counter + 2, exact); atomicAdd(A.cast_counter + 3, bound_tests); }
#ifdef RT3_PROFILE
    RT3_PHASE(pt_rest)
    if (lane == 0) {
        atomicAdd(A.cast_counter + 12, pt_fill); atomicAdd(A.cast_counter + 13, pt_scan); atomicAdd(A.cast_counter + 14, pt_push);
        atomicAdd(A.cast_counter + 15, pt_test); atomicAdd(A.cast_counter + 4, pt_rest);
    }
#endif
#undef RT3_PHASE
}

// Mode R through the matrix-core filter (camera at the origin, as k_mode_r_fast): one thread per pixel, 1024 pixels per
// workgroup, face bounding spheres streamed through LDS in tiles of 512; the reference's literal test (SequentialRenderer.cpp:53-98)
// runs on the candidates through the pair list; "t >= min_t rejects" of :71 == the lowest index wins ties == the key order.
__global__ __launch_bounds__(kMB) void k_mode_r_mfma(const float4* __restrict__ tri, const u32x4* __restrict__ tri_frags,
                                                    const float4* __restrict__ face_rgb, uint32_t n_faces, CamDev cam,
                                                    uint32_t width, uint32_t height, uint32_t* __restrict__ out, float tcx, float tcy, float tcz) {
    extern __shared__ u32x4 lds_dyn[];
    u32x4* s_frag = lds_dyn;
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_frag + 16 * 256);
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_bm + 16 * kMB);
    uint32_t* s_pairs = reinterpret_cast<uint32_t*>(s_key + kMB);
    const uint32_t tid = threadIdx.x, lane = lane_id();
    uint32_t* pairs = s_pairs + (tid / 64u) * kPairCap;
    unsigned long long* keys = s_key + (tid & ~63u);
    const uint32_t pixel = blockIdx.x * kMB + tid;
    const bool valid_px = pixel < width * height;
    const uint32_t x = valid_px ? pixel % width : 0u, y = valid_px ? pixel / width : 0u;
    const float u = (float)((double)(float)x / ((double)(float)width - 1.0));
    const float v = (float)((double)(float)(height - 1 - y) / ((double)(float)height - 1.0));
    const float ox = cam.ox, oy = cam.oy, oz = cam.oz;              // all zero (checked by the host)
    const float dx = ((cam.lx + u * cam.hx) + v * cam.vx) - ox;
    const float dy = ((cam.ly + u * cam.hy) + v * cam.vy) - oy;
    const float dz = ((cam.lz + u * cam.hz) + v * cam.vz) - oz;
    const float inv = 1.0f / __builtin_sqrtf(dot3(dx, dy, dz, dx, dy, dz));     // unit direction for the filter only
#if RT3_FACE_K32
    RayOperands32 R;
    build_ray_operands32(ox - tcx, oy - tcy, oz - tcz, dx * inv, dy * inv, dz * inv, valid_px, R);     // the fragments are about (tcx, tcy, tcz)
    constexpr uint32_t kVec = 128u, kTile = 32u;
#else
    RayOperands16 R;
    build_ray_operands16(ox - tcx, oy - tcy, oz - tcz, dx * inv, dy * inv, dz * inv, valid_px, R);     // the fragments are about (tcx, tcy, tcz)
    constexpr uint32_t kVec = 256u, kTile = 16u;
#endif
    const LaneRay ray = { ox, oy, oz, dx, dy, dz, true };
    keys[lane] = kKeyNone;
    uint32_t n_pairs = 0;

    auto test = [&](uint32_t pair, bool valid) {
        const uint32_t src = pair >> kPairLaneShift, j = pair & ((1u << kPairLaneShift) - 1u);
        const LaneRay r = fetch_ray<false>(ray, src);
        if (!valid || j >= n_faces) return;
        const float4* f = tri + (size_t)j * 4;
        const float4 n = f[0], p1 = f[1], p2 = f[2], p3 = f[3];
        const float t_hi = __uint_as_float((uint32_t)(keys[src] >> 32));
        float t;
        if (!face_t(n, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, true, t) || !(t >= 0.0f && t <= t_hi)) return;    // :70-71: literal formula, t < 0 rejects
        if (!face_inside(n, p1, p2, p3, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, t)) return;
        if (t < __builtin_inff()) atomicMin(&keys[src], hit_key(t, 0u, j));
    };
    const uint32_t total_blocks = (n_faces + 31u) / 32u;
    for (uint32_t b0 = 0; b0 < total_blocks; b0 += kTile) {
        const uint32_t nb = min(kTile, total_blocks - b0);
        fill_tile(s_frag, tri_frags + (size_t)b0 * kVec, nb * kVec, tid);
        for (uint32_t h0 = 0; h0 < nb; h0 += 16) {
            const uint32_t hb = min(16u, nb - h0);
#if RT3_FACE_K32
            const uint32_t nz = mfma32k_scan_tile(s_frag + (size_t)h0 * kVec, hb, R, s_bm + tid, lane, h0, 3);
#else
            const uint32_t nz = mfma16_scan_tile(s_frag + (size_t)h0 * kVec, hb, R, s_bm + tid, lane);
#endif
            push_pairs16<kMB, RT3_FACE_K32 && RT3_DECODE_FP6>(nz, hb, s_bm + tid, (b0 + h0) * 32u, lane, pairs, n_pairs, test);
        }
    }
    test_all_raw<RT3_FACE_K32 && RT3_DECODE_FP6>(lane, pairs, n_pairs, test);
    __builtin_amdgcn_wave_barrier();
    if (!valid_px) return;
    uint32_t kind, min_i;
    float min_t;
    key_decode(keys[lane], kind, min_i, min_t);
    float r, g, b;
    if (kind != 0u) { const float4 c = face_rgb[min_i]; r = c.x; g = c.y; b = c.z; }
    else sky(dx, dy, dz, r, g, b);
    out[pixel] = pack_pixel(r, g, b);
}

}  // namespace
