// Micro-benchmark: what bounds the K = 32 scan (rt3_matrix_filter.hpp, mfma32k_scan_tile) — per 32 rows x 64 rays and wave: 2 ds_read_b128 of
// operand fragments, 8 x v_mfma_f32_16x16x32_bf16 (128 matrix-pipe cycles), 32 sign decodes — and what other decodes of the 32 results would cost.
// THREADS / 256 waves per SIMD, one workgroup per CU, random bf16 operands.  Reports, per variant, the SIMD cycles per (wave, row block) at the
// in-kernel clock (s_memtime over s_memrealtime): the matrix pipe needs 128 / waves-sharing... i.e. 128 per wave-block whatever the wave count.
//   DEC 0  32 v_alignbit in ONE dependent chain (the product's decode)
//   DEC 1  two chains of 16 + 1 combine          DEC 2  four chains of 8 + 3 combines
//   DEC 3  32 v_cmp_lt_f32 -> SGPR pairs, combined on the scalar unit
//   DEC 4  16 v_max3_f32 (a fold: "any candidate in the block", the floor of a sparse decode)
//   DEC 5  no decode (MFMAs + operand reads alone)
//   DEC 6  32 v_alignbit, no MFMA (the decode alone)
//   DEC 7  16 v_max_f32 folding PAIRS of rows + 16 v_alignbit (half the filter's resolution: a candidate bit names two rows)
//   DEC 8  24 v_max_f32 folding QUADS of rows + 8 v_alignbit
//   SHAPE 32: the same K = 32 products as 4 x v_mfma_f32_32x32x16_bf16 (two column sets x two K steps): half the MFMA issue slots
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_decode tools/ubench_decode.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // (fmaxf() canonicalises both inputs first: three instructions)
__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ u32x4 rnd4(unsigned& s) {           // four dwords of two random bf16 in [-2, 2) each
    u32x4 v;
    for (int i = 0; i < 4; i++) { const unsigned a = rnd(s), b = rnd(s); v[i] = (0x3F80u | (a & 0x807Fu)) | ((0x3F80u | (b & 0x807Fu)) << 16); }
    return v;
}

template <int DEC, int THREADS, int SHAPE = 16>
__global__ __launch_bounds__(THREADS) void k(unsigned* out, unsigned long long* clk, int iters) {
    __shared__ u32x4 lds[4096];                                 // 64 KiB: 32 row blocks x 2 fragments x 64 lanes
    const unsigned t = threadIdx.x;
    unsigned seed = t * 2654435761u + 99u + blockIdx.x;
    for (unsigned i = t; i < 4096; i += THREADS) lds[i] = rnd4(seed);
    __syncthreads();
    u32x4 a0 = lds[t & 63], a1 = lds[64 + (t & 63)];
    u32x4 b[4];
    for (int i = 0; i < 4; i++) b[i] = rnd4(seed);
    unsigned acc = 0;
    unsigned long long sacc = 0;
    auto bf = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        const u32x4* f = lds + ((it & 31) * 128) + (t & 63);
        const f32x4 zero = { 0 };
        f32x4 d[8];
        if (SHAPE == 32) {
            const f32x16 z16 = { 0 };
            f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0), bf(b[0]), z16, 0, 0, 0);
            f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0), bf(b[1]), z16, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(b[2]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(b[3]), d1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) { d[i][j] = d0[4 * i + j]; d[4 + i][j] = d1[4 * i + j]; }
        } else if (DEC != 6) {
#pragma unroll
            for (int G = 3; G >= 0; G--) {
                d[2 * G + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a1), bf(b[G]), zero, 0, 0, 0);
                d[2 * G] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a0), bf(b[G]), zero, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = f32x4{ __uint_as_float(a0[i & 3] + it), __uint_as_float(a1[i & 3]), __uint_as_float(a0[(i + 1) & 3]), __uint_as_float(a1[(i + 2) & 3] ^ acc) };
        }
        a0 = f[0]; a1 = f[64];
        FENCE;
        if (DEC == 0 || DEC == 6) {
            unsigned n = ~0u;
#pragma unroll
            for (int i = 7; i >= 0; i--)
#pragma unroll
                for (int j = 3; j >= 0; j--) n = __builtin_amdgcn_alignbit(n, __float_as_uint(d[i][j]), 31);
            acc += n;
        } else if (DEC == 1) {
            unsigned na = ~0u, nb = 0u;
#pragma unroll
            for (int i = 3; i >= 0; i--)
#pragma unroll
                for (int j = 3; j >= 0; j--) { na = __builtin_amdgcn_alignbit(na, __float_as_uint(d[4 + i][j]), 31); nb = __builtin_amdgcn_alignbit(nb, __float_as_uint(d[i][j]), 31); }
            acc += (na << 16) | nb;
        } else if (DEC == 2) {
            unsigned n[4] = { ~0u, 0u, 0u, 0u };
#pragma unroll
            for (int i = 1; i >= 0; i--)
#pragma unroll
                for (int j = 3; j >= 0; j--)
#pragma unroll
                    for (int c = 0; c < 4; c++) n[c] = __builtin_amdgcn_alignbit(n[c], __float_as_uint(d[2 * c + i][j]), 31);
            acc += (n[0] << 24) | (n[1] << 16) | (n[2] << 8) | n[3];
        } else if (DEC == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) sacc ^= __ballot(d[i][j] < 0.0f);
        } else if (DEC == 4) {
            float m = d[0][0];
#pragma unroll
            for (int i = 0; i < 8; i++) { m = __builtin_fmaxf(__builtin_fmaxf(m, d[i][0]), d[i][1]); m = __builtin_fmaxf(__builtin_fmaxf(m, d[i][2]), d[i][3]); }
            acc += __float_as_uint(m);
        } else if (DEC == 7) {
            unsigned n = ~0u;
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                n = __builtin_amdgcn_alignbit(n, __float_as_uint(vmax(d[i][3], d[i][2])), 31);
                n = __builtin_amdgcn_alignbit(n, __float_as_uint(vmax(d[i][1], d[i][0])), 31);
            }
            acc += n;
        } else if (DEC == 8) {
            unsigned n = ~0u;
#pragma unroll
            for (int i = 7; i >= 0; i--) n = __builtin_amdgcn_alignbit(n, __float_as_uint(vmax(vmax(d[i][3], d[i][2]), vmax(d[i][1], d[i][0]))), 31);
            acc += n;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" :: "v"(d[i]));
        }
        FENCE;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * THREADS + t] = acc + (unsigned)sacc + (unsigned)(sacc >> 32);
}

template <int DEC, int THREADS, int SHAPE = 16>
void run(unsigned* d_out, unsigned long long* d_clk, int cus, const char* what) {
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<DEC, THREADS, SHAPE><<<cus, THREADS>>>(d_out, d_clk, 2000);
    hipEventRecord(e0);
    k<DEC, THREADS, SHAPE><<<cus, THREADS>>>(d_out, d_clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * cus);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < cus; i++) { ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1); cyc.push_back((double)c[2 * i]); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double waves_per_simd = THREADS / 256.0;
    // SIMD cycles per (wave, row block): HIP-event time x in-kernel clock (the s_memtime DIFFERENCE itself reads 2-3x too low on this part —
    // first run of this file — while its ratio to s_memrealtime gives a plausible clock: only the ratio is used)
    const double per_block = ms * 1e-3 * ghz[cus / 2] * 1e9 / ((double)iters * waves_per_simd);
    const double tf = DEC == 6 ? 0.0 : (double)cus * (THREADS / 64) * iters * 8 * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-46s %d waves/SIMD: %7.2f ms  %6.1f SIMD-cycles per wave-block  clock %.2f GHz  %7.1f TFLOP/s bf16 (%.1f %% of 2500)\n",
           what, THREADS / 256, ms, per_block, ghz[cus / 2], tf, tf / 25.0);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d_out; hipMalloc(&d_out, (size_t)cus * 1024 * 4);
    unsigned long long* d_clk; hipMalloc(&d_clk, (size_t)cus * 16);
    for (int rep = 0; rep < 2; rep++) {
        run<0, 1024>(d_out, d_clk, cus, "8 MFMA + 32 alignbit, one chain (product)");
        run<1, 1024>(d_out, d_clk, cus, "8 MFMA + 32 alignbit, two chains");
        run<2, 1024>(d_out, d_clk, cus, "8 MFMA + 32 alignbit, four chains");
        run<3, 1024>(d_out, d_clk, cus, "8 MFMA + 32 v_cmp -> SGPR");
        run<4, 1024>(d_out, d_clk, cus, "8 MFMA + 16 v_max3");
        run<5, 1024>(d_out, d_clk, cus, "8 MFMA, no decode");
        run<6, 1024>(d_out, d_clk, cus, "no MFMA, 32 alignbit one chain");
        run<7, 1024>(d_out, d_clk, cus, "8 MFMA + 16 v_max_f32 + 16 alignbit");
        run<8, 1024>(d_out, d_clk, cus, "8 MFMA + 24 v_max_f32 + 8 alignbit");
        run<0, 1024, 32>(d_out, d_clk, cus, "4 MFMA 32x32x16 + 32 alignbit");
        run<7, 1024, 32>(d_out, d_clk, cus, "4 MFMA 32x32x16 + 16 v_max_f32 + 16 alignbit");
        run<8, 1024, 32>(d_out, d_clk, cus, "4 MFMA 32x32x16 + 24 v_max_f32 + 8 alignbit");
        run<5, 1024, 32>(d_out, d_clk, cus, "4 MFMA 32x32x16, no decode");
    }
    return 0;
}
