// ubench_fp6_decode.hip — can ONE gfx950 instruction read the signs of 32 accumulators?
// The scan of the matrix-filter kernels is bound by vector-ALU issue: one v_alignbit_b32 (4.3 SIMD cycles with four waves per SIMD) per
// (ray, row) result, 32 per row block (tools/ubench_decode.hip, tools/ubench_valu_rate.hip).  gfx950's block-scaled conversions take whole
// register blocks: v_cvt_scalef32_2xpk16_fp6_f32 turns 2 x 16 f32 registers into 32 six-bit floats (6 registers).  If each keeps the sign of its
// input, the 32 sign bits of a row block cost one conversion and a handful of v_bfi_b32.  This probe answers, on the hardware:
//   (1) what the instruction costs (SIMD cycles per wave-instruction, four waves per SIMD, as ubench_valu_rate.hip measures it), beside
//       other packing candidates (v_cvt_scalef32_pk_fp8_f32, v_cvt_pk_fp8_f32, v_cvt_pkrtz_f16_f32, v_pk_mul_f32, v_pk_fma_f32);
//   (2) which bit of the 192 holds the sign of which input register;
//   (3) whether the sign survives for every class of input (zeros, denormals, underflow, overflow, infinities, NaNs of both signs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_fp6_decode tools/ubench_fp6_decode.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (1b) the scan's row block: 8 MFMAs (16x16x32 bf16) whose 32 results feed the decode — does the conversion overlap the matrix pipe of the other waves?
//   MODE 0: the MFMAs alone   1: + one v_cvt_scalef32_2xpk16_fp6_f32 + 9 merge instructions   2: + 32 v_alignbit_b32   3: the conversion alone
template <int MODE>
__global__ __launch_bounds__(1024) void block_rate(unsigned* out, unsigned long long* clk, int iters) {
    const unsigned t = threadIdx.x;
    bf16x8 a0, a1, b[4];
    for (int i = 0; i < 8; i++) { a0[i] = (__bf16)(float)(t % 7u + i); a1[i] = (__bf16)(float)(t % 5u) - (__bf16)(float)i; for (int g = 0; g < 4; g++) b[g][i] = (__bf16)(float)((t + g) % 3u) - (__bf16)1.0f; }
    unsigned acc = t;
    float f[5] = { 1.0f + (float)t, 0.5f, 0.25f, 2.0f, 1.0e-3f };
    const f32x4 zero = { 0.0f, 0.0f, 0.0f, 0.0f };
    f32x16 lo, hi;
    for (int i = 0; i < 16; i++) { lo[i] = (float)i - 3.0f; hi[i] = 5.0f - (float)i; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        f32x4 d[8];
        if (MODE != 3 && MODE != 4 && MODE != 5) {
#pragma unroll
            for (int g = 0; g < 4; g++) { d[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[g], zero, 0, 0, 0); d[4 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[g], zero, 0, 0, 0); }
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int j = 0; j < 4; j++) { lo[4 * g + j] = d[g][j]; hi[4 * g + j] = d[4 + g][j]; }
        }
        if (MODE == 1 || MODE == 3) {
            u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo, hi, 1.0f);
            const unsigned x = (r[0] & 0x20820820u) | (r[1] & 0x08208208u) | (r[2] & 0x82082082u);
            const unsigned y = (r[3] & 0x20820820u) | (r[4] & 0x08208208u) | (r[5] & 0x82082082u);
            acc ^= x | (y >> 1);
            if (MODE == 3) { lo[0] = __uint_as_float(acc & 0x3FFFFFFFu); }
        } else if (MODE >= 4) {                                       // which unit does the conversion occupy?
            //   4: conversion alone (its 6 results unused)   5: conversion + 16 independent v_fma_f32   6: 8 MFMA + 16 v_fma_f32
            //   7: 8 MFMA + conversion   8: 8 MFMA + conversion + 16 v_fma_f32
            if (MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8) {
                u32x6 r;
                asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, 1.0" : "=v"(r) : "v"(lo), "v"(hi));
                asm volatile("" :: "v"(r));
            } else {
#pragma unroll
                for (int g = 0; g < 8; g++) asm volatile("" :: "v"(d[g]));
            }
            if (MODE == 5 || MODE == 6 || MODE == 8) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i & 3]) : "v"(f[4]));
            }
        } else if (MODE == 2) {
            unsigned n = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; i++) { n = __builtin_amdgcn_alignbit(n, __float_as_uint(lo[i]), 31); n = __builtin_amdgcn_alignbit(n, __float_as_uint(hi[i]), 31); }
            acc ^= n;
        } else {
#pragma unroll
            for (int g = 0; g < 8; g++) asm volatile("" :: "v"(d[g]));
        }
        if (MODE < 4) a0[0] = (__bf16)__uint_as_float((acc & 0x00010000u) | 0x3F800000u);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 1024 + t] = acc ^ __float_as_uint(f[0] + f[1] + f[2] + f[3]);
}
template <int MODE>
void run_block(unsigned* d_out, unsigned long long* d_clk, int cus, const char* what) {
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    block_rate<MODE><<<cus, 1024>>>(d_out, d_clk, 500);
    hipEventRecord(e0);
    block_rate<MODE><<<cus, 1024>>>(d_out, d_clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * cus);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < cus; i++) ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double g = ghz[cus / 2];
    std::printf("%-64s %7.2f ms  clock %.2f GHz  %6.1f SIMD-cycles per row block and wave\n", what, ms, g, ms * 1e-3 * g * 1e9 / ((double)iters * 4));
    std::fflush(stdout);
}

template <int OP>
__global__ __launch_bounds__(1024) void rate(unsigned* out, unsigned long long* clk, int iters) {
    const unsigned t = threadIdx.x;
    f32x16 a, b;
    for (int i = 0; i < 16; i++) { a[i] = (float)(int)(t * 3u + i) - 40.0f; b[i] = 17.0f - (float)(int)(t + 5u * i); }
    unsigned acc = t;
    float s = 1.0f;
    f32x2 p = { a[0], a[1] }, q = { b[0], b[1] }, r2 = { 0.0f, 0.0f };
    unsigned w = t;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {
                u32x6 r;
                asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(s));
                asm volatile("" :: "v"(r));
            } else if (OP == 1) {
                u32x6 r;
                asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(s));
                asm volatile("" :: "v"(r));
            } else if (OP == 2) { asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(w) : "v"(a[u]), "v"(b[u]), "v"(s)); }
            else if (OP == 3) { asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(w) : "v"(a[u]), "v"(b[u])); }
            else if (OP == 4) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w) : "v"(a[u]), "v"(b[u])); asm volatile("" :: "v"(w)); }
            else if (OP == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(p), "v"(q)); asm volatile("" :: "v"(r2)); }
            else if (OP == 6) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r2) : "v"(p), "v"(q)); }
            else if (OP == 7) { asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc) : "v"(a[u])); }
            else if (OP == 8) { asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3" : "+v"(w) : "v"(a[u]), "v"(b[u]), "v"(s)); }
            else if (OP == 9) { asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(acc) : "v"(w), "v"(t)); }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 1024 + t] = acc ^ w ^ __float_as_uint(r2[0]) ^ __float_as_uint(r2[1]);
}

template <int OP>
void run_rate(unsigned* d_out, unsigned long long* d_clk, int cus, const char* what, int values_per_instruction) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<OP><<<cus, 1024>>>(d_out, d_clk, 500);
    hipEventRecord(e0);
    rate<OP><<<cus, 1024>>>(d_out, d_clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * cus);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < cus; i++) ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double g = ghz[cus / 2];
    const double cyc = ms * 1e-3 * g * 1e9 / ((double)iters * 8 * 4);                 // SIMD cycles per wave-instruction (4 waves per SIMD)
    std::printf("%-40s %7.2f ms  clock %.2f GHz  %6.2f SIMD-cycles per wave-instruction = %5.2f per value\n", what, ms, g, cyc, cyc / values_per_instruction);
    std::fflush(stdout);
}

// semantics: every lane converts its own 32 inputs, the host checks the words
template <int BF6>
__global__ void convert(const float* in, unsigned* out, float scale) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x16 a, b;
    for (int i = 0; i < 16; i++) { a[i] = in[(size_t)t * 32 + i]; b[i] = in[(size_t)t * 32 + 16 + i]; }
    u32x6 r;
    if (BF6) r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, scale);
    else r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; i++) out[(size_t)t * 6 + i] = r[i];
}

static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d_out; hipMalloc(&d_out, (size_t)cus * 1024 * 4);
    unsigned long long* d_clk; hipMalloc(&d_clk, (size_t)cus * 16);
    std::printf("(1) issue cost, 1024-thread workgroup per CU (4 waves per SIMD)\n");
    run_rate<7>(d_out, d_clk, cus, "v_alignbit_b32 (today's decode)", 1);
    run_rate<9>(d_out, d_clk, cus, "v_bfi_b32", 1);
    run_rate<0>(d_out, d_clk, cus, "v_cvt_scalef32_2xpk16_fp6_f32", 32);
    run_rate<1>(d_out, d_clk, cus, "v_cvt_scalef32_2xpk16_bf6_f32", 32);
    run_rate<2>(d_out, d_clk, cus, "v_cvt_scalef32_pk_fp8_f32", 2);
    run_rate<3>(d_out, d_clk, cus, "v_cvt_pk_fp8_f32", 2);
    run_rate<8>(d_out, d_clk, cus, "v_cvt_scalef32_pk_fp4_f32", 2);
    run_rate<4>(d_out, d_clk, cus, "v_cvt_pkrtz_f16_f32", 2);
    run_rate<5>(d_out, d_clk, cus, "v_pk_mul_f32", 2);
    run_rate<6>(d_out, d_clk, cus, "v_pk_fma_f32", 2);

    std::printf("(1b) a row block of the scan, four waves per SIMD (matrix pipe alone: 8 x 16 = 128 cycles)\n");
    run_block<0>(d_out, d_clk, cus, "8 MFMA 16x16x32 bf16");
    run_block<2>(d_out, d_clk, cus, "8 MFMA + 32 v_alignbit_b32");
    run_block<1>(d_out, d_clk, cus, "8 MFMA + v_cvt_scalef32_2xpk16_fp6_f32 + merge");
    run_block<3>(d_out, d_clk, cus, "v_cvt_scalef32_2xpk16_fp6_f32 + merge alone");
    run_block<4>(d_out, d_clk, cus, "conversion alone");
    run_block<5>(d_out, d_clk, cus, "conversion + 16 independent v_fma_f32");
    run_block<6>(d_out, d_clk, cus, "8 MFMA + 16 v_fma_f32");
    run_block<7>(d_out, d_clk, cus, "8 MFMA + conversion");
    run_block<8>(d_out, d_clk, cus, "8 MFMA + conversion + 16 v_fma_f32");

    // (2) layout: input j negative, the other 31 positive: which of the 192 bits differ from the all-positive result?
    const int kLanes = 64 * 64;
    std::vector<float> in((size_t)kLanes * 32);
    std::vector<unsigned> res((size_t)kLanes * 6);
    float* d_in; unsigned* d_res;
    hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_res, res.size() * 4);
    for (int bf6 = 0; bf6 < 2; bf6++) {
        for (int lane = 0; lane < kLanes; lane++)
            for (int j = 0; j < 32; j++) in[(size_t)lane * 32 + j] = (lane >= 1 && lane <= 32 && j == lane - 1) ? -1.0f : 1.0f;
        hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
        if (bf6) convert<1><<<kLanes / 64, 64>>>(d_in, d_res, 1.0f); else convert<0><<<kLanes / 64, 64>>>(d_in, d_res, 1.0f);
        hipMemcpy(res.data(), d_res, res.size() * 4, hipMemcpyDeviceToHost);
        std::printf("(2) %s: all +1.0 -> %08x %08x %08x %08x %08x %08x; sign bit of input j (a[0..15], b[0..15]) at bit:", bf6 ? "bf6" : "fp6",
                    res[0], res[1], res[2], res[3], res[4], res[5]);
        bool regular = true;
        for (int j = 0; j < 32; j++) {
            int where = -1, count = 0;
            for (int bit = 0; bit < 192; bit++)
                if (((res[(size_t)(j + 1) * 6 + bit / 32] ^ res[bit / 32]) >> (bit % 32)) & 1u) { where = bit; count++; }
            std::printf(" %d%s", where, count == 1 ? "" : "(!)");
            regular = regular && count == 1 && where == (j < 16 ? 12 * j + 5 : 12 * (j - 16) + 11);
        }
        std::printf("\n    => %s\n", regular ? "a[i] -> field 2 i (sign at bit 12 i + 5), b[i] -> field 2 i + 1 (bit 12 i + 11)" : "NOT the interleaved order: read the list");
        // (3) sign survival per class: all 32 inputs of a lane = the value
        const uint32_t cls[] = { 0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007FFFFFu, 0x807FFFFFu, 0x00800000u, 0x80800000u,
                                 0x0DA24260u, 0x8DA24260u, 0x2EDBE6FFu, 0xAEDBE6FFu, 0x3C23D70Au, 0xBC23D70Au, 0x3DCCCCCDu, 0xBDCCCCCDu, 0x3E000000u, 0xBE000000u,
                                 0x3F800000u, 0xBF800000u, 0x40F00000u, 0xC0F00000u, 0x42C80000u, 0xC2C80000u, 0x7149F2CAu, 0xF149F2CAu, 0x7F7FFFFFu, 0xFF7FFFFFu,
                                 0x7F800000u, 0xFF800000u, 0x7FC00000u, 0xFFC00000u, 0x7F800001u, 0xFF800001u, 0x7FFFFFFFu, 0xFFFFFFFFu };
        const int n_cls = (int)(sizeof(cls) / sizeof(cls[0]));
        const float scales[] = { 1.0f, 1.0e-30f, 1.0e30f };
        for (float scale : scales) {
            for (int lane = 0; lane < kLanes; lane++)
                for (int j = 0; j < 32; j++) in[(size_t)lane * 32 + j] = from_bits(cls[lane % n_cls]);
            hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
            if (bf6) convert<1><<<kLanes / 64, 64>>>(d_in, d_res, scale); else convert<0><<<kLanes / 64, 64>>>(d_in, d_res, scale);
            hipMemcpy(res.data(), d_res, res.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            std::printf("(3) %s scale %g: input bits -> field (of field 0) and whether all 32 signs equal the input's:\n   ", bf6 ? "bf6" : "fp6", scale);
            for (int c = 0; c < n_cls; c++) {
                unsigned signs = 0;
                for (int j = 0; j < 32; j++) signs |= ((res[(size_t)c * 6 + (6 * j + 5) / 32] >> ((6 * j + 5) % 32)) & 1u) << j;
                const unsigned want = (cls[c] >> 31) ? 0xFFFFFFFFu : 0u;
                std::printf(" %08x->%02x%s", cls[c], res[(size_t)c * 6] & 63u, signs == want ? "" : "(SIGN LOST)");
                bad += signs != want;
            }
            std::printf("\n    => %d classes lose the sign\n", bad);
        }
        // random values over the whole exponent range, both signs (4096 lanes x 32)
        uint32_t state = 12345u + bf6;
        for (size_t i = 0; i < in.size(); i++) { state = state * 1664525u + 1013904223u; uint32_t u = state; if ((u & 0x7F800000u) == 0x7F800000u) u &= 0xFF7FFFFFu; in[i] = from_bits(u); }
        hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
        if (bf6) convert<1><<<kLanes / 64, 64>>>(d_in, d_res, 1.0f); else convert<0><<<kLanes / 64, 64>>>(d_in, d_res, 1.0f);
        hipMemcpy(res.data(), d_res, res.size() * 4, hipMemcpyDeviceToHost);
        long lost = 0;
        for (int lane = 0; lane < kLanes; lane++)
            for (int j = 0; j < 32; j++) {
                uint32_t u; std::memcpy(&u, &in[(size_t)lane * 32 + j], 4);
                const int bit = j < 16 ? 12 * j + 5 : 12 * (j - 16) + 11;
                lost += ((res[(size_t)lane * 6 + bit / 32] >> (bit % 32)) & 1u) != (u >> 31);
            }
        std::printf("(3) %s: %d random finite values: %ld signs lost\n", bf6 ? "bf6" : "fp6", kLanes * 32, lost);
    }
    return 0;
}
