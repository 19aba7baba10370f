// Micro-benchmark: the scan's instruction mix (per 32 rows x 64 rays: 4 ds_read_b128 of operand fragments, K = 64 of bf16 MFMA, 32 v_alignbit
// on the results) with the two bf16 MFMA shapes — 8 x v_mfma_f32_32x32x16_bf16 against 16 x v_mfma_f32_16x16x32_bf16, the same FLOP —
// at 4 waves per SIMD on random operands.  MI355X_MICROARCH.md (DVFS give-back, item 7) reports the 16x16x32 shape holding a higher clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma_shape tools/ubench_mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ u32x4 rnd4(unsigned& s) {           // four dwords of two random bf16 in [-2, 2) each
    u32x4 v;
    for (int i = 0; i < 4; i++) { const unsigned a = rnd(s), b = rnd(s); v[i] = (0x3F80u | (a & 0x807Fu)) | ((0x3F80u | (b & 0x807Fu)) << 16); }
    return v;
}

template <int SHAPE, int DECODE = 2, int READS = 1>            // SHAPE 32: 32x32x16, 16: 16x16x32; DECODE 2: 32 v_alignbit, 1: 16 v_max3_f32, 0: none; READS: operand fragments re-read from LDS
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
    __shared__ u32x4 lds[4096];
    const unsigned t = threadIdx.x;
    unsigned seed = t * 2654435761u + 99u;
    for (unsigned i = t; i < 4096; i += 1024) lds[i] = rnd4(seed);
    __syncthreads();
    u32x4 a0 = lds[t & 63], a1 = lds[64 + (t & 63)], a2 = lds[128 + (t & 63)], a3 = lds[192 + (t & 63)];
    u32x4 b[8];
    for (int i = 0; i < 8; i++) b[i] = rnd4(seed);
    unsigned n0 = ~0u, n1 = ~0u, acc = 0;
    auto bf = [](const u32x4& v) { return __builtin_bit_cast(bf16x8, v); };
    for (int it = 0; it < iters; it++) {
        const u32x4* f = lds + ((it & 15) * 256) + (t & 63);
        if (SHAPE == 32) {
            const f32x16 zero = { 0 };
            f32x16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0), bf(b[0]), zero, 0, 0, 0);
            f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a0), bf(b[1]), zero, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(b[0]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(b[1]), d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(b[2]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(b[3]), d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(b[4]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(b[5]), d1, 0, 0, 0);
            a0 = f[0]; a1 = f[64]; a2 = f[128]; a3 = f[192];
            FENCE;
            for (int g = 0; g < 16; g++) n0 = __builtin_amdgcn_alignbit(n0, __float_as_uint(d0[g]), 31);
            for (int g = 0; g < 16; g++) n1 = __builtin_amdgcn_alignbit(n1, __float_as_uint(d1[g]), 31);
        } else {
            // 32 rows = 2 halves of 16 (a0/a1 = K 0..31 / 32..63 of half 0, a2/a3 of half 1); 64 rays = 4 groups of 16 (b[2G], b[2G+1])
            const f32x4 zero = { 0 };
            f32x4 d[8];
#pragma unroll
            for (int G = 0; G < 4; G++) {
                d[G] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a0), bf(b[2 * G]), zero, 0, 0, 0);
                d[4 + G] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a2), bf(b[2 * G]), zero, 0, 0, 0);
            }
#pragma unroll
            for (int G = 0; G < 4; G++) {
                d[G] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a1), bf(b[2 * G + 1]), d[G], 0, 0, 0);
                d[4 + G] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(a3), bf(b[2 * G + 1]), d[4 + G], 0, 0, 0);
            }
            if (READS) { a0 = f[0]; a1 = f[64]; a2 = f[128]; a3 = f[192]; }
            FENCE;
            if (DECODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    for (int g = 0; g < 4; g++) { n0 = __builtin_amdgcn_alignbit(n0, __float_as_uint(d[i][g]), 31); n1 = __builtin_amdgcn_alignbit(n1, __float_as_uint(d[4 + i][g]), 31); }
            } else if (DECODE == 1) {
                float m = __builtin_fmaxf(d[0][0], d[0][1]);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i == 0) m = __builtin_fmaxf(__builtin_fmaxf(m, d[0][2]), d[0][3]);
                    else { m = __builtin_fmaxf(__builtin_fmaxf(m, d[i][0]), d[i][1]); m = __builtin_fmaxf(__builtin_fmaxf(m, d[i][2]), d[i][3]); }
                }
                n0 ^= __float_as_uint(m);
            } else { n0 ^= __float_as_uint(d[0][0] + d[7][3]); }
        }
        FENCE;
        acc += n0 ^ n1;
    }
    out[blockIdx.x * 1024 + t] = acc;
}

template <int SHAPE, int DECODE = 2, int READS = 1>
void run(unsigned* d_out, int cus) {
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, DECODE, READS><<<cus, 1024>>>(d_out, 2000);
    hipEventRecord(e0);
    k<SHAPE, DECODE, READS><<<cus, 1024>>>(d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)cus * 16 * iters * 8 * 32768;
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("%s + %d ds_read_b128 + %s per 32x64 block: %8.2f ms  %7.1f TFLOP/s bf16 (%.1f %% of 2500)\n",
           SHAPE == 32 ? " 8 x v_mfma_f32_32x32x16_bf16" : "16 x v_mfma_f32_16x16x32_bf16", READS ? 4 : 0,
           DECODE == 2 ? "32 v_alignbit" : DECODE == 1 ? "16 v_max3_f32 " : "no decode   ", ms, tf, tf / 25.0);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d_out; hipMalloc(&d_out, (size_t)cus * 1024 * 4);
    for (int rep = 0; rep < 2; rep++) { run<32>(d_out, cus); run<16>(d_out, cus); }
    for (int rep = 0; rep < 2; rep++) { run<16, 1, 1>(d_out, cus); run<16, 0, 1>(d_out, cus); run<16, 2, 0>(d_out, cus); run<16, 0, 0>(d_out, cus); }
    return 0;
}
