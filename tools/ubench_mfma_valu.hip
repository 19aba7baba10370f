// Micro-benchmark: how well do v_mfma_f32_32x32x16_bf16 and dependent VALU work overlap at 4 waves per SIMD?
// Each wave loops over "blocks" of 8 MFMAs (two chains of 4) followed by NV v_alignbit on the accumulators (NV = 0, 16, 32, 48, 64),
// in two orders: all MFMAs then all VALU ("phased"), or 1 MFMA / NV/8 VALU interleaved ("pipelined", decode of the previous chain).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma_valu tools/ubench_mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE __builtin_amdgcn_sched_barrier(0)

template <int NV, bool PIPE, int LDS_READS>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
    __shared__ u32x4 lds[4096];
    const unsigned t = threadIdx.x;
    for (unsigned i = t; i < 4096; i += 1024) lds[i] = u32x4{ i, i * 3, i * 5, i * 7 };
    __syncthreads();
    u32x4 a0 = lds[t & 63], a1 = lds[64 + (t & 63)], a2 = lds[128 + (t & 63)], a3 = lds[192 + (t & 63)];
    const u32x4 b0 = { t, t + 1, t + 2, t + 3 }, b1 = { t * 2, t, 7, 9 };
    const f32x16 zero = { 0 };
    unsigned n0 = ~0u, n1 = ~0u, acc = 0;
    auto mfma = [](const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    f32x16 d0 = mfma(a0, b0, zero), d1 = zero;
    d0 = mfma(a1, b0, d0); d0 = mfma(a2, b0, d0); d0 = mfma(a3, b0, d0);
    for (int it = 0; it < iters; it++) {
        const u32x4* f = lds + ((it & 15) * 256) + (t & 63);
        if (PIPE) {
#define BITS(n, d, g, cnt) for (int q = 0; q < cnt; q++) n = __builtin_amdgcn_alignbit(n, __float_as_uint(d[(g + q) & 15]), 31)
            d1 = mfma(a0, b1, zero); if (LDS_READS) a0 = f[0];   BITS(n0, d0, 0, NV / 8);        FENCE;
            d1 = mfma(a1, b1, d1);   if (LDS_READS) a1 = f[64];  BITS(n0, d0, NV / 8, NV / 8);   FENCE;
            d1 = mfma(a2, b1, d1);   if (LDS_READS) a2 = f[128]; BITS(n0, d0, 2 * NV / 8, NV / 8); FENCE;
            d1 = mfma(a3, b1, d1);   if (LDS_READS) a3 = f[192]; BITS(n0, d0, 3 * NV / 8, NV / 8); FENCE;
            d0 = mfma(a0, b0, zero); BITS(n1, d1, 0, NV / 8);        FENCE;
            d0 = mfma(a1, b0, d0);   BITS(n1, d1, NV / 8, NV / 8);   FENCE;
            d0 = mfma(a2, b0, d0);   BITS(n1, d1, 2 * NV / 8, NV / 8); FENCE;
            d0 = mfma(a3, b0, d0);   BITS(n1, d1, 3 * NV / 8, NV / 8); FENCE;
        } else {
            d1 = mfma(a0, b1, zero); d1 = mfma(a1, b1, d1); d1 = mfma(a2, b1, d1); d1 = mfma(a3, b1, d1);
            if (LDS_READS) { a0 = f[0]; a1 = f[64]; a2 = f[128]; a3 = f[192]; }
            FENCE;
            BITS(n0, d0, 0, NV / 2); BITS(n1, d1, 0, NV / 2);
            FENCE;
            d0 = mfma(a0, b0, zero); d0 = mfma(a1, b0, d0); d0 = mfma(a2, b0, d0); d0 = mfma(a3, b0, d0);
            FENCE;
        }
        acc += n0 ^ n1;
    }
    out[blockIdx.x * 1024 + t] = acc + __float_as_uint(d0[3]) + __float_as_uint(d1[5]);
}

template <int NV, bool PIPE, int LDS_READS>
void run(unsigned* d_out, int cus) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, PIPE, LDS_READS><<<cus, 1024>>>(d_out, 100);
    hipEventRecord(e0);
    k<NV, PIPE, LDS_READS><<<cus, 1024>>>(d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)cus * 16 * iters * 8;
    const double tf = mfmas * 32768 / (ms * 1e-3) / 1e12;
    printf("NV %2d per 8 MFMA  %-9s lds %d: %7.2f ms  %7.1f TFLOP/s bf16 (%.0f %% of 2500)  cycles per MFMA and SIMD at 2.4 GHz: %.1f\n", NV,
           PIPE ? "pipelined" : "phased", LDS_READS, ms, tf, tf / 25.0, ms * 1e-3 * 2.4e9 / (iters * 8.0 * 4));
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d_out; hipMalloc(&d_out, (size_t)cus * 1024 * 4);
    run<0, false, 0>(d_out, cus);
    run<16, false, 0>(d_out, cus); run<16, true, 0>(d_out, cus);
    run<32, false, 0>(d_out, cus); run<32, true, 0>(d_out, cus);
    run<32, false, 1>(d_out, cus); run<32, true, 1>(d_out, cus);
    run<48, false, 0>(d_out, cus); run<48, true, 0>(d_out, cus);
    run<64, false, 1>(d_out, cus); run<64, true, 1>(d_out, cus);
    return 0;
}
