// Micro-benchmark: does an SGPR source operand slow a VALU instruction down on gfx950, and in which encodings?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b, float c, float d) {
    float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    const float va = a * 1.0001f, vb = b * 0.9999f;                     // VGPR copies
    for (int i = 0; i < iters; i++) {
#define EACH(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
        if (MODE == 0) {        // VOP2 sub, VGPR operands
#define A0(X) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(X) : "v"(va));
            EACH(A0) EACH(A0)
        } else if (MODE == 1) { // VOP2 sub, SGPR src0 (as the sphere loop: v_sub_f32 v, s, v), 4 different SGPRs
#define A1(X, S) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(X) : "s"(S));
            A1(x0, a) A1(x1, b) A1(x2, c) A1(x3, d) A1(x4, a) A1(x5, b) A1(x6, c) A1(x7, d)
            A1(x0, b) A1(x1, c) A1(x2, d) A1(x3, a) A1(x4, b) A1(x5, c) A1(x6, d) A1(x7, a)
        } else if (MODE == 2) { // VOP2 sub, the SAME SGPR every time
            A1(x0, a) A1(x1, a) A1(x2, a) A1(x3, a) A1(x4, a) A1(x5, a) A1(x6, a) A1(x7, a)
            A1(x0, a) A1(x1, a) A1(x2, a) A1(x3, a) A1(x4, a) A1(x5, a) A1(x6, a) A1(x7, a)
        } else if (MODE == 3) { // VOP3 fma, VGPR operands
#define A3(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(va), "v"(vb));
            EACH(A3) EACH(A3)
        } else if (MODE == 4) { // VOP3 fma, SGPR in src2 with neg (v_fma v, v, v, -s)
#define A4(X, S) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(X) : "v"(va), "s"(S));
            A4(x0, a) A4(x1, b) A4(x2, c) A4(x3, d) A4(x4, a) A4(x5, b) A4(x6, c) A4(x7, d)
            A4(x0, b) A4(x1, c) A4(x2, d) A4(x3, a) A4(x4, b) A4(x5, c) A4(x6, d) A4(x7, a)
        } else if (MODE == 5) { // VOP2 fmac, SGPR src0
#define A5(X, S) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(X) : "s"(S), "v"(va));
            A5(x0, a) A5(x1, b) A5(x2, c) A5(x3, d) A5(x4, a) A5(x5, b) A5(x6, c) A5(x7, d)
            A5(x0, b) A5(x1, c) A5(x2, d) A5(x3, a) A5(x4, b) A5(x5, c) A5(x6, d) A5(x7, a)
        } else if (MODE == 6) { // the sphere-test mix: 3 sub(s) + mul + 5 fmac/fma(v) + 1 fma(-s) + alignbit
            asm volatile(
                "v_sub_f32_e32 %0, %8, %4\n v_sub_f32_e32 %1, %9, %5\n v_mul_f32_e32 %2, %6, %0\n v_fma_f32 %3, %0, %0, -%11\n"
                "v_sub_f32_e32 %0, %10, %4\n v_fmac_f32_e32 %2, %1, %7\n v_fmac_f32_e32 %3, %1, %1\n v_fmac_f32_e32 %2, %0, %6\n"
                "v_fmac_f32_e32 %3, %0, %0\n v_fma_f32 %3, %2, %2, -%3\n v_alignbit_b32 %4, %4, %3, 31\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(va), "v"(vb), "s"(a), "s"(b), "s"(c), "s"(d));
            asm volatile(
                "v_sub_f32_e32 %0, %8, %4\n v_sub_f32_e32 %1, %9, %5\n v_mul_f32_e32 %2, %6, %0\n v_fma_f32 %3, %0, %0, -%11\n"
                "v_sub_f32_e32 %0, %10, %4\n v_fmac_f32_e32 %2, %1, %7\n v_fmac_f32_e32 %3, %1, %1\n v_fmac_f32_e32 %2, %0, %6\n"
                "v_fmac_f32_e32 %3, %0, %0\n v_fma_f32 %3, %2, %2, -%3\n v_alignbit_b32 %4, %4, %3, 31\n"
                : "+v"(x6), "+v"(x7), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(va), "v"(vb), "s"(b), "s"(c), "s"(d), "s"(a));
        } else {                // same mix, all operands in VGPRs
            asm volatile(
                "v_sub_f32_e32 %0, %8, %4\n v_sub_f32_e32 %1, %9, %5\n v_mul_f32_e32 %2, %6, %0\n v_fma_f32 %3, %0, %0, -%9\n"
                "v_sub_f32_e32 %0, %8, %4\n v_fmac_f32_e32 %2, %1, %7\n v_fmac_f32_e32 %3, %1, %1\n v_fmac_f32_e32 %2, %0, %6\n"
                "v_fmac_f32_e32 %3, %0, %0\n v_fma_f32 %3, %2, %2, -%3\n v_alignbit_b32 %4, %4, %3, 31\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(va), "v"(vb), "v"(va), "v"(vb));
            asm volatile(
                "v_sub_f32_e32 %0, %8, %4\n v_sub_f32_e32 %1, %9, %5\n v_mul_f32_e32 %2, %6, %0\n v_fma_f32 %3, %0, %0, -%9\n"
                "v_sub_f32_e32 %0, %8, %4\n v_fmac_f32_e32 %2, %1, %7\n v_fmac_f32_e32 %3, %1, %1\n v_fmac_f32_e32 %2, %0, %6\n"
                "v_fmac_f32_e32 %3, %0, %0\n v_fma_f32 %3, %2, %2, -%3\n v_alignbit_b32 %4, %4, %3, 31\n"
                : "+v"(x6), "+v"(x7), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(va), "v"(vb), "v"(vb), "v"(va));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char* name, double instr_per_iter) {
    float* out;
    const int blocks = 256 * 8, iters = 200000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1000, 1.0f, 2.0f, 3.0f, 4.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0f, 2.0f, 3.0f, 4.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = instr_per_iter * iters * blocks * 4.0;          // 4 waves per block
    printf("%-46s %8.3f ms   %6.3f wave-instr / SIMD / ns\n", name, ms, wave_instr / 1024.0 / (ms * 1e6));
    hipFree(out);
}

int main() {
    run<0>("VOP2 v_sub, VGPR operands", 16);
    run<1>("VOP2 v_sub, SGPR src0 (4 different)", 16);
    run<2>("VOP2 v_sub, SGPR src0 (always the same)", 16);
    run<3>("VOP3 v_fma, VGPR operands", 16);
    run<4>("VOP3 v_fma, -SGPR src2 (4 different)", 16);
    run<5>("VOP2 v_fmac, SGPR src0 (4 different)", 16);
    run<6>("sphere-test mix, sphere data in SGPRs", 22);
    run<7>("sphere-test mix, all VGPRs", 22);
    return 0;
}
