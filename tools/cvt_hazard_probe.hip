// cvt_hazard_probe.hip — counter-example kernel for the v_cvt_pk_bf16_f32 forwarding hazard (rt3_matrix_filter.hpp, pk_bf16()).
//   hipcc --offload-arch=gfx950 -O3 -o tools/cvt_hazard_probe tools/cvt_hazard_probe.hip && tools/cvt_hazard_probe
// Every lane converts a stream of float pairs with v_cvt_pk_bf16_f32 and reads the result in the NEXT issue slot (GAP = 0), one
// s_nop later (GAP = 1) or two (GAP = 2), and compares it with round-to-nearest-even done in integer arithmetic.  The destination
// register is reused every iteration with a different value, so a stale read is visible as a mismatch.  BUSY adds an MFMA stream in
// the same wave (as the trace kernels have around their conversions).  Prints mismatches per variant.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t bf16_rn(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

template <int GAP, bool BUSY>
__global__ __launch_bounds__(1024) void probe(uint32_t iters, unsigned long long* mismatches, float* sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t state = tid * 2654435761u + 12345u;
    unsigned long long bad = 0;
    f32x16 acc = { 0 };
    bf16x8 a = { 0 }, b = { 0 };
    for (uint32_t i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const float x = __uint_as_float(0x3F800000u | (state >> 9)), y = __uint_as_float(0x40000000u | ((state * 7u) >> 9));
        uint32_t packed, lo, hi;
        if (GAP == 0)
            asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n\tv_lshlrev_b32 %1, 16, %0\n\tv_and_b32 %2, 0xffff0000, %0"
                         : "=&v"(packed), "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y));
        else if (GAP == 1)
            asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n\ts_nop 0\n\tv_lshlrev_b32 %1, 16, %0\n\tv_and_b32 %2, 0xffff0000, %0"
                         : "=&v"(packed), "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y));
        else
            asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n\ts_nop 1\n\tv_lshlrev_b32 %1, 16, %0\n\tv_and_b32 %2, 0xffff0000, %0"
                         : "=&v"(packed), "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y));
        const uint32_t want_lo = bf16_rn(x) << 16, want_hi = bf16_rn(y) << 16;
        bad += (lo != want_lo) + (hi != want_hi);
        if (BUSY) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            a[0] = (__bf16)x;
        }
    }
    if (BUSY) sink[tid] = acc[0] + acc[7];
    if (bad) atomicAdd(mismatches, bad);
}

// The instruction sequence of the build that fails — hipcc's own code for the three-way bf16 split of two floats, copied from its ISA
// (profiles/r02_cvt_hazard.md): cvt, shift + mask in the next two slots, packed subtract with neg modifiers, cvt of the residuals in the
// next slot, ... — on fixed registers, against the same split done with integer rounding.  NOP = 1: one wait state behind every conversion.
#define RT3_CHAIN(NOPSTR)                                                                                              \
    asm volatile("v_mov_b32 v100, %3\n\tv_mov_b32 v101, %4\n\t"                                                       \
                 "v_cvt_pk_bf16_f32 v102, v100, v101\n\t" NOPSTR                                                       \
                 "v_lshlrev_b32 v104, 16, v102\n\tv_and_b32 v105, 0xffff0000, v102\n\t"                                \
                 "v_pk_add_f32 v[100:101], v[100:101], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n\t"                       \
                 "v_cvt_pk_bf16_f32 v103, v100, v101\n\t" NOPSTR                                                       \
                 "v_lshlrev_b32 v104, 16, v103\n\tv_and_b32 v105, 0xffff0000, v103\n\t"                                \
                 "v_pk_add_f32 v[100:101], v[100:101], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n\t"                       \
                 "v_cvt_pk_bf16_f32 v106, v100, v101\n\t" NOPSTR                                                       \
                 "v_mov_b32 %0, v102\n\tv_mov_b32 %1, v103\n\tv_mov_b32 %2, v106"                                      \
                 : "=v"(ph), "=v"(pm), "=v"(pl) : "v"(x0), "v"(x1) : "v100", "v101", "v102", "v103", "v104", "v105", "v106")
__device__ __forceinline__ void split3_ref(float x, uint32_t* p) {
    p[0] = bf16_rn(x);
    const float r1 = x - __uint_as_float(p[0] << 16);
    p[1] = bf16_rn(r1);
    p[2] = bf16_rn(r1 - __uint_as_float(p[1] << 16));
}
template <int NOP, bool BUSY>
__global__ __launch_bounds__(1024) void probe_chain(uint32_t iters, unsigned long long* mismatches, float* sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t state = tid * 2654435761u + 777u;
    unsigned long long bad = 0;
    f32x16 acc = { 0 };
    bf16x8 a = { 0 }, b = { 0 };
    for (uint32_t i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const float x0 = __uint_as_float(0x3F000000u | (state >> 9)), x1 = -__uint_as_float(0x3E800000u | ((state * 5u) >> 9));
        uint32_t ph, pm, pl;
        if (NOP == 0) RT3_CHAIN(""); else RT3_CHAIN("s_nop 0\n\t");
        uint32_t r0[3], r1[3];
        split3_ref(x0, r0); split3_ref(x1, r1);
        bad += (ph != (r0[0] | (r1[0] << 16))) + (pm != (r0[1] | (r1[1] << 16))) + (pl != (r0[2] | (r1[2] << 16)));
        if (BUSY) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            a[0] = (__bf16)x0;
        }
    }
    if (BUSY) sink[tid] = acc[0] + acc[7];
    if (bad) atomicAdd(mismatches, bad);
}
template <int NOP, bool BUSY>
unsigned long long run_chain(uint32_t iters, unsigned long long* d_bad, float* d_sink) {
    hipMemset(d_bad, 0, 8);
    hipLaunchKernelGGL((probe_chain<NOP, BUSY>), dim3(256), dim3(1024), 0, 0, iters, d_bad, d_sink);
    unsigned long long h = 0;
    hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    return h;
}

template <int GAP, bool BUSY>
unsigned long long run(uint32_t iters, unsigned long long* d_bad, float* d_sink) {
    hipMemset(d_bad, 0, 8);
    hipLaunchKernelGGL((probe<GAP, BUSY>), dim3(256), dim3(1024), 0, 0, iters, d_bad, d_sink);
    unsigned long long h = 0;
    hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    return h;
}

int main() {
    unsigned long long* d_bad; float* d_sink;
    hipMalloc((void**)&d_bad, 8);
    hipMalloc((void**)&d_sink, 256 * 1024 * 4);
    const uint32_t iters = 20000;                       // 256 x 1024 x 20000 = 5.2e9 conversions per variant
    printf("conversions per variant: %.3g\n", 256.0 * 1024.0 * iters);
    printf("gap 0 (consumer in the next slot), VALU only : %llu mismatches\n", run<0, false>(iters, d_bad, d_sink));
    printf("gap 1 (one wait state),            VALU only : %llu mismatches\n", run<1, false>(iters, d_bad, d_sink));
    printf("gap 2 (two wait states),           VALU only : %llu mismatches\n", run<2, false>(iters, d_bad, d_sink));
    printf("gap 0 (consumer in the next slot), MFMA busy : %llu mismatches\n", run<0, true>(iters, d_bad, d_sink));
    printf("gap 1 (one wait state),            MFMA busy : %llu mismatches\n", run<1, true>(iters, d_bad, d_sink));
    printf("gap 2 (two wait states),           MFMA busy : %llu mismatches\n", run<2, true>(iters, d_bad, d_sink));
    printf("hipcc's split chain, no wait states,       VALU only : %llu mismatches\n", run_chain<0, false>(iters, d_bad, d_sink));
    printf("hipcc's split chain, s_nop 0 after each,   VALU only : %llu mismatches\n", run_chain<1, false>(iters, d_bad, d_sink));
    printf("hipcc's split chain, no wait states,       MFMA busy : %llu mismatches\n", run_chain<0, true>(iters, d_bad, d_sink));
    printf("hipcc's split chain, s_nop 0 after each,   MFMA busy : %llu mismatches\n", run_chain<1, true>(iters, d_bad, d_sink));
    return 0;
}
