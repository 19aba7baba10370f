// Micro-benchmark: issue cost of single vector-ALU instructions on gfx950 with 4 waves per SIMD (1024-thread workgroup, one per CU),
// 32 instructions per loop trip in four independent dependency chains.  Prints SIMD cycles per wave-instruction, by HIP-event time x the
// in-kernel clock (s_memtime / s_memrealtime).  Question behind it (tools/ubench_decode.hip): the scan's sign decode is one v_alignbit_b32 per
// result and the loop is bound by vector-ALU issue — which instructions are cheaper?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu_rate tools/ubench_valu_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define OP4(asmtext)                                                                                      \
    asm volatile(asmtext : "+v"(a0) : "v"(x0), "v"(x1));                                                  \
    asm volatile(asmtext : "+v"(a1) : "v"(x1), "v"(x2));                                                  \
    asm volatile(asmtext : "+v"(a2) : "v"(x2), "v"(x3));                                                  \
    asm volatile(asmtext : "+v"(a3) : "v"(x3), "v"(x0));
#define OP32(asmtext) OP4(asmtext) OP4(asmtext) OP4(asmtext) OP4(asmtext) OP4(asmtext) OP4(asmtext) OP4(asmtext) OP4(asmtext)

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned long long* clk, int iters) {
    const unsigned t = threadIdx.x;
    unsigned a0 = t, a1 = t * 3u + 1u, a2 = t * 5u + 2u, a3 = t * 7u + 3u;
    unsigned x0 = 0x3F800000u + t, x1 = 0x3F900000u + 3u * t, x2 = 0xBF800000u + 5u * t, x3 = 0x3FA00000u + 7u * t;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { OP32("v_alignbit_b32 %0, %0, %1, 31") }
        else if (OP == 1) { OP32("v_max_f32 %0, %0, %1") }
        else if (OP == 2) { OP32("v_add_f32 %0, %0, %1") }
        else if (OP == 3) { OP32("v_fma_f32 %0, %0, %1, %2") }
        else if (OP == 4) { OP32("v_max3_f32 %0, %0, %1, %2") }
        else if (OP == 5) { OP32("v_and_b32 %0, %0, %1") }
        else if (OP == 6) { OP32("v_or_b32 %0, %0, %1") }
        else if (OP == 7) { OP32("v_lshl_or_b32 %0, %0, 1, %1") }
        else if (OP == 8) { OP32("v_perm_b32 %0, %0, %1, %2") }
        else if (OP == 9) { OP32("v_bfi_b32 %0, %1, %0, %2") }
        else if (OP == 10) { OP32("v_max_i32 %0, %0, %1") }
        else if (OP == 11) { OP32("v_mul_f32 %0, %0, %1") }
        else if (OP == 12) { OP32("v_and_or_b32 %0, %1, %2, %0") }
        else if (OP == 13) { OP32("v_min_f32 %0, %0, %1") }
        else if (OP == 14) { OP32("v_fma_f32 %0, %0, 2.0, %1 clamp") }
        else if (OP == 16) { OP32("v_xor_b32 %0, %0, %1") }
        else if (OP == 17) { OP32("v_med3_f32 %0, %0, %1, %2") }
        else if (OP == 18) { OP32("v_add_u32 %0, %0, %1") }
        else if (OP == 19) { OP32("v_alignbit_b32 %0, %1, %0, 31") }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 1024 + t] = a0 ^ a1 ^ a2 ^ a3;
}

template <int OP>
void run(unsigned* d_out, unsigned long long* d_clk, int cus, const char* what) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<cus, 1024>>>(d_out, d_clk, 1000);
    hipEventRecord(e0);
    k<OP><<<cus, 1024>>>(d_out, d_clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * cus);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < cus; i++) ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double g = ghz[cus / 2];
    const double cyc = ms * 1e-3 * g * 1e9 / ((double)iters * 32 * 4);                // SIMD cycles per wave-instruction (4 waves per SIMD)
    std::printf("%-34s %7.2f ms  clock %.2f GHz  %5.2f SIMD-cycles per wave-instruction\n", what, ms, g, cyc);
    std::fflush(stdout);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d_out; hipMalloc(&d_out, (size_t)cus * 1024 * 4);
    unsigned long long* d_clk; hipMalloc(&d_clk, (size_t)cus * 16);
    run<0>(d_out, d_clk, cus, "v_alignbit_b32 (acc, x, 31)");
    run<19>(d_out, d_clk, cus, "v_alignbit_b32 (x, acc, 31)");
    run<1>(d_out, d_clk, cus, "v_max_f32");
    run<13>(d_out, d_clk, cus, "v_min_f32");
    run<2>(d_out, d_clk, cus, "v_add_f32");
    run<11>(d_out, d_clk, cus, "v_mul_f32");
    run<3>(d_out, d_clk, cus, "v_fma_f32");
    run<14>(d_out, d_clk, cus, "v_fma_f32 clamp");
    run<4>(d_out, d_clk, cus, "v_max3_f32");
    run<17>(d_out, d_clk, cus, "v_med3_f32");
    run<5>(d_out, d_clk, cus, "v_and_b32");
    run<6>(d_out, d_clk, cus, "v_or_b32");
    run<16>(d_out, d_clk, cus, "v_xor_b32");
    run<18>(d_out, d_clk, cus, "v_add_u32");
    run<10>(d_out, d_clk, cus, "v_max_i32");
    run<7>(d_out, d_clk, cus, "v_lshl_or_b32");
    run<12>(d_out, d_clk, cus, "v_and_or_b32");
    run<8>(d_out, d_clk, cus, "v_perm_b32");
    run<9>(d_out, d_clk, cus, "v_bfi_b32");
    return 0;
}
