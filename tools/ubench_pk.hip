// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_pk_mul/add) on gfx950.
// Decides whether packing two sphere tests into one VALU instruction can pay.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float2v x0 = {a, b}, x1 = {b, a}, x2 = {a + 1, b}, x3 = {a, b + 1}, x4 = {a + 2, b}, x5 = {a, b + 2}, x6 = {a + 3, b}, x7 = {a, b + 3};
    const float2v m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    const float2v sp = {0.999f + 1e-9f * a, 1.001f + 1e-9f * b};     // wave-uniform -> SGPR pair in MODE 4
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // scalar fma: 16 independent v_fma_f32 per iteration
#define F2(V_) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5" : "+v"(V_.x), "+v"(V_.y) : "v"(m.x), "v"(c.x), "v"(m.y), "v"(c.y));
            F2(x0) F2(x1) F2(x2) F2(x3) F2(x4) F2(x5) F2(x6) F2(x7)
        } else if (MODE == 1) {   // packed fma: 8 v_pk_fma_f32 per iteration (same FLOPs)
#define P(V_) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(V_) : "v"(m), "v"(c));
            P(x0) P(x1) P(x2) P(x3) P(x4) P(x5) P(x6) P(x7)
        } else if (MODE == 3) {   // scalar fma with SGPR operands (as the sphere loop issues them)
#define G2(V_) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %3" : "+v"(V_.x), "+v"(V_.y) : "s"(a), "v"(c.x), "s"(b));
            G2(x0) G2(x1) G2(x2) G2(x3) G2(x4) G2(x5) G2(x6) G2(x7)
        } else if (MODE == 4) {   // packed fma with an SGPR pair operand
#define R(V_) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(V_) : "s"(sp), "v"(c));
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
        } else {                  // packed mul + packed add
#define Q(V_) asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(V_) : "v"(m), "v"(c));
            Q(x0) Q(x1) Q(x2) Q(x3) Q(x4) Q(x5) Q(x6) Q(x7)
        }
    }
    float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE>
void run(const char* name, double flop_per_iter_lane) {
    float* out;
    const int blocks = 256 * 8, iters = 200000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 1000, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters, 1.0f, 2.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double flop = flop_per_iter_lane * iters * blocks * 256.0;
    printf("%-28s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32 x16", 32.0);
    run<1>("v_pk_fma_f32 x8", 32.0);
    run<2>("v_pk_mul_f32+v_pk_add_f32 x8", 32.0);
    run<3>("v_fma_f32 x16, SGPR src", 32.0);
    run<4>("v_pk_fma_f32 x8, SGPR pair", 32.0);
    run<0>("v_fma_f32 x16 (again)", 32.0);
    run<1>("v_pk_fma_f32 x8 (again)", 32.0);
    return 0;
}
