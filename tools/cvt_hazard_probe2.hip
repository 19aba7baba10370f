// cvt_hazard_probe2.hip — which neighbourhood makes `v_cvt_pk_bf16_f32 ; <VALU reader in the next issue slot>` read a stale register?
// (profiles/r03_cvt_hazard.md; round 2's probe, tools/cvt_hazard_probe.hip, found no mismatch in 5e9 conversions with all 16 waves of a CU
// running the same conversion stream.)  hipcc inserts `s_nop 0` between a TRANSCENDENTAL op (v_sqrt_f32, v_rcp_f32, v_exp_f32 ...) and a vector-ALU
// reader of its result — the gfx940+ "trans use" forwarding hazard — and nothing between v_cvt_pk_bf16_f32 and its reader.  This probe runs the
// pair in a 1024-thread workgroup per CU (4 waves per SIMD): waves 0-3 are VICTIMS (the pair, preceded by a selectable instruction of the same
// wave), waves 4-15 AGGRESSORS (a selectable instruction stream on the same SIMDs), and counts results that differ from integer rounding.
//   hipcc --offload-arch=gfx950 -O3 -o tools/cvt_hazard_probe2 tools/cvt_hazard_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bf16_rn(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// victim prefixes (same wave, the slot(s) right before the conversion)
#define PRE_NONE   ""
#define PRE_SQRT   "v_sqrt_f32 %5, %6\n\t"
#define PRE_RCP2   "v_rcp_f32 %5, %6\n\tv_sqrt_f32 %7, %6\n\t"
#define PRE_EXP    "v_exp_f32 %5, %6\n\t"
#define PRE_PKADD  "v_pk_add_f32 %8, %8, %8\n\t"
#define PRE_FMA    "v_fma_f32 %5, %6, %6, %6\n\t"
#define PAIR(PRE, GAP)                                                                                                 \
    asm volatile(PRE "v_cvt_pk_bf16_f32 %0, %3, %4\n\t" GAP "v_lshlrev_b32 %1, 16, %0\n\tv_and_b32 %2, 0xffff0000, %0"   \
                 : "=&v"(packed), "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y), "v"(t0), "v"(z), "v"(t1), "v"(pk) : "memory")

template <int VICTIM, int AGGR, int GAP>
__global__ __launch_bounds__(1024) void probe(uint32_t iters, unsigned long long* mismatches, float* sink) {
    __shared__ float lds[4096];
    const uint32_t tid = threadIdx.x, wave = tid / 64u;
    for (uint32_t i = tid; i < 4096; i += 1024) lds[i] = (float)i;
    __syncthreads();
    uint32_t state = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
    if (wave < 4) {
        unsigned long long bad = 0;
        float t0 = 0.0f, t1 = 0.0f;
        double pk = 1.0;
        bf16x8 a = { 0 }, b = { 0 };
        f32x4 acc = { 0 };
        for (uint32_t i = 0; i < iters; i++) {
            state = state * 1664525u + 1013904223u;
            const float x = __uint_as_float(0x3F800000u | (state >> 9)), y = __uint_as_float(0x40000000u | ((state * 7u) >> 9));
            const float z = __uint_as_float(0x3F000000u | ((state * 13u) >> 9));
            uint32_t packed, lo, hi;
            if (VICTIM == 6) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); a[0] = (__bf16)z; }
            if (VICTIM == 7) t0 = lds[(state >> 20) & 4095u];
            if (GAP == 0) {
                if (VICTIM == 0 || VICTIM == 6 || VICTIM == 7) PAIR(PRE_NONE, "");
                else if (VICTIM == 1) PAIR(PRE_SQRT, "");
                else if (VICTIM == 2) PAIR(PRE_RCP2, "");
                else if (VICTIM == 3) PAIR(PRE_EXP, "");
                else if (VICTIM == 4) PAIR(PRE_PKADD, "");
                else PAIR(PRE_FMA, "");
            } else {
                if (VICTIM == 0 || VICTIM == 6 || VICTIM == 7) PAIR(PRE_NONE, "s_nop 0\n\t");
                else if (VICTIM == 1) PAIR(PRE_SQRT, "s_nop 0\n\t");
                else if (VICTIM == 2) PAIR(PRE_RCP2, "s_nop 0\n\t");
                else if (VICTIM == 3) PAIR(PRE_EXP, "s_nop 0\n\t");
                else if (VICTIM == 4) PAIR(PRE_PKADD, "s_nop 0\n\t");
                else PAIR(PRE_FMA, "s_nop 0\n\t");
            }
            bad += (lo != (bf16_rn(x) << 16)) + (hi != (bf16_rn(y) << 16));
        }
        sink[blockIdx.x * 1024u + tid] = t0 + t1 + (float)pk + acc[0];
        if (bad) atomicAdd(mismatches, bad);
    } else {
        // aggressors: 3 per SIMD, 1.5 x the victim's trip count of a denser loop, so they are busy for the victims' whole run
        float u = __uint_as_float(0x3F800000u | (state >> 9)), v = 1.25f, w = 0.0f;
        bf16x8 a = { 0 }, b = { 0 };
        f32x4 acc = { 0 };
        for (uint32_t i = 0; i < iters; i++) {
            if (AGGR == 1) { asm volatile("v_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0" : "+v"(u), "+v"(v)); }
            else if (AGGR == 2) { asm volatile("v_sqrt_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_sqrt_f32 %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(u), "+v"(v)); }
            else if (AGGR == 3) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0); }
            else if (AGGR == 4) { w += lds[(i * 17u + tid) & 4095u]; w += lds[(i * 29u + tid * 3u) & 4095u]; }
            else if (AGGR == 5) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %1, %1, %0" : "+v"(u), "+v"(v)); }
            else if (AGGR == 6) { asm volatile("v_exp_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_sin_f32 %0, %0\n\tv_cos_f32 %1, %1" : "+v"(u), "+v"(v)); }
        }
        sink[blockIdx.x * 1024u + tid] = u + v + w + acc[0];
    }
}

static unsigned long long* d_bad;
static float* d_sink;
template <int VICTIM, int AGGR, int GAP>
unsigned long long run(int cus, uint32_t iters) {
    hipMemset(d_bad, 0, 8);
    probe<VICTIM, AGGR, GAP><<<cus, 1024>>>(iters, d_bad, d_sink);
    unsigned long long bad = 0;
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    return bad;
}
template <int VICTIM>
void row(int cus, uint32_t iters, const char* what) {
    const unsigned long long r[7] = { run<VICTIM, 0, 0>(cus, iters), run<VICTIM, 1, 0>(cus, iters), run<VICTIM, 2, 0>(cus, iters), run<VICTIM, 3, 0>(cus, iters),
                                      run<VICTIM, 4, 0>(cus, iters), run<VICTIM, 5, 0>(cus, iters), run<VICTIM, 6, 0>(cus, iters) };
    const unsigned long long g = run<VICTIM, 2, 1>(cus, iters) + run<VICTIM, 6, 1>(cus, iters) + run<VICTIM, 3, 1>(cus, iters);
    std::printf("%-44s %10llu %10llu %10llu %10llu %10llu %10llu %10llu   | with s_nop 0 (sqrt/rcp + exp/log/sin/cos + mfma aggressors): %llu\n",
                what, r[0], r[1], r[2], r[3], r[4], r[5], r[6], g);
    std::fflush(stdout);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipMalloc(&d_bad, 8);
    hipMalloc(&d_sink, (size_t)cus * 1024 * 4);
    const uint32_t iters = 200000;                                   // x 256 CUs x 4 victim waves x 64 lanes x 2 halves = 2.6e10 checked halves per cell
    std::printf("mismatching halves of v_cvt_pk_bf16_f32 results read by the NEXT instruction; %u conversions per victim lane, %d CUs x 4 victim waves\n", iters, cus);
    std::printf("%-44s %10s %10s %10s %10s %10s %10s %10s\n", "victim prefix \\ aggressor waves", "idle", "add/mul", "sqrt/rcp", "mfma", "lds", "cvt_pk", "exp/log/sin");
    row<0>(cus, iters, "none");
    row<1>(cus, iters, "v_sqrt_f32 (independent) right before");
    row<2>(cus, iters, "v_rcp_f32 + v_sqrt_f32 right before");
    row<3>(cus, iters, "v_exp_f32 right before");
    row<4>(cus, iters, "v_pk_add_f32 right before");
    row<5>(cus, iters, "v_fma_f32 right before");
    row<6>(cus, iters, "mfma 16x16x32 earlier in the trip");
    row<7>(cus, iters, "ds_read earlier in the trip");
    return 0;
}
