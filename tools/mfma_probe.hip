// Probe: fragment layout of v_mfma_f32_32x32x16_bf16 on gfx950, checked with exact integer data (asymmetric A and B).
// D[i][j] = sum_k A[i][k] * B[k][j];  lane l: r = l & 31, h = l >> 5
//   A fragment element e (0..7) = A[r][8h + e];  B fragment element e = B[8h + e][r]
//   D register g (0..15) of lane l = D[(g & 3) + 8 (g >> 2) + 4 h][r]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint16_t* A, const uint16_t* B, float* D) {   // A[32][16], B[16][32] as bf16 bit patterns
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    u16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = A[r * 16 + 8 * h + e]; b[e] = B[(8 * h + e) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int g = 0; g < 16; g++) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}

static uint16_t bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
int main() {
    std::vector<uint16_t> A(32 * 16), B(16 * 32);
    std::vector<float> Af(32 * 16), Bf(16 * 32), D(32 * 32), R(32 * 32, 0.0f);
    for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) { Af[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5); A[i * 16 + k] = bf(Af[i * 16 + k]); }
    for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) { Bf[k * 32 + j] = (float)((k * 5 + j * 2 + (j > 7)) % 13 - 6); B[k * 32 + j] = bf(Bf[k * 32 + j]); }
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int k = 0; k < 16; k++) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += D[i] != R[i];
    printf("mfma_f32_32x32x16_bf16 layout probe: %d of 1024 mismatches (D[3][5] = %g, expected %g)\n", bad, D[3 * 32 + 5], R[3 * 32 + 5]);
    return bad != 0;
}
