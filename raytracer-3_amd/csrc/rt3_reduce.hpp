// rt3_reduce.hpp — k_accumulate / k_resolve (the reduce pass) and the arithmetic probe kernel of the tests
// Part of rt3_device.hip (one translation unit, gfx950 only); included from there, in this order.
#pragma once

namespace {

// reduce pass (what reduce_v1.glsl:66-76 was meant to be): samples are summed per pixel in sample order.  `accum` persists between
// launches (sample batches of one render, and the calls of a progressive render: rt3_render_path_range); VAR also keeps the sums of
// squares sq = fma(L, L, sq), in the same order (RT3_FLAG_VARIANCE).
template <bool VAR>
__global__ __launch_bounds__(kBlock) void k_accumulate(const Rgb* __restrict__ rad, float4* __restrict__ accum, float4* __restrict__ accum_sq,
                                                      uint32_t npix, uint32_t ns, int first) {
    const uint32_t pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= npix) return;
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 a = first ? zero : accum[pix];
    float4 q = (VAR && !first) ? accum_sq[pix] : zero;
    for (uint32_t s = 0; s < ns; s++) {
        const Rgb r = rad[(size_t)s * npix + pix];
        a.x = a.x + r.r; a.y = a.y + r.g; a.z = a.z + r.b;
        if (VAR) { q.x = fma_(r.r, r.r, q.x); q.y = fma_(r.g, r.g, q.y); q.z = fma_(r.b, r.b, q.z); }
    }
    accum[pix] = a;
    if (VAR) accum_sq[pix] = q;
}
__global__ __launch_bounds__(kBlock) void k_resolve(const float4* __restrict__ accum, uint32_t npix, uint32_t spp,
                                                   uint32_t flags, uint32_t* __restrict__ out) {
    const uint32_t pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= npix) return;
    const float4 a = accum[pix];
    const float n = (float)spp;
    float r = a.x / n, g = a.y / n, b = a.z / n;
    if (flags & RT3_FLAG_GAMMA2) {
        r = r > 0.0f ? __builtin_sqrtf(r) : 0.0f;
        g = g > 0.0f ? __builtin_sqrtf(g) : 0.0f;
        b = b > 0.0f ? __builtin_sqrtf(b) : 0.0f;
    }
    out[pix] = pack_pixel(r, g, b);
}

// device arithmetic probes for tests/test_gpu_arith.py
__global__ void k_debug_arith(const float* a, const float* b, uint32_t n, float* div, float* sq, float* fm,
                              float* cs, float* sn, float* sk, uint32_t* pk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    div[i] = a[i] / b[i];
    sq[i] = __builtin_sqrtf(__builtin_fabsf(a[i]));
    fm[i] = fma_(a[i], b[i], a[i]);
    const float u = u01(__float_as_uint(a[i]));
    sincos2pi(u, cs[i], sn[i]);
    float r, g, bl;
    sky(a[i], b[i], -2.0f, r, g, bl);
    sk[3 * i] = r; sk[3 * i + 1] = g; sk[3 * i + 2] = bl;
    pk[i] = pack_pixel(a[i], b[i], u);
}

}  // namespace
