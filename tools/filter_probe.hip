// Probe: the matrix-core candidate filter on the real hardware (companion of tools/filter_model.py), in its three forms:
// 32x32x16 K = 64 (k_trace_mfma), 16x16x32 K = 64 (faces in the tiled kernels, Mode R), 16x16x32 K = 32 (spheres in the tiled kernels).
// Runs the product's own operand builders (bound_frag_row / build_ray_operands) and MFMA chain on random and deliberately grazing
// (ray, sphere) pairs, then reports on the host (a) false negatives against the exact f32 rule of the kernels and (b) the worst
// |filter - (exact discriminant + margin)| relative to the margin.  Build + run on an MI355X:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -o tools/filter_probe tools/filter_probe.hip raytracer-3_amd/csrc/rt3_host.cpp
#include "../raytracer-3_amd/csrc/rt3_device.hip"
#include <random>

namespace {
__global__ void k_probe(const float* __restrict__ rays, const u32x4* __restrict__ frags, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const float* r = rays + ((size_t)t * 64 + lane) * 6;
    RayOperands R;
    build_ray_operands(r[0], r[1], r[2], r[3], r[4], r[5], true, R);
    const u32x4* fr = frags + (size_t)t * 256 + lane;
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, fr[0]), a1 = __builtin_bit_cast(bf16x8, fr[64]);
    const bf16x8 a2 = __builtin_bit_cast(bf16x8, fr[128]), a3 = __builtin_bit_cast(bf16x8, fr[192]);
    for (int S = 0; S < 2; S++) {
        const f32x16 zero = { 0 };
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, R.b[S][0]), zero, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, R.b[S][0]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(bf16x8, R.b[S][1]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, __builtin_bit_cast(bf16x8, R.b[S][2]), d, 0, 0, 0);
        for (int g = 0; g < 16; g++) out[(((size_t)t * 2 + S) * 64 + lane) * 16 + g] = d[g];
    }
}

// the same pairs through the 16x16x32 form of the tiled kernels: out16[t][lane][32] = the lane's 32 results of the row block, in the
// bit order of its candidate word (8 G + 4 h + j <-> ray lane 16 G + c, row 16 h + 4 g + j)
__global__ void k_probe16(const float* __restrict__ rays, const u32x4* __restrict__ frags16, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const float* r = rays + ((size_t)t * 64 + lane) * 6;
    RayOperands16 R;
    build_ray_operands16(r[0], r[1], r[2], r[3], r[4], r[5], true, R);
    const u32x4* fr = frags16 + (size_t)t * 256 + lane;
    const u32x4 a[2][2] = { { fr[0], fr[64] }, { fr[128], fr[192] } };
    for (int G = 0; G < 4; G++)
        for (int h = 0; h < 2; h++) {
            const f32x4v zero = { 0.0f, 0.0f, 0.0f, 0.0f };
            f32x4v d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[h][0]), __builtin_bit_cast(bf16x8, R.b[G][0]), zero, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[h][1]), __builtin_bit_cast(bf16x8, R.b[G][1]), d, 0, 0, 0);
            for (int j = 0; j < 4; j++) out[((size_t)t * 64 + lane) * 32 + 8 * G + 4 * h + j] = d[j];
        }
}
// ... and through the K = 32 form of the sphere pass (coordinates about a centre, eps = kFilterEps32): same bit order
__global__ void k_probe32(const float* __restrict__ rays, const u32x4* __restrict__ frags32, float cx, float cy, float cz, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const float* r = rays + ((size_t)t * 64 + lane) * 6;
    RayOperands32 R;
    build_ray_operands32(r[0] - cx, r[1] - cy, r[2] - cz, r[3], r[4], r[5], true, R);
    const u32x4* fr = frags32 + (size_t)t * 128 + lane;
    const u32x4 a[2] = { fr[0], fr[64] };
    for (int G = 0; G < 4; G++)
        for (int h = 0; h < 2; h++) {
            const f32x4v zero = { 0.0f, 0.0f, 0.0f, 0.0f };
            const f32x4v d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[h]), __builtin_bit_cast(bf16x8, R.b[G]), zero, 0, 0, 0);
            for (int j = 0; j < 4; j++) out[((size_t)t * 64 + lane) * 32 + 8 * G + 4 * h + j] = d[j];
        }
}

struct V { double x, y, z; };
V unit(V v) { const double n = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return { v.x / n, v.y / n, v.z / n }; }
}  // namespace

int main() {
    const int T = 8192;
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    std::vector<float> rays((size_t)T * 64 * 6), sph((size_t)T * 32 * 4);
    std::vector<uint32_t> frags((size_t)T * 256 * 4), frags16((size_t)T * 256 * 4), frags32((size_t)T * 128 * 4);
    const float centre[3] = { 1.5f, -0.25f, -3.0f };                // the K = 32 form works about a centre; any point near the data will do here
    const char* names[4] = { "book scale", "ground r=1000", "stress scale", "book scale x1000" };
    for (int t = 0; t < T; t++) {
        const int mode = t % 4;
        for (int b = 0; b < 32; b++) {
            float* s = &sph[((size_t)t * 32 + b) * 4];
            if (mode == 0 || mode == 3) { s[0] = (float)(U(rng) * 22 - 11); s[1] = 0.2f; s[2] = (float)(U(rng) * 22 - 11); s[3] = 0.2f; }
            else if (mode == 1) { s[0] = 0; s[1] = -1000; s[2] = 0; s[3] = 1000; }
            else { s[0] = (float)(U(rng) * 100 - 50); s[1] = (float)(0.2 + U(rng) * 19.8); s[2] = (float)(-100 * U(rng)); s[3] = (float)(0.05 + 0.35 * U(rng)); }
            if (mode == 3) for (int k = 0; k < 4; k++) s[k] *= 1000.0f;
        }
        for (int l = 0; l < 64; l++) {
            // a ray grazing sphere (l % 32): through a point at r (1 + delta) from the centre, perpendicular to the radius
            const float* s = &sph[((size_t)t * 32 + l % 32) * 4];
            const V tn = unit({ N(rng), N(rng), N(rng) });
            const double mags[5] = { 1e-3, 1e-5, 1e-6, 1e-7, 0.3 };
            const double delta = (2 * U(rng) - 1) * mags[(t / 4 + l) % 5];
            const V p = { s[0] + tn.x * s[3] * (1 + delta), s[1] + tn.y * s[3] * (1 + delta), s[2] + tn.z * s[3] * (1 + delta) };
            V w = { N(rng), N(rng), N(rng) };
            const double wt = w.x * tn.x + w.y * tn.y + w.z * tn.z;
            w = unit({ w.x - wt * tn.x, w.y - wt * tn.y, w.z - wt * tn.z });
            const double back = (mode == 1 ? 0.0 + 30 * U(rng) : 0.5 + 30 * U(rng)) * (mode == 3 ? 1000.0 : 1.0);
            const float o[3] = { (float)(p.x - w.x * back), (float)(p.y - w.y * back), (float)(p.z - w.z * back) };
            V d = unit({ p.x - o[0], p.y - o[1], p.z - o[2] });
            if (back == 0.0) d = unit({ N(rng), std::fabs(N(rng)) * 0.01, N(rng) });
            float df[3] = { (float)d.x, (float)d.y, (float)d.z };
            const float inv = 1.0f / std::sqrt(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]);
            float* r = &rays[((size_t)t * 64 + l) * 6];
            r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = df[0] * inv; r[4] = df[1] * inv; r[5] = df[2] * inv;
        }
        for (int b = 0; b < 32; b++) {
            const float* s = &sph[((size_t)t * 32 + b) * 4];
            uint32_t fr[4][2][4];
            const double c2 = (double)s[0] * s[0] + (double)s[1] * s[1] + (double)s[2] * s[2], r2 = (double)s[3] * s[3];
            bound_frag_row(s[0], s[1], s[2], filter_kj(c2, r2), fr);
            for (int q = 0; q < 4; q++)
                for (int hh = 0; hh < 2; hh++)
                    std::memcpy(&frags[(((size_t)t * 4 + q) * 64 + hh * 32 + frag_row_of(b)) * 4], fr[q][hh], 16);
            uint32_t f16[2][4][4];
            bound_frag16_row(s[0], s[1], s[2], filter_kj(c2, r2), f16);
            for (uint32_t q = 0; q < 2; q++)
                for (uint32_t g = 0; g < 4; g++) std::memcpy(&frags16[frag16_index((uint32_t)t, (uint32_t)b, q, g) * 4], f16[q][g], 16);
            uint32_t f32r[4][4];
            const float scale32 = (mode == 3 ? 1000.0f : 1.0f);
            const float rx = (float)((double)s[0] - centre[0] * scale32), ry = (float)((double)s[1] - centre[1] * scale32), rz = (float)((double)s[2] - centre[2] * scale32);
            bound_frag32_row(rx, ry, rz, filter_kj32((double)rx * rx + (double)ry * ry + (double)rz * rz, r2), f32r);
            for (uint32_t g = 0; g < 4; g++) std::memcpy(&frags32[frag32_index((uint32_t)t, (uint32_t)b, g) * 4], f32r[g], 16);
        }
    }
    float *d_rays, *d_out; u32x4* d_frags;
    std::vector<float> out((size_t)T * 2 * 64 * 16);
    if (hipMalloc(&d_rays, rays.size() * 4) || hipMalloc(&d_frags, frags.size() * 4) || hipMalloc(&d_out, out.size() * 4)) { printf("no GPU memory\n"); return 2; }
    hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_frags, frags.data(), frags.size() * 4, hipMemcpyHostToDevice);
    k_probe<<<T, 64>>>(d_rays, d_frags, d_out);
    if (hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }

    // the 16x16x32 forms on the same pairs (K = 32: one launch per coordinate scale, the centre scales with the scene)
    std::vector<float> out16((size_t)T * 64 * 32), out32((size_t)T * 64 * 32);
    {
        u32x4 *d_frags16, *d_frags32; float *d_out16, *d_out32;
        if (hipMalloc(&d_frags16, frags16.size() * 4) || hipMalloc(&d_out16, out16.size() * 4) || hipMalloc(&d_frags32, frags32.size() * 4) ||
            hipMalloc(&d_out32, out32.size() * 4)) { printf("no GPU memory\n"); return 2; }
        hipMemcpy(d_frags16, frags16.data(), frags16.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_frags32, frags32.data(), frags32.size() * 4, hipMemcpyHostToDevice);
        k_probe16<<<T, 64>>>(d_rays, d_frags16, d_out16);
        if (hipMemcpy(out16.data(), d_out16, out16.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel 16 failed\n"); return 2; }
        std::vector<float> tmp(out32.size());
        for (int pass = 0; pass < 2; pass++) {
            const float sc = pass ? 1000.0f : 1.0f;
            k_probe32<<<T, 64>>>(d_rays, d_frags32, centre[0] * sc, centre[1] * sc, centre[2] * sc, d_out32);
            if (hipMemcpy(tmp.data(), d_out32, tmp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel 32 failed\n"); return 2; }
            for (int t = 0; t < T; t++) if ((t % 4 == 3) == (pass == 1)) std::memcpy(&out32[(size_t)t * 64 * 32], &tmp[(size_t)t * 64 * 32], 64 * 32 * 4);
        }
    }
    long fn_total = 0;
    for (int form = 0; form < 3; form++)
    for (int mode = 0; mode < 4; mode++) {
        long pairs = 0, exact = 0, filt = 0, fn = 0;
        double worst = 0.0;
        for (int t = mode; t < T; t += 4)
            for (int S = 0; S < 2; S++)
                for (int l = 0; l < 64; l++)
                    for (int g = 0; g < 16; g++) {
                        int b, ray;
                        float f;
                        if (form == 0) { const int w = l >> 5; b = 16 * w + 15 - g; ray = 32 * S + (l & 31); f = out[(((size_t)t * 2 + S) * 64 + l) * 16 + g]; }
                        else { const int bit = 16 * S + g; b = 16 * ((bit >> 2) & 1) + 4 * (l >> 4) + (bit & 3); ray = 16 * (bit >> 3) + (l & 15);
                               f = (form == 1 ? out16 : out32)[((size_t)t * 64 + l) * 32 + bit]; }
                        const float* s = &sph[((size_t)t * 32 + b) * 4];
                        const float* r = &rays[((size_t)t * 64 + ray) * 6];
                        // the kernels' exact rule
                        const float cx = s[0] - r[0], cy = s[1] - r[1], cz = s[2] - r[2];
                        const float h = std::fmaf(cz, r[5], std::fmaf(cy, r[4], cx * r[3]));
                        const float c = std::fmaf(cz, cz, std::fmaf(cy, cy, std::fmaf(cx, cx, -(s[3] * s[3]))));
                        const float disc = std::fmaf(h, h, -c);
                        const bool cand = (c < 0.0f) | ((disc > 0.0f) & (h > 0.0f));
                        const double ocx = (double)s[0] - r[0], ocy = (double)s[1] - r[1], ocz = (double)s[2] - r[2];
                        const double hd = ocx * r[3] + ocy * r[4] + ocz * r[5], dd = (double)r[3] * r[3] + (double)r[4] * r[4] + (double)r[5] * r[5];
                        const double scale = (double)s[0] * s[0] + (double)s[1] * s[1] + (double)s[2] * s[2] + (double)s[3] * s[3] +
                                             (double)r[0] * r[0] + (double)r[1] * r[1] + (double)r[2] * r[2];
                        double eps = kFilterEps, sc = scale;
                        if (form == 2) {                            // eps32, and |C|^2 + |o|^2 about the centre
                            const double k = (mode == 3 ? 1000.0 : 1.0), ax = centre[0] * k, ay = centre[1] * k, az = centre[2] * k;
                            eps = kFilterEps32;
                            sc = (s[0] - ax) * (s[0] - ax) + (s[1] - ay) * (s[1] - ay) + (s[2] - az) * (s[2] - az) + (double)s[3] * s[3] +
                                 (r[0] - ax) * (r[0] - ax) + (r[1] - ay) * (r[1] - ay) + (r[2] - az) * (r[2] - az);
                        }
                        const double truth = hd * hd - (ocx * ocx + ocy * ocy + ocz * ocz) * dd + (double)s[3] * s[3] + eps * sc;
                        worst = std::max(worst, std::fabs((double)f - truth) / (eps * sc));
                        pairs++; exact += cand; filt += !(f < 0.0f) && !std::signbit(f); fn += cand && std::signbit(f);
                    }
        printf("%s %-18s pairs %9ld  exact candidates %8ld  filter candidates %8ld  FALSE NEGATIVES %ld  worst error/margin %.4f\n",
               form == 0 ? "32x32x16 K=64" : form == 1 ? "16x16x32 K=64" : "16x16x32 K=32", names[mode], pairs, exact, filt, fn, worst);
        fn_total += fn;
    }
    printf("total false negatives: %ld\n", fn_total);
    return fn_total != 0;
}
