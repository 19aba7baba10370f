// rt3_scene_kernels.hpp — device-side scene assembly: tessellated spheres and the merge into the render layout (k_commit_mesh)
// Part of rt3_device.hip (one translation unit, gfx950 only); included from there, in this order.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------------
// Device-side scene assembly: HIP equivalents of the reference's pre-render shaders and of the merge
// ------------------------------------------------------------------------------------------------------
struct SphereGen { float cx, cy, cz, radius; uint32_t m, p; float r, g, b; uint32_t face_offset, vertex_offset; };

// compute_point (Sphere.cpp:69-79 / pre_render_sphere_v2_vertices.glsl:77-83).  The CPU form is followed (double
// trig on a float ratio, rounded to float per component), not the shader's float trig, so that a device-tessellated
// sphere equals a host-tessellated one.
__device__ __forceinline__ float4 sphere_point(const SphereGen& s, float fx, float fy) {
    const double ty = M_PI * (double)(fy / (float)(s.p - 1));
    const double tx = 2 * M_PI * (double)(fx / (float)s.m);
    const float ux = (float)(sin(ty) * cos(tx)), uy = (float)cos(ty), uz = (float)(sin(ty) * sin(tx));
    return make_float4(s.cx + s.radius * ux, s.cy + s.radius * uy, s.cz + s.radius * uz, 0.0f);
}

// pre_render_sphere_v2_vertices.glsl:88-113: one thread per (meridian x, parallel y)
__global__ void k_prerender_sphere_vertices(SphereGen s, float4* __restrict__ verts) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x == 0 && y == 0) verts[s.vertex_offset] = sphere_point(s, 0.0f, 0.0f);
    else if (x < s.m && y > 0 && y < s.p - 1) verts[s.vertex_offset + 1 + (y - 1) * s.m + x] = sphere_point(s, (float)x, (float)y);
    else if (x == 0 && y == s.p - 1) verts[s.vertex_offset + 1 + (y - 1) * s.m] = sphere_point(s, 0.0f, (float)y);
}

__device__ __forceinline__ void store_face(rt3_gface* f, uint32_t a, uint32_t b, uint32_t c, float4 pa, float4 pb, float4 pc,
                                           const SphereGen& s) {
    // normal = normalize(cross(c - a, b - a)) with glm's evaluation order; colour = colour * |n . (0,0,-1)| (Sphere.cpp:153-155)
    const float ex = pc.x - pa.x, ey = pc.y - pa.y, ez = pc.z - pa.z, fx = pb.x - pa.x, fy = pb.y - pa.y, fz = pb.z - pa.z;
    const float nx = ey * fz - fy * ez, ny = ez * fx - fz * ex, nz = ex * fy - fx * ey;
    const float inv = 1.0f / __builtin_sqrtf(dot3(nx, ny, nz, nx, ny, nz));
    const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
    const float shade = __builtin_fabsf(ux * 0.0f + uy * 0.0f + uz * -1.0f);
    f->v1 = a; f->v2 = b; f->v3 = c; f->_pad0 = 0;
    f->normal[0] = ux; f->normal[1] = uy; f->normal[2] = uz; f->_pad1 = 0;
    f->color[0] = s.r * shade; f->color[1] = s.g * shade; f->color[2] = s.b * shade; f->_pad2 = 0;
}

// pre_render_sphere_v2_faces.glsl:83-194: one thread per (x, y >= 1); reads the vertices the first kernel wrote
__global__ void k_prerender_sphere_faces(SphereGen s, rt3_gface* __restrict__ faces, const float4* __restrict__ verts) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (x >= s.m || y >= s.p) return;
    const uint32_t xm1 = x > 0 ? x - 1 : s.m - 1, vo = s.vertex_offset;
    rt3_gface* out = faces + s.face_offset;
    if (y == 1) {
        const uint32_t a = vo, b = vo + 1 + xm1, c = vo + 1 + x;
        store_face(out + x, a, b, c, verts[a], verts[b], verts[c], s);
    } else if (y < s.p - 1) {
        const uint32_t base = s.m + 2 * (y - 2) * s.m;
        const uint32_t p1 = vo + 1 + (y - 2) * s.m + xm1, p2 = vo + 1 + (y - 2) * s.m + x;
        const uint32_t p3 = vo + 1 + (y - 1) * s.m + xm1, p4 = vo + 1 + (y - 1) * s.m + x;
        store_face(out + base + 2 * x, p1, p3, p4, verts[p1], verts[p3], verts[p4], s);
        store_face(out + base + 2 * x + 1, p1, p2, p4, verts[p1], verts[p2], verts[p4], s);
    } else {
        const uint32_t base = s.m + 2 * (y - 2) * s.m;
        const uint32_t a = vo + 1 + (y - 1) * s.m, b = vo + 1 + (y - 2) * s.m + xm1, c = vo + 1 + (y - 2) * s.m + x;
        store_face(out + base + x, a, b, c, verts[a], verts[b], verts[c], s);
    }
}

// De-indexes the merged GFace[] / vec4[] into what the render kernels read: 4 float4 per face (n + plane distance, p1, p2,
// p3), the bounding sphere of §5.1, the material.  Entries [n_faces, n_pad) of `bound` become never-hit records.
// The box of the finite vertex coordinates, as order-preserving integers: box[0..2] = min x, y, z; box[3..5] = max (start: ~0 / 0).
__host__ __device__ inline uint32_t ordered_bits(float f) { const uint32_t b = __builtin_bit_cast(uint32_t, f); return (b >> 31) ? ~b : (b | 0x80000000u); }
__host__ __device__ inline float ordered_float(uint32_t k) { return __builtin_bit_cast(float, (k >> 31) ? (k & 0x7FFFFFFFu) : ~k); }
// Centre of that box (0 for an empty or unusable one): what the faces' filter coordinates are taken about.
__host__ __device__ inline void box_centre(const uint32_t box[6], float out[3]) {
    for (int a = 0; a < 3; a++) {
        const float lo = ordered_float(box[a]), hi = ordered_float(box[3 + a]);
        const float c = (float)(0.5 * ((double)lo + (double)hi));
        out[a] = (box[a] <= box[3 + a] && c - c == 0.0f) ? c : 0.0f;
    }
}
__global__ void k_vertex_box(const float4* __restrict__ verts, uint32_t n_verts, uint32_t* __restrict__ box) {
    float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_verts; i += gridDim.x * blockDim.x) {
        const float4 v = verts[i];
        const float c[3] = { v.x, v.y, v.z };
        for (int a = 0; a < 3; a++)
            if (c[a] - c[a] == 0.0f) { lo[a] = fminf(lo[a], c[a]); hi[a] = fmaxf(hi[a], c[a]); }      // finite values only
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
        if ((threadIdx.x & 63u) == 0u && lo[a] <= hi[a]) { atomicMin(box + a, ordered_bits(lo[a])); atomicMax(box + 3 + a, ordered_bits(hi[a])); }
    }
}

__global__ void k_commit_mesh(const rt3_gface* __restrict__ faces, const float4* __restrict__ verts, uint32_t n_faces, uint32_t n_pad,
                              uint32_t n_verts, const rt3_material* __restrict__ mats, float4* __restrict__ tri, float4* __restrict__ bound,
                              float4* __restrict__ mat, uint32_t* __restrict__ kind, uint32_t* __restrict__ error_flag,
                              u32x4* __restrict__ frag, uint32_t n_frag_rows, const uint32_t* __restrict__ box, float centre_scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float centre[3];                                                // centre_scale: 1 for the path tracer; 0.5: Mode R's fragments (rt3_device.hip),
    box_centre(box, centre);                                        //   a later launch that writes `frag` only
    for (int a = 0; a < 3; a++) centre[a] *= centre_scale;
    const bool all = centre_scale == 1.0f;
    // matrix-filter fragments (16x16x32 form): row i of block i/32, both K halves, all four lane groups (padding rows: never candidates)
    auto write_frag = [&](float cx, float cy, float cz, float kj) {
        if (i >= n_frag_rows) return;
#if RT3_FACE_K32
        uint32_t fr[4][4];
        bound_frag32_row(cx, cy, cz, kj, fr);
        for (uint32_t g = 0; g < 4; g++) frag[frag32_index(i / 32, i % 32, g)] = u32x4{ fr[g][0], fr[g][1], fr[g][2], fr[g][3] };
#else
        uint32_t fr[2][4][4];
        bound_frag16_row(cx, cy, cz, kj, fr);
        for (uint32_t q = 0; q < 2; q++)
            for (uint32_t g = 0; g < 4; g++)
                frag[frag16_index(i / 32, i % 32, q, g)] = u32x4{ fr[q][g][0], fr[q][g][1], fr[q][g][2], fr[q][g][3] };
#endif
    };
    if (i >= n_pad && i >= n_frag_rows) return;
    if (i >= n_faces) { if (all && i < n_pad) bound[i] = kPadSphere; write_frag(0.0f, 0.0f, 0.0f, kNeverCandidate); return; }
    const rt3_gface f = faces[i];
    if (f.v1 >= n_verts || f.v2 >= n_verts || f.v3 >= n_verts) { if (all) { atomicOr(error_flag, 1u); bound[i] = kPadSphere; } write_frag(0.0f, 0.0f, 0.0f, kNeverCandidate); return; }
    const float4 p1 = verts[f.v1], p2 = verts[f.v2], p3 = verts[f.v3];
    if (all) {
        tri[4 * (size_t)i] = make_float4(f.normal[0], f.normal[1], f.normal[2], dot3(f.normal[0], f.normal[1], f.normal[2], p1.x, p1.y, p1.z));
        tri[4 * (size_t)i + 1] = make_float4(p1.x, p1.y, p1.z, 0.0f);
        tri[4 * (size_t)i + 2] = make_float4(p2.x, p2.y, p2.z, 0.0f);
        tri[4 * (size_t)i + 3] = make_float4(p3.x, p3.y, p3.z, 0.0f);
    }
    // Bounding sphere of the region in which the REFERENCE'S test can report a hit (it must never be smaller: the exact test runs only
    // on the faces whose bound the ray's line meets).  That region is the triangle p1, p2', p3' in the plane through p1 normal to the
    // STORED normal, p' = p shifted along that normal (the three edge functions only see the projection along it) — the triangle
    // itself when the normal is perpendicular to it, which nothing in the interface guarantees.  Centroid + largest distance to the
    // vertices and their projections, in double, inflated by 0.1 % + 1e-5 (1 + max |coordinate|), r^2 rounded up.
    // Faces whose edge functions vanish identically have NO bounded hit region — two coincident vertices (the zero edge's function is
    // -0 >= 0 for every point: three coincident vertices make the test an infinite plane, which is what tiny faces far from the origin
    // collapse to in f32) or three collinear ones — and non-finite input cannot be bounded either: they are always candidates and the
    // exact test alone decides, as in the reference (found by tools/fuzz_filter.py).
    const double cx = ((double)p1.x + p2.x + p3.x) / 3.0, cy = ((double)p1.y + p2.y + p3.y) / 3.0, cz = ((double)p1.z + p2.z + p3.z) / 3.0;
    double r2 = 0.0, big = 0.0;
    const float4 ps[3] = { p1, p2, p3 };
    const double nx = f.normal[0], ny = f.normal[1], nz = f.normal[2], nn = nx * nx + ny * ny + nz * nz;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ddx = ps[k].x - cx, ddy = ps[k].y - cy, ddz = ps[k].z - cz;
        r2 = fmax(r2, ddx * ddx + ddy * ddy + ddz * ddz);
        big = fmax(big, fmax(fabs((double)ps[k].x), fmax(fabs((double)ps[k].y), fabs((double)ps[k].z))));
        if (k > 0 && nn > 0.0) {                                    // the vertex as the edge functions see it
            const double s = (((double)ps[k].x - p1.x) * nx + ((double)ps[k].y - p1.y) * ny + ((double)ps[k].z - p1.z) * nz) / nn;
            const double qx = ps[k].x - s * nx - cx, qy = ps[k].y - s * ny - cy, qz = ps[k].z - s * nz - cz;
            r2 = fmax(r2, qx * qx + qy * qy + qz * qz);
        }
    }
    const double e1x = (double)p2.x - p1.x, e1y = (double)p2.y - p1.y, e1z = (double)p2.z - p1.z;
    const double e2x = (double)p3.x - p1.x, e2y = (double)p3.y - p1.y, e2z = (double)p3.z - p1.z;
    const bool degenerate = (e1y * e2z - e2y * e1z == 0.0) && (e1z * e2x - e2z * e1x == 0.0) && (e1x * e2y - e2x * e1y == 0.0);   // incl. coincident vertices
    const double r = sqrt(r2) * 1.001 + 1e-5 * (1.0 + big);
    float r2f = (float)(r * r);
    if ((double)r2f < r * r) r2f = __uint_as_float(__float_as_uint(r2f) + 1u);
    const bool always = degenerate || !(r2f < __builtin_inff());    // (NaN compares false)
    if (always) r2f = 3e38f;                                        // finite: the VALU scan's h^2 - c + 1e-5 c must stay a number
    if (all) bound[i] = make_float4((float)cx, (float)cy, (float)cz, r2f);
    {
        // the filter works in coordinates about the centre of the vertices' box (the VALU scan's `bound` stays in world coordinates)
        const float fx = (float)(cx - (double)centre[0]), fy = (float)(cy - (double)centre[1]), fz = (float)(cz - (double)centre[2]);
        const double c2 = (double)fx * fx + (double)fy * fy + (double)fz * fz;
        #if RT3_FACE_K32
        write_frag(fx, fy, fz, always ? kAlwaysCandidate : filter_kj32(c2, (double)r2f));
#else
        write_frag(fx, fy, fz, always ? kAlwaysCandidate : filter_kj(c2, (double)r2f));
#endif
    }
    if (!all) return;
    if (mats) {
        const rt3_material m = mats[i];
        if (m.kind == RT3_MAT_DIELECTRIC) {                         // same packing as pack_material() on the host
            const float ri_f = 1.0f / m.param, ri_b = m.param;
            float r0f = (1.0f - ri_f) / (1.0f + ri_f), r0b = (1.0f - ri_b) / (1.0f + ri_b);
            mat[i] = make_float4(ri_f, r0f * r0f, r0b * r0b, m.param);
        } else mat[i] = make_float4(m.rgb[0], m.rgb[1], m.rgb[2], m.param);
        kind[i] = m.kind;
    }
    else { mat[i] = make_float4(f.color[0], f.color[1], f.color[2], 0.0f); kind[i] = RT3_MAT_FLAT; }
}


// ------------------------------------------------------------------------------------------------------
// Group rows of the two-level candidate filter
// ------------------------------------------------------------------------------------------------------
// `bounds` holds one sphere (cx, cy, cz, r^2) per primitive — the analytic sphere itself, or the bounding sphere of a face (k_commit_mesh) —
// in GROUP ORDER: group g = entries [g G, (g + 1) G).  Row g of the matrix filter becomes ONE sphere that encloses the G members, so the
// scan costs 1 / G of the flat filter's matrix and decode work; a ray that is a candidate for the row then runs the members' own tests.
// What the row must guarantee is what every filter row guarantees (5.2c): whenever the exact test of ANY member accepts a ray, the row's
// accumulator has its sign bit clear.
//   C = mean of the members' centres, R = max (|c_m - C| + r_m): every line that comes within r_m of c_m comes within R of C, and for such
//   a line the group's discriminant R^2 - dist(C)^2 is at least the member's r_m^2 - dist(c_m)^2 (0 <= dist(c_m) <= r_m).
//   The exact SPHERE test evaluates its discriminant in f32 and accepts down to a true value of -eta, eta <= 12 u |c_m - o|^2 (u = 2^-24):
//   the line may miss the member by delta <= min(sqrt(eta), eta / 2 r_m), which costs the group's discriminant up to 2 (R + delta) delta
//     <= 1.7e-3 R |o'| + 1.7e-3 R rho + 2.9e-6 (|o'|^2 + rho^2),      rho = |C'| + R  (primes: about the filter's centre)
//     <= [0.034 R^2 + 1.7e-3 R rho + 2.9e-6 rho^2]  +  2.4e-5 |o'|^2    (R |o'| <= (40 R^2 + |o'|^2 / 40) / 2).
//   The ray-side part is a third of what the K = 32 margin leaves free (0.35 eps32 |o'|^2 = 7.7e-5 |o'|^2, 5.2c); the bracket is added to
//   the row's radius here, rounded up: R_eff^2 = 1.04 R^2 + 2e-3 R rho + 4e-6 rho^2.  (The exact FACE test never accepts a line that misses
//   the face's bound — its inflation is 80x the rounding of a hit point — so for faces the same inflation is pure slack.)
// A member with r^2 >= 3e38 (a face without a bounded hit region) makes its row an always-candidate; members with a negative or non-finite
// record (padding; spheres no exact test can ever accept) do not count; a row without members can never be a candidate.
// The bounding sphere of group g: (C', R_eff^2) with C' in WORLD coordinates, or r^2 = 3e38 (always a candidate) / -1e30 (never: no usable member).
__device__ __forceinline__ float4 group_bound(const float4* __restrict__ bounds, uint32_t n_entries, uint32_t group, uint32_t g, const float centre[3]) {
    double sx = 0.0, sy = 0.0, sz = 0.0;
    uint32_t members = 0;
    bool always = false;
    auto usable = [](const float4& b) { return b.w >= 0.0f && b.w < 3e38f && b.x - b.x == 0.0f && b.y - b.y == 0.0f && b.z - b.z == 0.0f; };
    for (uint32_t m = 0; m < group; m++) {
        const uint64_t j = (uint64_t)g * group + m;
        if (j >= n_entries) break;
        const float4 b = bounds[j];
        if (b.w >= 3e38f) { always = true; continue; }              // (whatever its centre is)
        if (!usable(b)) continue;
        sx += b.x; sy += b.y; sz += b.z; members++;
    }
    if (always) return make_float4(0.0f, 0.0f, 0.0f, 3e38f);
    if (members == 0) return kPadSphere;
    // centre: towards the smallest enclosing sphere of the members' spheres (Badoiu-Clarkson: start at the mean, step 1 / (k + 1) towards the
    // farthest member, 32 times) — R^2 comes out a quarter smaller than about the mean on the 100 000-sphere scene, and with it the candidates;
    // any centre is correct, R below is measured from whichever one this ends on
    double cx = sx / members, cy = sy / members, cz = sz / members;
    for (int it = 1; it <= 32 && members > 1; it++) {
        double far = -1.0, tx = cx, ty = cy, tz = cz;
        for (uint32_t m = 0; m < group; m++) {
            const uint64_t j = (uint64_t)g * group + m;
            if (j >= n_entries) break;
            const float4 b = bounds[j];
            if (!usable(b)) continue;
            const double ddx = b.x - cx, ddy = b.y - cy, ddz = b.z - cz, d = sqrt(ddx * ddx + ddy * ddy + ddz * ddz) + sqrt((double)b.w);
            if (d > far) { far = d; tx = b.x; ty = b.y; tz = b.z; }
        }
        const double step = 1.0 / (it + 1);
        cx += (tx - cx) * step; cy += (ty - cy) * step; cz += (tz - cz) * step;
    }
    const float fcx = (float)cx, fcy = (float)cy, fcz = (float)cz;  // the centre as it is stored: R is measured from THIS point
    double R = 0.0;
    for (uint32_t m = 0; m < group; m++) {
        const uint64_t j = (uint64_t)g * group + m;
        if (j >= n_entries) break;
        const float4 b = bounds[j];
        if (!usable(b)) continue;
        const double ddx = (double)b.x - fcx, ddy = (double)b.y - fcy, ddz = (double)b.z - fcz;
        R = fmax(R, sqrt(ddx * ddx + ddy * ddy + ddz * ddz) + sqrt((double)b.w));
    }
    const double qx = (double)fcx - centre[0], qy = (double)fcy - centre[1], qz = (double)fcz - centre[2];
    const double rho = sqrt(qx * qx + qy * qy + qz * qz) + R;
    const double r2 = (1.04 * R * R + 2e-3 * R * rho + 4e-6 * rho * rho) * (1.0 + 1e-6) + 1e-30;
    if (!(r2 < 1e30)) return make_float4(0.0f, 0.0f, 0.0f, 3e38f);
    float r2f = (float)r2;
    if ((double)r2f < r2) r2f = __uint_as_float(__float_as_uint(r2f) + 1u);
    return make_float4(fcx, fcy, fcz, r2f);
}
// The faces' exact-test records (4 x float4) in group order: rec[4 k ..] = tri[4 perm[k] ..]; positions without a face (padding) get zeros — their
// bound is a never-candidate, nothing reads them.
__global__ void k_gather_face_records(const float4* __restrict__ tri, const uint32_t* __restrict__ perm, uint32_t n_pos, float4* __restrict__ rec) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos * 4u) return;
    const uint32_t j = perm[i >> 2];
    rec[i] = j == 0xFFFFFFFFu ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : tri[(size_t)j * 4u + (i & 3u)];
}
// The groups' bounds as records (three-level filter: the leaves' spheres, which the middle level tests in f32).
__global__ void k_group_bounds(const float4* __restrict__ bounds, uint32_t n_entries, uint32_t group, uint32_t n_groups, const uint32_t* __restrict__ box,
                               float ecx, float ecy, float ecz, float4* __restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    float centre[3] = { ecx, ecy, ecz };
    if (box) box_centre(box, centre);
    out[g] = group_bound(bounds, n_entries, group, g, centre);
}
// The groups' bounds as rows of the matrix filter (K = 32 fragments about the filter's centre).
__global__ void k_group_frags(const float4* __restrict__ bounds, uint32_t n_entries, uint32_t group, uint32_t n_rows_padded,
                              const uint32_t* __restrict__ box, float ecx, float ecy, float ecz, u32x4* __restrict__ frag) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_rows_padded) return;
    float centre[3] = { ecx, ecy, ecz };
    if (box) box_centre(box, centre);
    const float4 gb = group_bound(bounds, n_entries, group, g, centre);
    float fx = 0.0f, fy = 0.0f, fz = 0.0f, kj = kNeverCandidate;
    if (gb.w >= 3e38f) kj = kAlwaysCandidate;
    else if (gb.w >= 0.0f) {
        fx = (float)((double)gb.x - (double)centre[0]); fy = (float)((double)gb.y - (double)centre[1]); fz = (float)((double)gb.z - (double)centre[2]);
        const double c2 = (double)fx * fx + (double)fy * fy + (double)fz * fz;
        kj = filter_kj32(c2, (double)gb.w);
    }
    uint32_t fr[4][4];
    bound_frag32_row(fx, fy, fz, kj, fr);
    for (uint32_t q = 0; q < 4; q++) frag[frag32_index(g / 32, g % 32, q)] = u32x4{ fr[q][0], fr[q][1], fr[q][2], fr[q][3] };
}

}  // namespace
