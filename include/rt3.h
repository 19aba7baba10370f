/* rt3.h — C ABI of the MI355X-native render path (librt3hip.so).
 *
 * Drop-in boundary for the reference's per-pixel ray-trace loop.  The reference's own boundary is the
 * C++ class RayTracer::Renderer (src/lib/renderer/Renderer.hpp:34-63: prerender() / render() /
 * initialize_renderer()); the C++ mirror of that class lives in raytracer-3_amd/host/ and is a thin
 * wrapper over the entry points declared here.  Everything below is plain pointers and sizes so that any
 * FFI (ctypes, cgo, JNI ...) can bind it; INTEGRATION.md shows the binding a reference maintainer adds.
 *
 * Two render modes:
 *   Mode R  ("reference mode")  rt3_render*      — exactly SequentialRenderer::render + ray_color
 *            (src/lib/renderer/SequentialRenderer.cpp:47-109, 269-308): 1 primary ray per pixel,
 *            brute-force nearest indexed triangle, flat baked face colour, sky gradient, RGBA8 pack.
 *   Mode X  ("extension mode")  rt3_render_path* — analytic spheres, materials, spp, depth, counter-based
 *            RNG, as sketched (never finished) by src/lib/shaders/raytracer/raytracer_v4.glsl and
 *            random_v1.glsl; semantics are specified in DESIGN.md and restated on the CPU in oracle/.
 *
 * All functions returning int return 0 on success and a negative code on failure; the message is
 * available from rt3_last_error().  Nothing here falls back to the CPU: without a HIP device every
 * device entry point fails with RT3_E_DEVICE.
 */
#ifndef RT3_H
#define RT3_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: bumped whenever a wire struct changes size or meaning.  A host compares it with what the library
 * it loaded reports before it passes any struct (rt3_stats grew from 56 to 64 bytes between versions 1 and 2; rt3_get_stats writes
 * sizeof(rt3_stats) bytes of THIS version).  History: 1 = round 1; 2 = + mfma_instructions / exact_tests in rt3_stats, progressive
 * accumulation, rt3_gather_rows; 3 = + rt3_abi_version itself, one stream convention (below),
 * filter_tests / bound_tests in rt3_stats (80 bytes). */
#define RT3_ABI_VERSION 3u
uint32_t rt3_abi_version(void);

/* Stream convention of every entry point that takes a `stream` (a hipStream_t): the work is queued on that stream; NULL means the
 * CONTEXT'S OWN stream (rt3_stream()), never the legacy default stream.  Calls on one context are ordered with each other whatever
 * streams they name: a render that continues, overwrites or reads the accumulation waits (on the device) for the event the previous
 * render recorded, rt3_accum_download / rt3_accum_upload wait for it on the host. */

#define RT3_E_ARG     (-1)   /* bad argument */
#define RT3_E_DEVICE  (-2)   /* HIP runtime / device failure (includes "no device") */
#define RT3_E_IO      (-3)   /* file could not be opened / parsed */
#define RT3_E_STATE   (-4)   /* no scene set, etc. */

/* ---------------------------------------------------------------------------------------------------
 * Wire structs
 * ------------------------------------------------------------------------------------------------- */

/* == RayTracer::GFace (src/lib/renderer/Vertex.hpp:39-51; GLSL std430 twin raytracer_v3.glsl:33-45).
 * 48 bytes: u32 v1,v2,v3 @0/4/8; vec3 normal @16; vec3 color @32. */
typedef struct rt3_gface {
    uint32_t v1, v2, v3;
    uint32_t _pad0;
    float    normal[3];
    uint32_t _pad1;
    float    color[3];
    uint32_t _pad2;
} rt3_gface;

/* == the four public vectors of RayTracer::Camera (src/lib/camera/Camera.hpp:27-34), i.e. the
 * GCameraData block the Vulkan backend uploads (src/lib/renderer/VulkanRenderer.hpp:32-37). */
typedef struct rt3_camera {
    float origin[3];
    float horizontal[3];
    float vertical[3];
    float lower_left_corner[3];
} rt3_camera;

/* Material kinds (Mode X).  RT3_MAT_FLAT is the only material the reference has: the hit returns the
 * baked colour and the path ends (SequentialRenderer.cpp:101-103); it doubles as a diffuse emitter. */
#define RT3_MAT_FLAT        0u   /* radiance += throughput * rgb ; path ends                     */
#define RT3_MAT_LAMBERT     1u   /* rgb = albedo                                                  */
#define RT3_MAT_METAL       2u   /* rgb = albedo, param = fuzz in [0,1]                           */
#define RT3_MAT_DIELECTRIC  3u   /* param = index of refraction (rgb ignored, attenuation = 1)    */

typedef struct rt3_material {
    float    rgb[3];
    float    param;
    uint32_t kind;
} rt3_material;

#define RT3_FLAG_GAMMA2            1u   /* sqrt() each channel before packing (book gamma 2)           */
#define RT3_FLAG_BLACK_BACKGROUND  2u   /* a miss contributes nothing (default: the reference's sky)    */
/* The primary ray (ray cast 0) is traced exactly as SequentialRenderer.cpp:289-297 does it: the direction is left
 * UNNORMALISED (:293), faces are tested with the reference's literal formula t = (n.o + n.p1) / (n.d) (:70, sic) and
 * a miss shades sky(d) of that unnormalised d (:105-107).  With spp 1, max_depth 1, flat faces, t_min 0 and no
 * gamma, Mode X then IS Mode R, byte for byte (consequence 1(i)); later ray casts are the
 * normal Mode-X ones.  Triangle-only scenes. */
#define RT3_FLAG_REFERENCE_PRIMARY 4u
/* Keep a per-pixel, per-channel sum of squared sample radiances beside the sums (variance estimates; the
 * "sample storage" half of reduce_v1.glsl's intent).  Read both back with rt3_accum_download(). */
#define RT3_FLAG_VARIANCE          8u

/* Parameters of a Mode-X render.  tile_*: interleaved row-block sharding of the framebuffer
 * (design intent: BlockInfo{x,y,w,h} of raytracer_v4.glsl:70-79).  Row-block b (tile_rows rows) belongs
 * to shard (b mod tile_count); a shard renders only its own rows, into a compact buffer of
 * rt3_rows_owned() rows.  tile_count = 1 renders the whole frame. */
typedef struct rt3_params {
    uint32_t width, height;      /* full frame */
    uint32_t spp;                /* samples per pixel, >= 1 */
    uint32_t max_depth;          /* ray casts per path, >= 1 (book "max_depth") */
    uint32_t seed;
    uint32_t flags;
    float    lens_radius;        /* 0 = pinhole */
    float    t_min;              /* self-intersection cut-off; book value 0.001 */
    uint32_t tile_rows, tile_index, tile_count;
} rt3_params;

/* Counters of the last render on a context (device-side counts, HIP-event timings on the render stream). */
typedef struct rt3_stats {
    uint64_t ray_casts;          /* rays traced (primary + scattered)                                */
    uint64_t prim_tests;         /* ray-primitive tests = ray_casts * (n_spheres + n_faces)          */
    uint64_t samples;            /* pixels * spp rendered by this call                               */
    float    trace_ms;           /* dominant kernel (trace / mode-R) duration, summed over launches  */
    float    total_ms;           /* first launch -> last launch of the call (device time)            */
    uint32_t launches;           /* launches of the dominant kernel                                  */
    uint32_t n_spheres, n_faces;
    uint32_t mfma_flop_per_instruction;   /* 32768 (v_mfma_f32_32x32x16_bf16: k_trace_mfma) or 16384 (v_mfma_f32_16x16x32_bf16: tiled kernels) */
    uint64_t mfma_instructions;  /* bf16 MFMA wave-instructions issued by the candidate filter (0: VALU scan / brute force) */
    uint64_t exact_tests;        /* (ray, primitive) pairs that survived the filter(s) and went through the exact test
                                    (counted by the pair-list kernels; 0 elsewhere)                                      */
    uint64_t filter_tests;       /* (ray, row) pairs the matrix filter evaluated = ray_casts * rows; a row is one primitive in the
                                    flat filter and a group of primitives in the two-level filter, so this is what
                                    the matrix cores executed, while prim_tests is the brute-force-equivalent count              */
    uint64_t bound_tests;        /* two-level filter, faces: members of candidate groups checked against their own bounding sphere */
} rt3_stats;

typedef struct rt3_ctx rt3_ctx;

/* ---------------------------------------------------------------------------------------------------
 * Device context   (replaces the Vulkan Instance/GPU/MemoryPool bring-up of VulkanRenderer.cpp:43-94)
 * ------------------------------------------------------------------------------------------------- */
rt3_ctx*    rt3_create(int device_id);
void        rt3_destroy(rt3_ctx* ctx);
/* ctx may be NULL to read the error of a failed rt3_create(). */
const char* rt3_last_error(const rt3_ctx* ctx);
/* Optional: cap (bytes) for the per-sample radiance storage; more spp than fit are rendered in batches. */
int         rt3_set_sample_storage_cap(rt3_ctx* ctx, uint64_t bytes);

/* ---------------------------------------------------------------------------------------------------
 * Scene upload   (replaces Renderer::prerender's upload half: SequentialRenderer.cpp:174-195,246 /
 *                 VulkanRenderer.cpp:210-261,355-387).  Replaces any previous scene of that kind.
 * ------------------------------------------------------------------------------------------------- */
/* faces/vertices are the merged arrays exactly as the reference keeps them (GFace[], vec4[] w=0).
 * face_materials may be NULL: every face is then RT3_MAT_FLAT with its own GFace colour (Mode R). */
int rt3_set_mesh(rt3_ctx* ctx, const rt3_gface* faces, uint32_t n_faces,
                 const float* vertices_xyzw, uint32_t n_vertices,
                 const rt3_material* face_materials);
/* Incremental form of the same upload, mirroring VulkanRenderer::prerender (VulkanRenderer.cpp:266-399): size the two
 * device buffers from the entities' pre-declared counts, then fill them entity by entity at running offsets —
 *   rt3_mesh_put     a CPU-pre-rendered entity; indices are rebased by vertex_offset (transfer_entity, :210-261)
 *   rt3_mesh_sphere  a sphere tessellated ON THE DEVICE straight into the buffers (gpu_pre_render_sphere,
 *                    src/lib/entities/Sphere.cpp:355-491; shaders pre_render_sphere_v2_vertices/faces.glsl)
 * and finally rt3_mesh_commit() de-indexes the merged arrays for the render kernels (face_materials as in rt3_set_mesh).
 * rt3_mesh_download() copies the merged GFace[] / vec4[] back (the check the author left commented out at
 * VulkanRenderer.cpp:329-353). */
int rt3_mesh_begin(rt3_ctx* ctx, uint32_t n_faces, uint32_t n_vertices);
int rt3_mesh_put(rt3_ctx* ctx, const rt3_gface* faces, uint32_t n_faces, const float* vertices_xyzw, uint32_t n_vertices,
                 uint32_t face_offset, uint32_t vertex_offset);
int rt3_mesh_sphere(rt3_ctx* ctx, const float center[3], float radius, uint32_t n_meridians, uint32_t n_parallels,
                    const float color[3], uint32_t face_offset, uint32_t vertex_offset);
int rt3_mesh_commit(rt3_ctx* ctx, const rt3_material* face_materials);
int rt3_mesh_download(rt3_ctx* ctx, rt3_gface* faces, float* vertices_xyzw);

/* center_radius: 4 floats per sphere (cx,cy,cz,r), r > 0.  (Sphere{vec3 center; float radius; vec3 color},
 * raytracer_v4.glsl:42-49.) */
int rt3_set_spheres(rt3_ctx* ctx, const float* center_radius, const rt3_material* materials, uint32_t n);

/* ---------------------------------------------------------------------------------------------------
 * Render   (replaces Renderer::render, Renderer.hpp:50)
 * ------------------------------------------------------------------------------------------------- */
/* Mode R, synchronous, host output: out_pixels[w*h] in the reference's word layout
 * (0xFF | B<<8 | G<<16 | R<<24, row 0 = top; SequentialRenderer.cpp:297).  All rows are written; row h-1
 * follows the GLSL twin (raytracer_v3.glsl:193-196) because the CPU loop never writes it. */
int rt3_render(rt3_ctx* ctx, const rt3_camera* cam, uint32_t width, uint32_t height, uint32_t* out_pixels);
/* Mode R, asynchronous on `stream` (a hipStream_t; NULL = the context's own stream), device output buffer of w*h words. */
int rt3_render_device(rt3_ctx* ctx, const rt3_camera* cam, uint32_t width, uint32_t height,
                      void* d_out_pixels, void* stream);

/* Mode X, synchronous, host output of rt3_rows_owned(params)*width words (compact tile rows). */
int rt3_render_path(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* params, uint32_t* out_pixels);
/* Mode X, asynchronous on `stream` (NULL = the context's own stream), device output. */
int rt3_render_path_device(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* params,
                           void* d_out_pixels, void* stream);

/* Progressive / resumable form (design intent reduce_v1.glsl:28-76 + the SampleStorage of
 * raytracer_v4.glsl:107-111): renders samples [sample_begin, sample_begin + sample_count) of the params->spp samples
 * per pixel and adds them, in sample order, to the per-pixel accumulators the context keeps between calls; the output
 * is the frame resolved over the samples accumulated so far (sum / (sample_begin + sample_count)).  sample_begin == 0
 * starts a new accumulation; otherwise it must equal the number of samples already accumulated for the SAME camera and
 * params (RT3_E_STATE if not).  Any partition of [0, spp) into consecutive calls gives the frame of one
 * rt3_render_path*() call, bit for bit.  rt3_render_path_device(p) == rt3_render_path_range_device(p, 0, p->spp). */
int rt3_render_path_range(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* params,
                          uint32_t sample_begin, uint32_t sample_count, uint32_t* out_pixels);
int rt3_render_path_range_device(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* params,
                                 uint32_t sample_begin, uint32_t sample_count, void* d_out_pixels, void* stream);
/* Checkpoint / resume of that accumulation.  Download: sum (and, when the accumulation runs with
 * RT3_FLAG_VARIANCE and sum_sq != NULL, the sums of squares) as 4 floats per owned pixel (r, g, b, 0), compact tile
 * rows as rt3_render_path writes them; *samples_done = samples accumulated.  Upload: restores such a state for
 * (cam, params) — possibly into another context or process — so that the next rt3_render_path_range() call continues
 * at sample_begin == samples_done.  sum_sq may be NULL when params->flags lacks RT3_FLAG_VARIANCE. */
int rt3_accum_download(rt3_ctx* ctx, float* sum_rgba, float* sum_sq_rgba, uint32_t* samples_done);
int rt3_accum_upload(rt3_ctx* ctx, const rt3_camera* cam, const rt3_params* params,
                     const float* sum_rgba, const float* sum_sq_rgba, uint32_t samples_done);

/* Multi-GPU gather of final pixels without the host ("hipMemcpyPeerAsync into rank 0's buffer at
 * the right offsets, no de-interleave needed"; tiling intent BlockInfo, raytracer_v4.glsl:70-79).  d_tile holds the
 * compact rows of the shard described by shard_params (as rt3_render_path_device wrote them) on the device of
 * `shard`; they are copied to their interleaved positions inside the full frame d_frame (width*height words) on the
 * device of `root` — one strided 2-D device-to-device copy (peer-to-peer over xGMI when the two contexts sit on
 * different GPUs; peer access is enabled on first use) plus one 1-D copy when the last row block is ragged.  The
 * copies are issued on `stream`, which must be a stream of the SHARD's device (NULL = the shard context's own stream),
 * so that they run behind the shard's resolve kernel.  root == shard is allowed (single GPU). */
int rt3_gather_rows(rt3_ctx* root, void* d_frame, rt3_ctx* shard, const void* d_tile,
                    const rt3_params* shard_params, void* stream);
/* What rt3_gather_rows copies, as plain arithmetic (no device needed: a host can check or re-use it, tests/test_gather_plan.py): at most two
 * strided copies of `rows` pieces of `row_bytes` bytes each, piece i from tile byte src_offset + i * src_pitch to frame byte dst_offset +
 * i * dst_pitch.  Returns the number of copies written to out[0..1] (0: the shard owns no row) or RT3_E_ARG. */
typedef struct rt3_gather_copy {
    uint64_t dst_offset, src_offset;   /* bytes into the frame / into the compact tile */
    uint64_t dst_pitch, src_pitch;     /* bytes between consecutive pieces */
    uint64_t row_bytes;                /* bytes per piece (a whole row block of the shard, or the ragged last one) */
    uint32_t rows;                     /* pieces */
} rt3_gather_copy;
int rt3_gather_plan(const rt3_params* shard_params, rt3_gather_copy out[2]);
/* The context's own stream (a hipStream_t) and a wait for it — what a multi-device host needs around rt3_gather_rows. */
void* rt3_stream(rt3_ctx* ctx);
int   rt3_synchronize(rt3_ctx* ctx);
/* Device frame buffer helpers for hosts that have no other HIP binding: allocate / free n_words uint32 on the
 * context's device, and copy such a buffer to the host (synchronous, after everything queued on the ctx stream). */
void* rt3_device_alloc_words(rt3_ctx* ctx, uint64_t n_words);
void  rt3_device_free(rt3_ctx* ctx, void* d_ptr);
int   rt3_device_read_words(rt3_ctx* ctx, const void* d_ptr, uint64_t n_words, uint32_t* out);

/* Rows of the frame owned by shard tile_index (see rt3_params), and the frame row of local row i. */
uint32_t rt3_rows_owned(const rt3_params* params);
uint32_t rt3_row_of_local(const rt3_params* params, uint32_t local_row);

/* Waits for the last asynchronous render and fills `out` (may be called after the synchronous forms too). */
int rt3_get_stats(rt3_ctx* ctx, rt3_stats* out);

/* ---------------------------------------------------------------------------------------------------
 * Host-side scene API   (the step before the path: entities -> GFace[]/vec4[]; plain CPU code)
 * ------------------------------------------------------------------------------------------------- */
/* cpu_pre_render_triangle (src/lib/entities/Triangle.cpp:28-76): 1 face, 3 vertices (xyzw). */
void     rt3_prerender_triangle(const float p1[3], const float p2[3], const float p3[3], const float color[3],
                                rt3_gface* faces, float* vertices_xyzw);
/* create_sphere counts (src/lib/entities/Sphere.cpp:101-102). */
uint32_t rt3_sphere_face_count(uint32_t n_meridians, uint32_t n_parallels);
uint32_t rt3_sphere_vertex_count(uint32_t n_meridians, uint32_t n_parallels);
/* cpu_pre_render_sphere (src/lib/entities/Sphere.cpp:120-261). */
void     rt3_prerender_sphere(const float center[3], float radius, uint32_t n_meridians, uint32_t n_parallels,
                              const float color[3], rt3_gface* faces, float* vertices_xyzw);
/* create_object's counting pass (src/lib/entities/Object.cpp:84-119). */
int      rt3_object_count(const char* path, uint32_t* n_faces, uint32_t* n_vertices);
/* cpu_pre_render_object (src/lib/entities/Object.cpp:131-199). */
int      rt3_prerender_object(const char* path, const float center[3], float scale, const float color[3],
                              rt3_gface* faces, uint32_t n_faces, float* vertices_xyzw, uint32_t n_vertices);
/* SequentialRenderer::transfer_entity (SequentialRenderer.cpp:174-195): append with index rebasing.
 * dst_* must have room; *dst_nf / *dst_nv are advanced. */
void     rt3_transfer_entity(rt3_gface* dst_faces, uint32_t* dst_nf, float* dst_vertices_xyzw, uint32_t* dst_nv,
                             const rt3_gface* faces, uint32_t nf, const float* vertices_xyzw, uint32_t nv);

/* Camera::update (src/lib/camera/Camera.cpp:77-96). */
void     rt3_camera_update(rt3_camera* cam, float focal_length, float viewport_width, float viewport_height);
/* Extension (the reference camera cannot look-from/look-at): book camera, vfov in degrees. */
void     rt3_camera_look_at(rt3_camera* cam, const float from[3], const float at[3], const float vup[3],
                            float vfov_deg, float aspect, float focus_dist);

/* Frame::to_ppm bytes (src/lib/camera/Frame.cpp:109-148): header + 3*w*h bytes.  Returns bytes written to
 * `out` (capacity cap) or the required size when out == NULL. */
uint64_t rt3_frame_ppm_bytes(const uint32_t* pixels, uint32_t width, uint32_t height, uint8_t* out, uint64_t cap);
int      rt3_frame_to_ppm(const uint32_t* pixels, uint32_t width, uint32_t height, const char* path);

/* Benchmark scenes (build-owned).  Each returns the sphere count of the scene — the count REQUIRED, whatever
 * cap is (snprintf convention) — and writes at most cap spheres; NULL outputs write nothing.  All randomness comes from the reference's hash RNG
 * (src/lib/shaders/random_v1.glsl:22-52) keyed by (seed, slot, dimension). */
uint32_t rt3_scene_three_spheres(float* center_radius, rt3_material* materials, uint32_t cap);
uint32_t rt3_scene_weekend(uint32_t seed, float* center_radius, rt3_material* materials, uint32_t cap);
uint32_t rt3_scene_stress(uint32_t n, uint32_t seed, float* center_radius, rt3_material* materials, uint32_t cap);
/* Cornell-style box tessellated into triangles (grid x grid quads per wall) with one emissive quad.
 * Returns the scene's face count (required, as above; at most cap_faces are written); vertices = 3 per face (unindexed).
 * NULL outputs -> counts only. */
uint32_t rt3_scene_cornell(uint32_t grid, rt3_gface* faces, float* vertices_xyzw, rt3_material* face_materials,
                           uint32_t cap_faces);

/* Reference hash RNG, exported for known-answer tests (random_v1.glsl:22-52). */
uint32_t rt3_hash_u32(uint32_t x);
float    rt3_random_float(uint32_t m);

/* Debug probe used by the parity tests only: element-wise DEVICE arithmetic on n inputs —
 * div = a/b, sq = sqrt(|a|), fm = fma(a,b,a), (cs,sn) = sincos2pi(frac bits of a), sk3 = sky(a,b,-2) (3 per
 * element), pk = pack(a,b,u).  Lets the tests prove the device's IEEE behaviour matches the host's. */
/* Debug switch used by the parity tests only: non-zero makes rt3_render* always take the plain brute-force Mode-R kernel
 * instead of the bounding-sphere-filtered one (both must give identical pixels). */
int      rt3_debug_force_plain_mode_r(rt3_ctx* ctx, int on);
/* Debug switch used by the parity tests and the fuzzers only: non-zero makes rt3_render_path* take k_trace_brute, the
 * UNFILTERED Mode-X kernel (every ray against every primitive in index order, no bounding spheres, no matrix cores) —
 * the on-GPU arbiter for the candidate filters.  Same effect: environment variable RT3_BRUTE=1. */
int      rt3_debug_force_brute(rt3_ctx* ctx, int on);
/* Debug switch (tests, A/B): non-zero makes the tiled matrix-filter kernel scan one row per primitive (the flat filter) instead of the
 * two-level filter's group rows.  Same pixels.  Same effect: environment variable RT3_NO_GROUPS=1. */
int      rt3_debug_force_flat_filter(rt3_ctx* ctx, int on);
int      rt3_debug_arith(rt3_ctx* ctx, const float* a, const float* b, uint32_t n, float* div, float* sq, float* fm,
                         float* cs, float* sn, float* sk3, uint32_t* pk);

#ifdef __cplusplus
}
#endif
#endif /* RT3_H */
