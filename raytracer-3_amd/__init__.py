"""rt3 — MI355X-native render path of Lut99/RayTracer-3, Python host mirror.

The package directory is ``raytracer-3_amd``; import it with
``importlib.import_module("raytracer-3_amd")`` (the hyphen rules out a plain ``import``).

This module is a ctypes binding over ``librt3hip.so`` (C ABI: ``include/rt3.h``) and mirrors the reference's
host interface for the render path, same names and argument meaning:

=====================================  ==========================================================
reference (C++)                        here
=====================================  ==========================================================
``RayTracer::initialize_renderer()``   :func:`initialize_renderer`  (Renderer.hpp:63)
``Renderer::prerender(entities)``      :meth:`HipRenderer.prerender` (Renderer.hpp:48)
``Renderer::render(camera)``           :meth:`HipRenderer.render`    (Renderer.hpp:50)
``ECS::create_triangle/sphere/object`` :func:`create_triangle` ...   (entities/*.hpp)
``Camera::update`` / ``get_frame``     :class:`Camera`               (camera/Camera.hpp)
``Frame::d/w/h/to_ppm``                :class:`Frame`                (camera/Frame.hpp)
=====================================  ==========================================================

There is no CPU fallback: if the HIP library is missing or no GPU is present, construction of the
renderer raises :class:`Fatal` (the reference's ``CppDebugger::Fatal`` convention, Main.cpp:305).
Nothing here imports ``oracle/``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT3_LIB_PATH", os.path.join(_HERE, "librt3hip.so"))      # override: A/B two builds

# --------------------------------------------------------------------------------------------------------
# wire structs (include/rt3.h)
# --------------------------------------------------------------------------------------------------------
GFACE = np.dtype([("v1", "<u4"), ("v2", "<u4"), ("v3", "<u4"), ("_p0", "<u4"),
                  ("normal", "<f4", 3), ("_p1", "<u4"), ("color", "<f4", 3), ("_p2", "<u4")])
MATERIAL = np.dtype([("rgb", "<f4", 3), ("param", "<f4"), ("kind", "<u4")])

MAT_FLAT, MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2, 3
FLAG_GAMMA2, FLAG_BLACK_BACKGROUND, FLAG_REFERENCE_PRIMARY, FLAG_VARIANCE = 1, 2, 4, 8


class rt3_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("lower_left_corner", C.c_float * 3)]


class rt3_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("seed", C.c_uint32), ("flags", C.c_uint32), ("lens_radius", C.c_float), ("t_min", C.c_float),
                ("tile_rows", C.c_uint32), ("tile_index", C.c_uint32), ("tile_count", C.c_uint32)]


class rt3_stats(C.Structure):
    _fields_ = [("ray_casts", C.c_uint64), ("prim_tests", C.c_uint64), ("samples", C.c_uint64),
                ("trace_ms", C.c_float), ("total_ms", C.c_float), ("launches", C.c_uint32),
                ("n_spheres", C.c_uint32), ("n_faces", C.c_uint32), ("mfma_flop_per_instruction", C.c_uint32), ("mfma_instructions", C.c_uint64),
                ("exact_tests", C.c_uint64), ("filter_tests", C.c_uint64), ("bound_tests", C.c_uint64)]


class rt3_gather_copy(C.Structure):
    _fields_ = [("dst_offset", C.c_uint64), ("src_offset", C.c_uint64), ("dst_pitch", C.c_uint64), ("src_pitch", C.c_uint64),
                ("row_bytes", C.c_uint64), ("rows", C.c_uint32)]


def gather_plan(params):
    """The copies rt3_gather_rows issues for a shard, as a list of rt3_gather_copy (pure arithmetic, no device)."""
    out = (rt3_gather_copy * 2)()
    n = lib().rt3_gather_plan(C.byref(params), out)
    if n < 0:
        raise Fatal("rt3_gather_plan: bad shard parameters")
    return [out[i] for i in range(n)]


class Fatal(RuntimeError):
    """Mirror of CppDebugger::Fatal: every backend error is fatal (Main.cpp:305-308)."""


# every symbol include/rt3.h declares; tests check the library exports all of them
EXPORTS = [
    "rt3_create", "rt3_destroy", "rt3_last_error", "rt3_set_sample_storage_cap", "rt3_set_mesh", "rt3_set_spheres",
    "rt3_render", "rt3_render_device", "rt3_render_path", "rt3_render_path_device", "rt3_rows_owned",
    "rt3_row_of_local", "rt3_get_stats", "rt3_prerender_triangle", "rt3_sphere_face_count",
    "rt3_sphere_vertex_count", "rt3_prerender_sphere", "rt3_object_count", "rt3_prerender_object",
    "rt3_transfer_entity", "rt3_camera_update", "rt3_camera_look_at", "rt3_frame_ppm_bytes", "rt3_frame_to_ppm",
    "rt3_scene_three_spheres", "rt3_scene_weekend", "rt3_scene_stress", "rt3_scene_cornell", "rt3_hash_u32",
    "rt3_random_float", "rt3_debug_arith", "rt3_debug_force_plain_mode_r",
    "rt3_mesh_begin", "rt3_mesh_put", "rt3_mesh_sphere", "rt3_mesh_commit", "rt3_mesh_download",
    "rt3_render_path_range", "rt3_render_path_range_device", "rt3_accum_download", "rt3_accum_upload", "rt3_gather_rows",
    "rt3_stream", "rt3_synchronize", "rt3_device_alloc_words", "rt3_device_free", "rt3_device_read_words", "rt3_debug_force_brute",
    "rt3_abi_version", "rt3_debug_force_flat_filter", "rt3_gather_plan",
]
ABI_VERSION = 3          # RT3_ABI_VERSION of include/rt3.h these bindings (the STATS / PARAMS struct layouts below) were written against

_lib = None


def lib():
    """Loads librt3hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C raytracer-3_amd``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Fatal("librt3hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                    "there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 under the same SONAME.  If torch is going to
    # be used in this process (device buffers, streams, RCCL) it has to be loaded first, or its later CUDA init fails
    # with "No HIP GPUs are available"; librt3hip.so then binds to the runtime that is already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int
    sigs = {
        "rt3_create": (vp, [i32]), "rt3_destroy": (None, [vp]), "rt3_last_error": (C.c_char_p, [vp]),
        "rt3_set_sample_storage_cap": (i32, [vp, u64]),
        "rt3_set_mesh": (i32, [vp, vp, u32, vp, u32, vp]), "rt3_set_spheres": (i32, [vp, vp, vp, u32]),
        "rt3_render": (i32, [vp, vp, u32, u32, vp]), "rt3_render_device": (i32, [vp, vp, u32, u32, vp, vp]),
        "rt3_render_path": (i32, [vp, vp, vp, vp]), "rt3_render_path_device": (i32, [vp, vp, vp, vp, vp]),
        "rt3_rows_owned": (u32, [vp]), "rt3_row_of_local": (u32, [vp, u32]), "rt3_get_stats": (i32, [vp, vp]),
        "rt3_prerender_triangle": (None, [vp, vp, vp, vp, vp, vp]),
        "rt3_sphere_face_count": (u32, [u32, u32]), "rt3_sphere_vertex_count": (u32, [u32, u32]),
        "rt3_prerender_sphere": (None, [vp, f32, u32, u32, vp, vp, vp]),
        "rt3_object_count": (i32, [C.c_char_p, vp, vp]),
        "rt3_prerender_object": (i32, [C.c_char_p, vp, f32, vp, vp, u32, vp, u32]),
        "rt3_transfer_entity": (None, [vp, vp, vp, vp, vp, u32, vp, u32]),
        "rt3_camera_update": (None, [vp, f32, f32, f32]), "rt3_camera_look_at": (None, [vp, vp, vp, vp, f32, f32, f32]),
        "rt3_frame_ppm_bytes": (u64, [vp, u32, u32, vp, u64]), "rt3_frame_to_ppm": (i32, [vp, u32, u32, C.c_char_p]),
        "rt3_scene_three_spheres": (u32, [vp, vp, u32]), "rt3_scene_weekend": (u32, [u32, vp, vp, u32]),
        "rt3_scene_stress": (u32, [u32, u32, vp, vp, u32]), "rt3_scene_cornell": (u32, [u32, vp, vp, vp, u32]),
        "rt3_hash_u32": (u32, [u32]), "rt3_random_float": (f32, [u32]),
        "rt3_debug_arith": (i32, [vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]),
        "rt3_debug_force_plain_mode_r": (i32, [vp, i32]),
        "rt3_mesh_begin": (i32, [vp, u32, u32]), "rt3_mesh_put": (i32, [vp, vp, u32, vp, u32, u32, u32]),
        "rt3_mesh_sphere": (i32, [vp, vp, f32, u32, u32, vp, u32, u32]), "rt3_mesh_commit": (i32, [vp, vp]),
        "rt3_mesh_download": (i32, [vp, vp, vp]),
        "rt3_render_path_range": (i32, [vp, vp, vp, u32, u32, vp]),
        "rt3_render_path_range_device": (i32, [vp, vp, vp, u32, u32, vp, vp]),
        "rt3_accum_download": (i32, [vp, vp, vp, vp]), "rt3_accum_upload": (i32, [vp, vp, vp, vp, vp, u32]),
        "rt3_gather_rows": (i32, [vp, vp, vp, vp, vp, vp]), "rt3_stream": (vp, [vp]), "rt3_synchronize": (i32, [vp]),
        "rt3_device_alloc_words": (vp, [vp, u64]), "rt3_device_free": (None, [vp, vp]),
        "rt3_device_read_words": (i32, [vp, vp, u64, vp]), "rt3_debug_force_brute": (i32, [vp, i32]),
        "rt3_abi_version": (u32, []), "rt3_debug_force_flat_filter": (i32, [vp, i32]), "rt3_gather_plan": (i32, [vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.rt3_abi_version() != ABI_VERSION:
        raise Fatal("%s reports ABI version %d, these bindings were written against %d (include/rt3.h: RT3_ABI_VERSION)"
                    % (LIB_PATH, L.rt3_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


# --------------------------------------------------------------------------------------------------------
# Frame / Camera  (src/lib/camera/Frame.hpp:41-82, Camera.hpp:25-68)
# --------------------------------------------------------------------------------------------------------
class Frame:
    """uint32 RGBA8 frame, row 0 on top, word = 0xFF | B<<8 | G<<16 | R<<24 (SequentialRenderer.cpp:297)."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.data = np.zeros((self.height, self.width), np.uint32)

    def w(self):
        return self.width

    def h(self):
        return self.height

    def d(self):
        return self.data

    def ppm_bytes(self):
        """Exactly the bytes Frame::to_ppm writes (Frame.cpp:125-143)."""
        need = lib().rt3_frame_ppm_bytes(_p(self.data), self.width, self.height, None, 0)
        buf = np.zeros(need, np.uint8)
        lib().rt3_frame_ppm_bytes(_p(self.data), self.width, self.height, _p(buf), need)
        return buf.tobytes()

    def to_ppm(self, path):
        if lib().rt3_frame_to_ppm(_p(self.data), self.width, self.height, os.fsencode(path)) != 0:
            raise Fatal("Could not open '%s'" % path)

    def rgb(self):
        """(h, w, 3) uint8 view of the frame as PPM/PNG would show it."""
        d = self.data
        return np.stack([(d >> 24) & 0xFF, (d >> 16) & 0xFF, (d >> 8) & 0xFF], axis=-1).astype(np.uint8)


class Camera:
    """Pinhole camera owning a Frame; four public vectors as in Camera.hpp:27-34."""

    def __init__(self):
        self.c = rt3_camera()
        self.frame = None

    def update(self, width, height, focal_length, viewport_width, viewport_height):
        """Camera::update (Camera.cpp:77-96): origin 0, axis-aligned viewport, new Frame(width, height)."""
        self.frame = Frame(width, height)
        lib().rt3_camera_update(C.byref(self.c), np.float32(focal_length), np.float32(viewport_width),
                                np.float32(viewport_height))
        return self

    def look_at(self, width, height, look_from, look_at, vup=(0.0, 1.0, 0.0), vfov=20.0, focus_dist=1.0):
        """Extension: the book's look-from/look-at camera in the reference's four vectors."""
        self.frame = Frame(width, height)
        aspect = np.float32(np.float32(width) / np.float32(height))
        lib().rt3_camera_look_at(C.byref(self.c), _f3(look_from), _f3(look_at), _f3(vup), np.float32(vfov), aspect,
                                 np.float32(focus_dist))
        return self

    @property
    def origin(self):
        return tuple(self.c.origin)

    @property
    def horizontal(self):
        return tuple(self.c.horizontal)

    @property
    def vertical(self):
        return tuple(self.c.vertical)

    @property
    def lower_left_corner(self):
        return tuple(self.c.lower_left_corner)

    def w(self):
        return self.frame.w()

    def h(self):
        return self.frame.h()

    def get_frame(self):
        return self.frame


def main_camera(width, height):
    """The camera Main.cpp:272 builds: update(W, H, 2.0, (float(W)/float(H))*2.0f, 2.0f)."""
    vw = np.float32(np.float32(width) / np.float32(height)) * np.float32(2.0)
    return Camera().update(width, height, 2.0, vw, 2.0)


# --------------------------------------------------------------------------------------------------------
# Entities  (src/lib/entities/*.hpp) — plain records; pre-rendering happens in HipRenderer.prerender
# --------------------------------------------------------------------------------------------------------
et_none, et_triangle, et_sphere, et_object, et_analytic_sphere = 0, 1, 2, 3, 4
eprmf_none, eprmf_cpu, eprmf_gpu = 0, 1, 2
epro_none, epro_generate_triangle, epro_generate_sphere, epro_load_object_file = 0, 1, 2, 3


class RenderEntity:
    """RenderEntity.hpp:76-89."""
    type = et_none
    pre_render_mode = eprmf_cpu
    pre_render_operation = epro_none
    pre_render_faces = 0
    pre_render_vertices = 0
    material = None          # extension: rt3 material (kind, rgb, param); None = the reference's flat colour


def _material(kind, rgb, param=0.0):
    m = np.zeros(1, MATERIAL)
    m["rgb"][0] = rgb
    m["param"][0] = param
    m["kind"][0] = kind
    return m


def create_triangle(p1, p2, p3, color, material=None):
    """ECS::create_triangle (Triangle.cpp:28-53)."""
    e = RenderEntity()
    e.type, e.pre_render_operation = et_triangle, epro_generate_triangle
    e.pre_render_faces, e.pre_render_vertices = 1, 3
    e.points, e.color, e.material = (tuple(p1), tuple(p2), tuple(p3)), tuple(color), material
    return e


def create_sphere(center, radius, n_meridians, n_parallels, color, material=None):
    """ECS::create_sphere (Sphere.cpp:87-115).  Extension: n_meridians == n_parallels == 0 asks for an analytic
    sphere (Mode X) instead of a tessellation."""
    e = RenderEntity()
    e.center, e.radius, e.color, e.material = tuple(center), float(radius), tuple(color), material
    e.n_meridians, e.n_parallels = int(n_meridians), int(n_parallels)
    if n_meridians == 0 and n_parallels == 0:
        e.type = et_analytic_sphere
        return e
    e.type, e.pre_render_operation = et_sphere, epro_generate_sphere
    e.pre_render_mode = eprmf_cpu | eprmf_gpu          # Sphere.cpp:94-98: spheres can also be tessellated on the GPU
    e.pre_render_faces = lib().rt3_sphere_face_count(n_meridians, n_parallels)
    e.pre_render_vertices = lib().rt3_sphere_vertex_count(n_meridians, n_parallels)
    return e


def create_object(file_path, center, scale, color, material=None):
    """ECS::create_object (Object.cpp:54-126): opens the file once to count faces and vertices."""
    e = RenderEntity()
    e.type, e.pre_render_operation = et_object, epro_load_object_file
    e.file_path, e.center, e.scale, e.color, e.material = file_path, tuple(center), np.float32(scale), tuple(color), material
    nf, nv = C.c_uint32(), C.c_uint32()
    rc = lib().rt3_object_count(os.fsencode(file_path), C.byref(nf), C.byref(nv))
    if rc != 0:
        raise Fatal("Could not open file or unreadable line: %s" % file_path)
    e.pre_render_faces, e.pre_render_vertices = nf.value, nv.value
    return e


def lambertian(rgb):
    return _material(MAT_LAMBERT, rgb)


def metal(rgb, fuzz):
    return _material(MAT_METAL, rgb, fuzz)


def dielectric(ior):
    return _material(MAT_DIELECTRIC, (1.0, 1.0, 1.0), ior)


def emissive(rgb):
    return _material(MAT_FLAT, rgb)


def pre_render_entity(e):
    """cpu_pre_render_{triangle,sphere,object}: entity -> (GFace[], vec4[])."""
    faces = np.zeros(e.pre_render_faces, GFACE)
    verts = np.zeros((e.pre_render_vertices, 4), np.float32)
    if e.pre_render_operation == epro_generate_triangle:
        lib().rt3_prerender_triangle(_f3(e.points[0]), _f3(e.points[1]), _f3(e.points[2]), _f3(e.color), _p(faces), _p(verts))
    elif e.pre_render_operation == epro_generate_sphere:
        lib().rt3_prerender_sphere(_f3(e.center), np.float32(e.radius), e.n_meridians, e.n_parallels, _f3(e.color),
                                   _p(faces), _p(verts))
    elif e.pre_render_operation == epro_load_object_file:
        rc = lib().rt3_prerender_object(os.fsencode(e.file_path), _f3(e.center), e.scale, _f3(e.color), _p(faces),
                                        len(faces), _p(verts), len(verts))
        if rc != 0:
            raise Fatal("Could not load object file '%s'" % e.file_path)
    else:
        raise Fatal("Entity wants to be pre-rendered using unsupported operation %d" % e.pre_render_operation)
    return faces, verts


def merge_entities(parts):
    """SequentialRenderer::transfer_entity over a list of (faces, verts): indices rebased by running vertex count."""
    nf = sum(len(f) for f, _ in parts)
    nv = sum(len(v) for _, v in parts)
    faces = np.zeros(nf, GFACE)
    verts = np.zeros((nv, 4), np.float32)
    cf, cv = C.c_uint32(0), C.c_uint32(0)
    for f, v in parts:
        f = np.ascontiguousarray(f)
        v = np.ascontiguousarray(v, np.float32)
        lib().rt3_transfer_entity(_p(faces), C.byref(cf), _p(verts), C.byref(cv), _p(f), len(f), _p(v), len(v))
    return faces, verts


# --------------------------------------------------------------------------------------------------------
# benchmark scenes
# --------------------------------------------------------------------------------------------------------
def _sphere_scene(fn, *args):
    n = fn(*args, None, None, 0)
    cr = np.zeros((n, 4), np.float32)
    mats = np.zeros(n, MATERIAL)
    got = fn(*args, _p(cr), _p(mats), n)
    assert got == n
    return cr, mats


def scene_three_spheres():
    return _sphere_scene(lib().rt3_scene_three_spheres)


def scene_weekend(seed=42):
    return _sphere_scene(lib().rt3_scene_weekend, seed)


def scene_stress(n=100000, seed=43):
    return _sphere_scene(lib().rt3_scene_stress, n, seed)


def scene_cornell(grid=64):
    n = lib().rt3_scene_cornell(grid, None, None, None, 0)
    faces = np.zeros(n, GFACE)
    verts = np.zeros((3 * n, 4), np.float32)
    mats = np.zeros(n, MATERIAL)
    got = lib().rt3_scene_cornell(grid, _p(faces), _p(verts), _p(mats), n)
    assert got == n
    return faces, verts, mats


def weekend_camera(width, height):
    """Book final-scene camera: from (13,2,3) at (0,0,0), vfov 20, focus distance 10."""
    return Camera().look_at(width, height, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, 10.0)


def make_params(width, height, spp=1, max_depth=1, seed=1, flags=0, lens_radius=0.0, t_min=0.001,
                tile_rows=8, tile_index=0, tile_count=1):
    return rt3_params(width, height, spp, max_depth, seed, flags, lens_radius, t_min, tile_rows, tile_index, tile_count)


# --------------------------------------------------------------------------------------------------------
# Renderer  (src/lib/renderer/Renderer.hpp:34-63)
# --------------------------------------------------------------------------------------------------------
class Renderer:
    """Abstract backend API of the reference."""

    def prerender(self, entities):
        raise NotImplementedError

    def render(self, camera):
        raise NotImplementedError


class HipRenderer(Renderer):
    """The MI355X backend.  Mode R by default; ``configure(spp=..., max_depth=...)`` switches render() to Mode X."""

    def __init__(self, device=0):
        self._ctx = lib().rt3_create(device)
        if not self._ctx:
            raise Fatal(lib().rt3_last_error(None).decode())
        self._path = None            # rt3_params template when Mode X is requested
        self.n_faces = 0
        self.n_spheres = 0
        self._mesh_counts = (0, 0)

    def close(self):
        if getattr(self, "_ctx", None):
            lib().rt3_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise Fatal(lib().rt3_last_error(self._ctx).decode())

    # -- scene ---------------------------------------------------------------------------------------
    def prerender(self, entities, gpu_prerender=False):
        """Renderer::prerender: flatten the entities (order = array order) into the device buffers.  Replaces the old
        scene.  With gpu_prerender, entities flagged eprmf_gpu (tessellated spheres, as in the reference's Vulkan build)
        are generated on the device at their running offsets (VulkanRenderer.cpp:305-327); everything else is
        pre-rendered on the host and transferred (VulkanRenderer.cpp:355-387)."""
        mesh_ents, spheres, smats = [], [], []
        for i, e in enumerate(entities):
            if e.type == et_analytic_sphere:
                spheres.append((e.center[0], e.center[1], e.center[2], e.radius))
                smats.append(e.material if e.material is not None else _material(MAT_FLAT, e.color))
            elif not (e.pre_render_mode & (eprmf_cpu | eprmf_gpu)):
                raise Fatal("Entity %d cannot be pre-rendered by this back-end." % i)
            else:
                mesh_ents.append(e)
        nf = sum(e.pre_render_faces for e in mesh_ents)
        nv = sum(e.pre_render_vertices for e in mesh_ents)
        any_mat = any(e.material is not None for e in mesh_ents)
        self._check(lib().rt3_mesh_begin(self._ctx, nf, nv))
        fo = vo = 0
        part_mats = []
        for e in mesh_ents:
            if gpu_prerender and (e.pre_render_mode & eprmf_gpu) and e.pre_render_operation == epro_generate_sphere:
                self._check(lib().rt3_mesh_sphere(self._ctx, _f3(e.center), np.float32(e.radius), e.n_meridians, e.n_parallels,
                                                  _f3(e.color), fo, vo))
                colors = None
            else:
                f, v = pre_render_entity(e)
                self._check(lib().rt3_mesh_put(self._ctx, _p(f), len(f), _p(v), len(v), fo, vo))
                colors = f["color"]
            if any_mat:
                if e.material is not None:
                    part_mats.append(np.repeat(e.material, e.pre_render_faces))
                else:
                    if colors is None:
                        raise Fatal("a device-tessellated sphere needs a material when other entities have one")
                    m = np.zeros(e.pre_render_faces, MATERIAL)
                    m["rgb"] = colors
                    m["kind"] = MAT_FLAT
                    part_mats.append(m)
            fo += e.pre_render_faces
            vo += e.pre_render_vertices
        fmats = np.ascontiguousarray(np.concatenate(part_mats)) if (mesh_ents and any_mat) else None
        self._check(lib().rt3_mesh_commit(self._ctx, _p(fmats)))
        self.n_faces = nf
        self._mesh_counts = (nf, nv)
        if spheres:
            self.set_spheres(np.array(spheres, np.float32), np.concatenate(smats))
        else:
            self.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, MATERIAL))

    def mesh_download(self):
        """The merged GFace[] / vec4[] as they sit on the device (after prerender / set_mesh)."""
        nf, nv = getattr(self, "_mesh_counts", (0, 0))
        faces = np.zeros(nf, GFACE)
        verts = np.zeros((nv, 4), np.float32)
        self._check(lib().rt3_mesh_download(self._ctx, _p(faces), _p(verts)))
        return faces, verts

    def set_mesh(self, faces, verts, face_materials=None):
        faces = np.ascontiguousarray(faces)
        verts = np.ascontiguousarray(verts, np.float32)
        assert faces.dtype == GFACE
        if face_materials is not None:
            face_materials = np.ascontiguousarray(face_materials)
            assert face_materials.dtype == MATERIAL and len(face_materials) == len(faces)
        self._check(lib().rt3_set_mesh(self._ctx, _p(faces), len(faces), _p(verts), len(verts), _p(face_materials)))
        self.n_faces = len(faces)
        self._mesh_counts = (len(faces), len(verts))

    def set_spheres(self, center_radius, materials):
        cr = np.ascontiguousarray(center_radius, np.float32).reshape(-1, 4)
        materials = np.ascontiguousarray(materials)
        assert materials.dtype == MATERIAL and len(materials) == len(cr)
        self._check(lib().rt3_set_spheres(self._ctx, _p(cr), _p(materials), len(cr)))
        self.n_spheres = len(cr)

    # -- render --------------------------------------------------------------------------------------
    def configure(self, spp=None, max_depth=None, seed=1, flags=0, lens_radius=0.0, t_min=0.001):
        """spp=None returns render() to Mode R."""
        self._path = None if spp is None else dict(spp=spp, max_depth=max_depth or 1, seed=seed, flags=flags,
                                                   lens_radius=lens_radius, t_min=t_min)

    def render(self, camera):
        """Renderer::render: writes the pixels into camera.get_frame().d()."""
        frame = camera.get_frame()
        if self._path is None:
            self._check(lib().rt3_render(self._ctx, C.byref(camera.c), frame.w(), frame.h(), _p(frame.data)))
        else:
            p = make_params(frame.w(), frame.h(), **self._path)
            self._check(lib().rt3_render_path(self._ctx, C.byref(camera.c), C.byref(p), _p(frame.data)))

    def render_path(self, camera_c, params):
        """Mode X with explicit params (tiles included); returns the compact (rows_owned, width) pixel array."""
        rows = lib().rt3_rows_owned(C.byref(params))
        out = np.zeros((rows, params.width), np.uint32)
        self._check(lib().rt3_render_path(self._ctx, C.byref(camera_c), C.byref(params), _p(out)))
        return out

    def render_path_range(self, camera_c, params, sample_begin, sample_count):
        """Progressive Mode X (rt3_render_path_range): adds samples [begin, begin + count) to the accumulation this renderer
        keeps and returns the frame resolved over the samples so far.  begin == 0 starts over."""
        rows = lib().rt3_rows_owned(C.byref(params))
        out = np.zeros((rows, params.width), np.uint32)
        self._check(lib().rt3_render_path_range(self._ctx, C.byref(camera_c), C.byref(params), sample_begin, sample_count, _p(out)))
        return out

    def render_path_range_device(self, camera_c, params, sample_begin, sample_count, d_out_ptr, stream_ptr=None):
        self._check(lib().rt3_render_path_range_device(self._ctx, C.byref(camera_c), C.byref(params), sample_begin, sample_count,
                                                       C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr or 0)))

    def accum_download(self, params, want_sq=False):
        """Checkpoint of the accumulation: (sum[rows, w, 4], sum_sq or None, samples_done)."""
        rows = lib().rt3_rows_owned(C.byref(params))
        acc = np.zeros((rows, params.width, 4), np.float32)
        sq = np.zeros((rows, params.width, 4), np.float32) if want_sq else None
        done = C.c_uint32(0)
        self._check(lib().rt3_accum_download(self._ctx, _p(acc), _p(sq), C.byref(done)))
        return acc, sq, done.value

    def accum_upload(self, camera_c, params, acc, sq, samples_done):
        acc = np.ascontiguousarray(acc, np.float32)
        sq = None if sq is None else np.ascontiguousarray(sq, np.float32)
        self._check(lib().rt3_accum_upload(self._ctx, C.byref(camera_c), C.byref(params), _p(acc), _p(sq), samples_done))

    def gather_rows(self, d_frame_ptr, shard_renderer, d_tile_ptr, shard_params, stream_ptr=None):
        """rt3_gather_rows: the shard's compact rows -> their interleaved places in this (root) renderer's device frame."""
        self._check(lib().rt3_gather_rows(self._ctx, C.c_void_p(d_frame_ptr), shard_renderer._ctx, C.c_void_p(d_tile_ptr),
                                          C.byref(shard_params), C.c_void_p(stream_ptr or 0)))

    def force_brute(self, on):
        """Tests / fuzzers: the unfiltered Mode-X kernel (every ray against every primitive)."""
        self._check(lib().rt3_debug_force_brute(self._ctx, 1 if on else 0))

    def force_flat_filter(self, on):
        """Tests / A-B: the tiled matrix-filter kernel with one row per primitive instead of the two-level filter's group rows (same pixels)."""
        self._check(lib().rt3_debug_force_flat_filter(self._ctx, 1 if on else 0))

    def render_path_device(self, camera_c, params, d_out_ptr, stream_ptr=None):
        """Asynchronous Mode X into a device buffer (e.g. a torch tensor's data_ptr()) on a HIP stream."""
        self._check(lib().rt3_render_path_device(self._ctx, C.byref(camera_c), C.byref(params), C.c_void_p(d_out_ptr),
                                                 C.c_void_p(stream_ptr or 0)))

    def render_device(self, camera_c, width, height, d_out_ptr, stream_ptr=None):
        self._check(lib().rt3_render_device(self._ctx, C.byref(camera_c), width, height, C.c_void_p(d_out_ptr),
                                            C.c_void_p(stream_ptr or 0)))

    def set_sample_storage_cap(self, nbytes):
        self._check(lib().rt3_set_sample_storage_cap(self._ctx, nbytes))

    def stats(self):
        s = rt3_stats()
        self._check(lib().rt3_get_stats(self._ctx, C.byref(s)))
        return s

    def force_plain_mode_r(self, on):
        self._check(lib().rt3_debug_force_plain_mode_r(self._ctx, 1 if on else 0))

    def debug_arith(self, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        n = len(a)
        outs = [np.zeros(n, np.float32) for _ in range(5)] + [np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)]
        self._check(lib().rt3_debug_arith(self._ctx, _p(a), _p(b), n, *[_p(o) for o in outs]))
        return outs


def initialize_renderer(device=0):
    """RayTracer::initialize_renderer (Renderer.hpp:63): the link-time factory; here it always builds the HIP backend."""
    return HipRenderer(device)


def rows_owned(params):
    return lib().rt3_rows_owned(C.byref(params))


def row_of_local(params, local_row):
    return lib().rt3_row_of_local(C.byref(params), local_row)


def deinterleave(tiles, params_list, height, width):
    """Assembles the full frame from the compact per-shard row buffers (the host side of the RCCL gather)."""
    frame = np.zeros((height, width), np.uint32)
    for tile, p in zip(tiles, params_list):
        for r in range(tile.shape[0]):
            frame[row_of_local(p, r)] = tile[r]
    return frame
