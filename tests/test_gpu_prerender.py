"""Device-side scene assembly: the HIP equivalents of pre_render_sphere_v2_vertices/faces.glsl write a
tessellated sphere straight into the merged device buffers at running offsets.  Parity criterion = the one the reference's
author left in VulkanRenderer.cpp:329-353: indices exact, normals within 1e-6 of cpu_pre_render_sphere — and, because the
device follows the CPU's double-precision trig rather than the shader's float trig, the arrays are expected to be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mp", [(8, 8), (3, 3), (16, 9), (100, 57), (256, 256)])
def test_device_tessellation_matches_cpu_pre_render(rt3, renderer, oracle, mp):
    m, p = mp
    args = ((-2.0, 0.3, -5.0), 1.25, m, p, (0.2, 0.4, 1.0))
    renderer.prerender([rt3.create_sphere(*args)], gpu_prerender=True)
    faces, verts = renderer.mesh_download()
    of, ov = oracle.prerender_sphere(*args)
    assert len(faces) == len(of) and len(verts) == len(ov)
    for k in ("v1", "v2", "v3"):
        assert np.array_equal(faces[k], of[k])                                   # indices: exact
    assert np.abs(verts - ov).max() <= 1e-6 and np.abs(faces["normal"] - of["normal"]).max() <= 1e-6   # the author's tolerance
    assert np.abs(faces["color"] - of["color"]).max() <= 1e-6
    same = verts.tobytes() == ov.tobytes() and faces.tobytes() == of.tobytes()
    assert same, "device double trig differs from libm somewhere: %d vertex floats differ" % int((verts != ov).sum())


def test_mixed_scene_offsets_and_render(rt3, renderer, oracle):
    """Host-pre-rendered and device-tessellated entities interleaved: running offsets, index rebasing, then Mode R."""
    ents = [rt3.create_triangle((1.5, -0.5, -3.0), (0.2, -0.5, -3.0), (0.8, 0.9, -3.5), (1.0, 0.0, 0.0)),
            rt3.create_sphere((-0.5, 0.0, -4.0), 1.0, 24, 17, (0.0, 0.0, 1.0)),
            rt3.create_triangle((-2.5, -0.5, -3.0), (-1.6, -0.5, -3.0), (-2.0, 0.9, -3.5), (0.0, 1.0, 0.0)),
            rt3.create_sphere((1.0, 0.6, -6.0), 1.5, 40, 40, (0.2, 0.9, 0.1))]
    renderer.prerender(ents, gpu_prerender=True)
    faces, verts = renderer.mesh_download()
    parts = [oracle.prerender_triangle((1.5, -0.5, -3.0), (0.2, -0.5, -3.0), (0.8, 0.9, -3.5), (1.0, 0.0, 0.0)),
             oracle.prerender_sphere((-0.5, 0.0, -4.0), 1.0, 24, 17, (0.0, 0.0, 1.0)),
             oracle.prerender_triangle((-2.5, -0.5, -3.0), (-1.6, -0.5, -3.0), (-2.0, 0.9, -3.5), (0.0, 1.0, 0.0)),
             oracle.prerender_sphere((1.0, 0.6, -6.0), 1.5, 40, 40, (0.2, 0.9, 0.1))]
    of, ov = oracle.merge(parts)
    assert faces.tobytes() == of.tobytes() and verts.tobytes() == ov.tobytes()
    renderer.configure(spp=None)
    cam = rt3.main_camera(320, 180)
    renderer.render(cam)
    assert np.array_equal(cam.get_frame().d(), oracle.render_mode_r(of, ov, oracle.camera_update(320, 180), 320, 180))
    # the host path gives the same device scene
    renderer.prerender(ents, gpu_prerender=False)
    f2, v2 = renderer.mesh_download()
    assert f2.tobytes() == faces.tobytes() and v2.tobytes() == verts.tobytes()


def test_mesh_put_bounds_are_checked(rt3, renderer):
    L = rt3.lib()
    assert L.rt3_mesh_begin(renderer._ctx, 4, 6) == 0
    f = np.zeros(3, rt3.GFACE)
    v = np.zeros((4, 4), np.float32)
    assert L.rt3_mesh_put(renderer._ctx, f.ctypes.data, 3, v.ctypes.data, 4, 2, 0) != 0        # 2 + 3 faces > 4
    assert b"does not fit" in L.rt3_last_error(renderer._ctx)
    import ctypes as C
    c3 = (C.c_float * 3)(0, 0, -3)
    assert L.rt3_mesh_sphere(renderer._ctx, c3, 1.0, 8, 8, c3, 0, 0) != 0                      # 96 faces > 4
    assert L.rt3_mesh_begin(renderer._ctx, 0, 0) == 0 and L.rt3_mesh_commit(renderer._ctx, None) == 0
