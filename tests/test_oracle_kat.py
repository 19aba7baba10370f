"""Known-answer tests of the oracle's per-ray semantics (SURVEY.md test plan T4-T6): the predicates of ray_color
(SequentialRenderer.cpp:53-98) and of the analytic sphere hit (raytracer_v4.glsl:157-178 + the book's far root)."""
import math

import numpy as np

INF = float("inf")


def tri_scene(oracle, tris, colors):
    parts = [oracle.prerender_triangle(*t, c) for t, c in zip(tris, colors)]
    return oracle.merge(parts)


def test_tie_keeps_the_lower_face_index(oracle):
    # two coincident triangles, different colours: `t >= min_t` rejects the second (SequentialRenderer.cpp:71)
    t = ((1.0, -1.0, -3.0), (-1.0, -1.0, -3.0), (0.0, 1.0, -3.0))
    faces, verts = tri_scene(oracle, [t, t], [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0)])
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, 0.0, -1.0)) == (1.0, 0.0, 0.0)
    faces, verts = tri_scene(oracle, [t, t], [(0.0, 1.0, 0.0), (1.0, 0.0, 0.0)])
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, 0.0, -1.0)) == (0.0, 1.0, 0.0)
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, -1.0), faces=faces, verts=verts, tmin=0.0) == (1, 3.0, 0)


def test_edge_and_vertex_hits_are_accepted(oracle):
    # `>= 0.0` on all three edge tests (SequentialRenderer.cpp:87-89): a ray through an edge or a vertex hits
    t = ((1.0, 0.0, -2.0), (-1.0, 0.0, -2.0), (0.0, 1.0, -2.0))
    faces, verts = tri_scene(oracle, [t], [(1.0, 0.5, 0.25)])
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, 0.0, -1.0)) == (1.0, 0.5, 0.25)     # on the bottom edge
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, 1.0, -2.0)) == (1.0, 0.5, 0.25)     # through the apex
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, -0.001, -2.0)) != (1.0, 0.5, 0.25)  # just below: sky


def test_parallel_ray_and_behind_are_skipped(oracle):
    t = ((1.0, 0.0, -2.0), (-1.0, 0.0, -2.0), (0.0, 1.0, -2.0))
    faces, verts = tri_scene(oracle, [t], [(1.0, 0.0, 0.0)])
    sky = oracle.sky((1.0, 0.0, 0.0))
    assert oracle.ray_color(faces, verts, (0, 0, 0), (1.0, 0.0, 0.0)) == sky                  # n.d == 0 (:56)
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, 0.25, 1.0)) == oracle.sky((0.0, 0.25, 1.0))   # t < 0 (:71)


def test_degenerate_face_never_hits(oracle):
    # zero-area face: normal = 0 * inf = NaN (glm::normalize), every comparison with NaN is false
    t = ((0.0, 0.0, -2.0), (0.0, 0.0, -2.0), (0.0, 0.0, -2.0))
    faces, verts = tri_scene(oracle, [t], [(1.0, 0.0, 0.0)])
    assert np.isnan(faces["normal"]).all()
    assert oracle.ray_color(faces, verts, (0, 0, 0), (0.0, 0.0, -1.0)) == oracle.sky((0.0, 0.0, -1.0))
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, -1.0), faces=faces, verts=verts)[0] == 0


def test_reference_plus_sign_only_matters_off_origin(oracle):
    # SequentialRenderer.cpp:70 adds n.o instead of subtracting it: exact for origin 0, wrong elsewhere (kept, Mode R)
    t = ((1.0, -1.0, -3.0), (-1.0, -1.0, -3.0), (0.0, 1.0, -3.0))
    faces, verts = tri_scene(oracle, [t], [(1.0, 0.0, 0.0)])
    assert oracle.ray_color(faces, verts, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0)) == (1.0, 0.0, 0.0)
    # Mode X uses the corrected sign: from z = 1 the plane is 4 away
    assert oracle.nearest((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), faces=faces, verts=verts) == (1, 4.0, 0)


def test_sphere_known_answers(oracle):
    s = np.array([[0.0, 0.0, -1.0, 0.5]], np.float32)
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, -1.0), spheres=s) == (2, 0.5, 0)           # t = 0.5
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, 1.0), spheres=s)[0] == 0                    # sphere behind the ray
    assert oracle.nearest((0, 0, 0), (0.0, 1.0, 0.0), spheres=s)[0] == 0                    # miss
    # origin inside: raytracer_v4's near-root-only rule would miss; the book's rule returns the far root
    assert oracle.nearest((0.0, 0.0, -1.0), (0.0, 0.0, -1.0), spheres=s) == (2, 0.5, 0)
    # origin on the surface, leaving: t_min rejects the self hit (near root ~0) and the far root is behind
    assert oracle.nearest((0.0, 0.0, -0.5), (0.0, 0.0, 1.0), spheres=s)[0] == 0
    # origin on the surface, entering: self hit rejected, far root accepted (a refracted ray inside glass)
    kind, t, i = oracle.nearest((0.0, 0.0, -0.5), (0.0, 0.0, -1.0), spheres=s)
    assert (kind, i) == (2, 0) and abs(t - 1.0) < 1e-6
    # exact tangent (disc == 0) is a miss: the candidate test is strict
    assert oracle.nearest((0.5, 0.0, 0.0), (0.0, 0.0, -1.0), spheres=s)[0] == 0


def test_sphere_tie_and_order(oracle):
    two = np.array([[0.0, 0.0, -2.0, 0.5], [0.0, 0.0, -2.0, 0.5]], np.float32)
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, -1.0), spheres=two) == (2, 1.5, 0)         # equal t: lower index
    near_far = np.array([[0.0, 0.0, -5.0, 0.5], [0.0, 0.0, -2.0, 0.5]], np.float32)
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, -1.0), spheres=near_far) == (2, 1.5, 1)
    # a triangle at exactly the same t wins over the sphere (faces are tested first, raytracer_v4.glsl:226-246)
    tri = oracle.prerender_triangle((1.0, -1.0, -1.5), (-1.0, -1.0, -1.5), (0.0, 1.0, -1.5), (1, 0, 0))
    assert oracle.nearest((0, 0, 0), (0.0, 0.0, -1.0), spheres=two, faces=tri[0], verts=tri[1]) == (1, 1.5, 0)


def test_sincos_polynomial_accuracy(oracle):
    worst = 0.0
    for u in np.linspace(0.0, 1.0, 4097, endpoint=False, dtype=np.float32):
        c, s = oracle.sincos2pi(float(u))
        worst = max(worst, abs(c - math.cos(2 * math.pi * float(u))), abs(s - math.sin(2 * math.pi * float(u))))
    assert worst < 5e-7
    assert oracle.sincos2pi(0.0) == (1.0, 0.0) and oracle.sincos2pi(0.25) == (-0.0, 1.0) or oracle.sincos2pi(0.25) == (0.0, 1.0)


def test_random_stream_is_roughly_uniform(oracle):
    L = oracle.lib()
    v = np.array([L.oracle_random_float(L.oracle_hash_u32(7 ^ L.oracle_hash_u32(i))) for i in range(1, 20001)])
    assert 0.0 <= v.min() and v.max() < 1.0
    assert abs(v.mean() - 0.5) < 0.01 and abs(v.var() - 1.0 / 12.0) < 0.005
    assert L.oracle_random_float(L.oracle_hash_u32(0)) == 0.0      # why keys start at 1
