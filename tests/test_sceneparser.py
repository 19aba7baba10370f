"""SceneLang front end (spec src/lib/sceneparser/SceneLang.md, sample src/lib/sceneparser/tests/test.scene —
the reference's own parser class is an empty stub).  `rt3 --scene file.scene --dump-scene` parses without a GPU."""
import numpy as np
import pytest

from test_cli import run

SAMPLE = """
/* close to the reference's tests/test.scene */
data {
    .obj triangle {
        v -1.0 0.0 0.0
        v 1.0 0.0 0.0
        v 0.0 1.0 0.0
        f 1 2 3
    }
    @suppress unused-data
    extern .obj teddy_bear: "bin/objects/teddy.obj";
}
global { float r: 0.5 * 2; vec3 tint: 0.0 1.0 0.0; uint cells: 4; }
// a second entities section continues the first (sections may repeat, spec section 2)
entities {
    triangle triangle_1 { p1: -1.0 0.0 -3.0; p2: 1.0 0.0 -3.0; p3: 0.0 1.0 -3.0; color: 1.0 0.0 0.0; }
    sphere sphere_1 { center: -1.5 0.5 -4.0; radius: global.r; n_meridians: 16; n_parallels: global.cells * 2 + 1; color: global.tint; }
}
entities {
    object triangle_2 { center: 2.0 -0.5 -4.0; scale: (float) 3 / 2; data: .obj triangle; color: 0.0 0.0 1.0; }
    sphere glass { vec3 center: 1.0 sphere_1.radius -2.0; float radius: 0.25; material: dielectric; ior: 1.5; }
}
"""


def dump(tmp_path, text, name="s.scene"):
    p = tmp_path / name
    p.write_text(text)
    return run("--scene", str(p), "--dump-scene")


def test_sample_scene_parses(tmp_path):
    rc, out, err = dump(tmp_path, SAMPLE)
    assert rc == 0, err
    lines = out.strip().splitlines()
    assert lines[0] == "triangle faces=1 vertices=3 p1=(-1,0,-3) p2=(1,0,-3) p3=(0,1,-3) color=(1,0,0)"
    assert lines[1] == "sphere faces=224 vertices=114 center=(-1.5,0.5,-4) radius=1 grid=16x9 color=(0,1,0)"
    assert lines[2] == "object faces=1 vertices=3 center=(2,-0.5,-4) scale=1.5 color=(0,0,1)"
    assert lines[3] == "analytic_sphere faces=0 vertices=0 center=(1,1,-2) radius=0.25 grid=0x0 color=(1,1,1) material=3 param=1.5"


def test_expressions_and_whitespace_rule(tmp_path):
    text = EXPRESSIONS
    rc, out, err = dump(tmp_path, text)
    assert rc == 0, err
    # `a -b` starts a new vec3 component, `a - b` and `a-b` subtract
    assert "p1=(13,20,-3) p2=(1,-2,3) p3=(-1,3,4.5) color=(2,0,0)" in out


def test_include_and_extern(tmp_path):
    (tmp_path / "mesh.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3\nf 2 4 3\n")
    (tmp_path / "common.scene").write_text('data { extern .obj quad: "%s"; }\nglobal { vec3 where: 0 0 -5; }\n' % (tmp_path / "mesh.obj"))
    rc, out, err = dump(tmp_path, '#include "common.scene"\nentities { object q { center: global.where; scale: 2.0; data: .obj quad; color: 1 1 0; } }')
    assert rc == 0, err
    assert out.strip() == "object faces=2 vertices=4 center=(0,0,-5) scale=2 color=(1,1,0)"


SCENE_ERRORS = [
    ("entities { cube c { } }", "unknown entity type 'cube'"),
    ("entities { sphere s { center: 0 0 -1; } }", "needs 'center' and 'radius'"),
    ("entities { triangle t { p1: 0 0 0; p2: 1 0 0; color: 1 0 0; } }", "lacks 'p3'"),
    ("entities { object o { center: 0 0 0; data: .obj nope; } }", "unknown data 'nope'"),
    ("entities { sphere a { center: 0 0 0; radius: 1; } sphere a { center: 0 0 0; radius: 1; } }", "duplicate entity identifier 'a'"),
    ("entities { sphere a { center: b.center; radius: 1; } }", "unknown reference 'b.center'"),
    ("lights { }", "unknown section 'lights'"),
    ('@error "stop here" entities { }', "@error: stop here"),
    ("entities { sphere a { center: 0 0 0; radius: 1 } }", "expected"),
    ("/* never closed", "unterminated comment"),
]
EXPRESSIONS = """global { float a: 2 + 3 * 4; float b: (2 + 3) * 4; int c: 7 / 2; float d: 7.0 / 2; float e: -(1 + 1) - -3; float f: 10 % 4; }
    entities { triangle t { p1: global.a - 1 global.b -global.c; p2: 1 -2 3; p3: 1 - 2 3 4.5; color: sqrt(4) 0 0; } }"""


@pytest.mark.parametrize("text,msg", SCENE_ERRORS)
def test_errors_are_fatal_with_file_and_line(tmp_path, text, msg):
    rc, out, err = dump(tmp_path, text)
    assert rc == -1 and msg in err and "s.scene:" in err


@pytest.mark.gpu
def test_scene_file_renders_like_the_same_entities_built_by_hand(tmp_path, rt3, oracle):
    p = tmp_path / "s.scene"
    p.write_text(SAMPLE.replace('sphere glass { vec3 center: 1.0 sphere_1.radius -2.0; float radius: 0.25; material: dielectric; ior: 1.5; }', ""))
    rc, out, err = run("-f", "ppm", "-W", "200", "-H", "112", "--scene", str(p), str(tmp_path / "out.ppm"))
    assert rc == 0, err
    tri_obj = tmp_path / "tri.obj"
    tri_obj.write_text("v -1.0 0.0 0.0\nv 1.0 0.0 0.0\nv 0.0 1.0 0.0\nf 1 2 3\n")
    parts = [oracle.prerender_triangle((-1.0, 0.0, -3.0), (1.0, 0.0, -3.0), (0.0, 1.0, -3.0), (1.0, 0.0, 0.0)),
             oracle.prerender_sphere((-1.5, 0.5, -4.0), 1.0, 16, 9, (0.0, 1.0, 0.0)),
             oracle.prerender_object(str(tri_obj), (2.0, -0.5, -4.0), np.float32(1.5), (0.0, 0.0, 1.0))]
    faces, verts = oracle.merge(parts)
    ref = oracle.render_mode_r(faces, verts, oracle.camera_update(200, 112), 200, 112)
    assert (tmp_path / "out.ppm").read_bytes() == oracle.ppm_bytes(ref)
