"""The C-ABI library loads, exports every symbol include/rt3.h declares, and refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rt3.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt3_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt3):
    L = rt3.lib()
    declared = header_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(rt3.EXPORTS) == declared          # the Python binding covers the whole header


def test_wire_struct_layouts(rt3):
    # GFace: Vertex.hpp:39-51 — 48 bytes, u32 x3 @0/4/8, normal @16, color @32
    assert rt3.GFACE.itemsize == 48
    assert [rt3.GFACE.fields[k][1] for k in ("v1", "v2", "v3", "normal", "color")] == [0, 4, 8, 16, 32]
    assert rt3.MATERIAL.itemsize == 20
    assert C.sizeof(rt3.rt3_camera) == 48           # 4 x vec3 (Camera.hpp:27-34)
    assert C.sizeof(rt3.rt3_params) == 44
    assert C.sizeof(rt3.rt3_stats) == 80 and rt3.ABI_VERSION == 3          # include/rt3.h: RT3_ABI_VERSION 3 (round 3: + filter_tests, bound_tests)


def test_no_cpu_fallback_without_gpu(rt3):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt3.Fatal, match="no CPU fallback"):
        rt3.initialize_renderer(0)
    assert rt3.lib().rt3_create(0) is None
    assert b"no HIP device" in rt3.lib().rt3_last_error(None)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "raytracer-3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for line in text.splitlines():
                    if re.search(r"(^\s*(import|from)\s.*oracle|#\s*include.*oracle|CDLL\(.*oracle|dlopen\(.*oracle|-lrt3oracle|oracle_lib)", line):
                        raise AssertionError("%s references the oracle: %s" % (f, line.strip()))


def test_rows_owned_partition(rt3):
    # every frame row belongs to exactly one shard, for ragged heights too
    for h, rows, n in ((1080, 8, 8), (225, 8, 3), (17, 4, 5), (9, 16, 2), (2160, 8, 8)):
        seen = np.zeros(h, np.int32)
        for i in range(n):
            p = rt3.make_params(64, h, tile_rows=rows, tile_index=i, tile_count=n)
            owned = rt3.rows_owned(p)
            ys = [rt3.row_of_local(p, k) for k in range(owned)]
            assert ys == sorted(ys) and all(0 <= y < h for y in ys)
            seen[ys] += 1
        assert (seen == 1).all()
