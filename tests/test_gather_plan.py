"""rt3_gather_plan — the pitch / offset arithmetic of rt3_gather_rows (the device-to-device gather of final pixels) as a pure
function: for many (height, width, tile_rows, shard count) shapes the copies of all shards, replayed with numpy on host arrays, must assemble exactly
the frame whose row y holds the rows of its owner (rt3_rows_owned / rt3_row_of_local are the ownership rule the render kernels use)."""
import itertools

import numpy as np
import pytest


def replay(rt3, plist, height, width):
    frame = np.full(height * width * 4, 0xEE, np.uint8)
    written = np.zeros(height * width * 4, np.uint8)
    for i, p in enumerate(plist):
        rows = rt3.rows_owned(p)
        tile = np.zeros((rows, width), np.uint32)
        for k in range(rows):
            tile[k, :] = (i << 24) | (rt3.row_of_local(p, k) << 8) | np.arange(width, dtype=np.uint32) % 251     # who, which frame row, which column
        tb = tile.view(np.uint8).reshape(-1)
        copies = rt3.gather_plan(p)
        assert len(copies) <= 2 and (rows > 0) == (len(copies) > 0)
        for c in copies:
            for r in range(c.rows):
                src = tb[c.src_offset + r * c.src_pitch:c.src_offset + r * c.src_pitch + c.row_bytes]
                assert len(src) == c.row_bytes, "copy reads past the tile"
                d0 = c.dst_offset + r * c.dst_pitch
                assert d0 + c.row_bytes <= frame.size, "copy writes past the frame"
                frame[d0:d0 + c.row_bytes] = src
                written[d0:d0 + c.row_bytes] += 1
    assert (written == 1).all(), "every byte of the frame is written exactly once"
    return frame.view(np.uint32).reshape(height, width)


SHAPES = [(h, w, tr, n) for h, w in ((1080, 1920), (225, 400), (17, 5), (9, 3), (2160, 64), (64, 7), (2, 2), (1000, 33))
          for tr in (1, 2, 3, 8, 16, 1080) for n in (1, 2, 3, 4, 5, 8, 13)]


@pytest.mark.parametrize("height,width,tile_rows,count", SHAPES)
def test_copies_of_all_shards_assemble_the_frame(rt3, height, width, tile_rows, count):
    plist = [rt3.make_params(width, height, tile_rows=tile_rows, tile_index=i, tile_count=count) for i in range(count)]
    frame = replay(rt3, plist, height, width)
    for y in range(height):
        owner = (y // tile_rows) % count if count > 1 else 0
        assert (frame[y] >> 24 == owner).all() and ((frame[y] >> 8) & 0xFFFF == y).all(), "row %d" % y


def test_bad_parameters_are_refused(rt3):
    p = rt3.make_params(64, 64, tile_rows=0, tile_index=0, tile_count=2)
    with pytest.raises(rt3.Fatal):
        rt3.gather_plan(p)
    p = rt3.make_params(64, 64, tile_rows=4, tile_index=5, tile_count=3)
    with pytest.raises(rt3.Fatal):
        rt3.gather_plan(p)
