"""The arithmetic of the block-conversion decode (rt3_matrix_filter.hpp, mfma32k_scan_tile / cand_bit / pair_decode), replayed on the CPU.

v_cvt_scalef32_2xpk16_fp6_f32 puts input a[i] into six-bit field 2 i and b[i] into field 2 i + 1 of a 192-bit string, the sign of field f at bit
6 f + 5 (tools/ubench_fp6_decode.hip measured that on the hardware: profiles/r03_ubench_fp6_decode.log).  The kernel merges the three words of each
half with v_bfi_b32 under two masks, shifts the second half down by one and merges again; cand_bit() maps a bit of the resulting word back to
(ray group G, row half h, accumulator j).  This test rebuilds the word bit by bit from the layout and checks masks, formula and round trip —
no GPU, no library: if somebody edits one of the constants, the CPU suite says so."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "raytracer-3_amd", "csrc", "rt3_matrix_filter.hpp")).read()


def sign_bit_of_input(which, i):
    """Bit (0..191) that holds the sign of a[i] (which = 0) or b[i] (which = 1)."""
    return 6 * (2 * i + which) + 5


def merged_word(signs_a, signs_b, m0, m01, mall):
    """The kernel's merge: words r0..r5 hold garbage except at the sign bits; bfi(mask, x, y) = (x & mask) | (y & ~mask)."""
    r = [0xFFFFFFFF ^ 0] * 6                                           # garbage everywhere ...
    r = [0x5A5A5A5A, 0xC3C3C3C3, 0x0F0F0F0F, 0x96969696, 0x3C3C3C3C, 0xA5A5A5A5]
    for which, signs in ((0, signs_a), (1, signs_b)):
        for i, sg in enumerate(signs):
            bit = sign_bit_of_input(which, i)
            w, b = divmod(bit, 32)
            r[w] = (r[w] & ~(1 << b)) | (sg << b)                       # ... except where the conversion put a sign
    bfi = lambda m, x, y: (x & m) | (y & ~m & 0xFFFFFFFF)
    t = bfi(m01, bfi(m0, r[0], r[1]), r[2])
    v = bfi(m01, bfi(m0, r[3], r[4]), r[5])
    return bfi(mall, t, v >> 1)


def cand_bit(p):
    """cand_bit<true>() of the header."""
    f = ((11 * (p >> 1) + 10) & 15) | ((~p & 1) << 4)
    return f >> 3, f & 1, (f >> 1) & 3                                  # G, h, j


def test_masks_in_the_source_are_the_sign_positions():
    m = re.search(r"bfi32\((0x[0-9A-Fa-f]+)u, bfi32\((0x[0-9A-Fa-f]+)u, r\[0\], r\[1\]\), r\[2\]\)", SRC)
    assert m, "merge expression not found"
    m01, m0 = int(m.group(1), 16), int(m.group(2), 16)
    mall = int(re.search(r"return bfi32\((0x[0-9A-Fa-f]+)u, t, v >> 1\)", SRC).group(1), 16)
    pos = [[], [], []]
    for f in range(16):                                                 # the first half: fields 0..15 in words 0..2
        w, b = divmod(6 * f + 5, 32)
        pos[w].append(b)
    assert m0 == sum(1 << b for b in pos[0]) == 0x20820820
    assert m01 == m0 | sum(1 << b for b in pos[1]) == 0x28A28A28
    assert sum(1 << b for b in pos[2]) == 0x82082082 and mall == m01 | 0x82082082 == 0xAAAAAAAA
    assert not set(pos[0]) & set(pos[1]) and not set(pos[1]) & set(pos[2]) and not set(pos[0]) & set(pos[2])     # disjoint: OR-able
    for f in range(16, 32):                                             # the second half repeats the pattern three words later
        w, b = divmod(6 * f + 5, 32)
        assert b in pos[w - 3]


def test_every_bit_of_the_word_is_one_input_and_cand_bit_finds_it():
    seen = set()
    for which in (0, 1):
        for i in range(16):
            a = [0] * 16
            b = [0] * 16
            (b if which else a)[i] = 1
            n = merged_word(a, b, 0x20820820, 0x28A28A28, 0xAAAAAAAA)
            assert bin(n).count("1") == 1, "garbage bits survived the merge"
            p = n.bit_length() - 1
            seen.add(p)
            G, h, j = cand_bit(p)
            # the kernel feeds lo = (d00, d01, d02, d03) as a and hi = (d10, ..) as b: input i of half `which` is accumulator i % 4 of ray group i / 4
            assert (G, h, j) == (i // 4, which, i % 4), (which, i, p, G, h, j)
    assert seen == set(range(32))
    assert merged_word([1] * 16, [1] * 16, 0x20820820, 0x28A28A28, 0xAAAAAAAA) == 0xFFFFFFFF          # no candidate: the word the scan counts as empty
    assert merged_word([0] * 16, [0] * 16, 0x20820820, 0x28A28A28, 0xAAAAAAAA) == 0


def test_pair_decode_restores_ray_lane_and_row():
    """pair_decode(): raw = pushing lane << 26 | row0 + 32 blk + bit -> (ray lane << 26 | row); lane (g, c) = (lane >> 4, lane & 15) holds, for ray
    group G, the results of rays 16 G + c against rows 32 blk + 16 h + 4 g + j."""
    for lane in range(64):
        for p in range(32):
            for base in (0, 32 * 7, 32 * 1000):
                raw = (lane << 26) | (base + p)
                G, h, j = cand_bit(raw & 31)
                frm = raw >> 26
                out = (((G << 4) + (frm & 15)) << 26) | ((raw & ((1 << 26) - 32)) + (h << 4) + (frm >> 4) * 4 + j)
                assert out >> 26 == 16 * G + (lane & 15) and out & ((1 << 26) - 1) == base + 16 * h + 4 * (lane >> 4) + j
    body = re.search(r"uint32_t pair_decode\(uint32_t raw\) \{(.*?)\n\}", SRC, re.S).group(1)
    assert "(G << 4) + (from & 15u)" in body and "(h << 4) + (from >> 4) * 4u + j" in body
