#!/usr/bin/env python3
"""ISA check around v_cvt_pk_bf16_f32 (profiles/r03_cvt_hazard.md; rt3_matrix_filter.hpp, pk_bf16()).

A build in which a vector-ALU instruction reads the result of v_cvt_pk_bf16_f32 in the very next issue slot loses candidates at random on the
MI355X (about 3 per 10^6 ray casts; every failing build seen has such a pair, every passing one has at least one instruction or wait state in
between).  hipcc has no hazard rule for the opcode, so nothing but the source (an inline-asm conversion followed by `s_nop 0`) keeps the pair
apart.  This script makes that a BUILD-TIME property:

    python tools/cvt_isa_check.py [listing.s]        exit 1 if any v_cvt_pk_bf16_f32 of the listing has a reader of its destination register
                                                     in the next issue slot; without an argument it compiles the product sources itself
                                                     (hipcc -S, device only) — tests/test_cvt_isa.py runs it that way in the CPU suite
    python tools/cvt_isa_check.py --table a.s b.s    per kernel: distance (wait states) from every conversion to its first reader, and back to the
                                                     last MFMA that READ the destination register as an A/B operand (the write-after-read window)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTR = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(?:;.*)?$")


def regs(operand_text):
    """Set of VGPR numbers named in an operand string: v12, v[4:7]."""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels(path):
    """name -> list of (mnemonic, destination registers, source registers, wait states the instruction itself takes)."""
    out, name, body = {}, None, None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z[\w]+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if ".end_amdhsa_kernel" in line or line.startswith("\t.section") and body:
            pass
        m = INSTR.match(line)
        if not m or m.group(1).startswith("."):
            continue
        op, rest = m.group(1), m.group(2)
        if op == "s_endpgm":
            out[name] = body
            name = None
            continue
        parts = [p.strip() for p in re.split(r",(?![^\[]*\])", rest)] if rest else []
        if op.startswith(("global_store", "ds_write", "buffer_store", "flat_store", "v_cmp", "s_")) or not parts:
            dst, src = set(), regs(rest)
        else:
            dst, src = regs(parts[0]), regs(",".join(parts[1:]))
            if op.startswith("v_mfma"):
                src = regs(",".join(parts[1:3]))                      # A and B only; the accumulator is read late (its own, known hazard class)
        states = int(rest.split()[0]) + 1 if op == "s_nop" and rest else 1
        body.append((op, dst, src, states))
    return out


def analyse(body):
    """For every conversion: (wait states to the first reader of its result, wait states back to the last MFMA reading the destination as A/B)."""
    rows = []
    for i, (op, dst, _src, _w) in enumerate(body):
        if op != "v_cvt_pk_bf16_f32":
            continue
        fwd, d = None, 0
        for op2, dst2, src2, w2 in body[i + 1:i + 400]:
            if dst & src2:
                fwd = d
                break
            if dst & dst2:
                break                                                  # overwritten before anybody read it
            d += w2
        back, d = None, 0
        for op2, _dst2, src2, w2 in reversed(body[max(0, i - 400):i]):
            if op2.startswith("v_mfma") and dst & src2:
                back = d
                break
            d += w2
        rows.append((fwd, back, body[i + 1][0] if i + 1 < len(body) else ""))
    return rows


def compile_product():
    out = os.path.join(tempfile.mkdtemp(prefix="rt3_isa_"), "rt3_device.s")
    src = os.path.join(ROOT, "raytracer-3_amd", "csrc", "rt3_device.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == "--table":
        for path in args[1:]:
            print("== %s" % path)
            for name, body in sorted(kernels(path).items()):
                rows = analyse(body)
                if not rows:
                    continue
                fwd = collections.Counter("none" if r[0] is None else min(r[0], 8) for r in rows)
                back = collections.Counter("none" if r[1] is None else ("<8" if r[1] < 8 else "<32" if r[1] < 32 else ">=32") for r in rows)
                print("  %-70s conversions %3d | wait states to first reader (8 = 8+): %s | back to last MFMA reading the register as A/B: %s"
                      % (name[:70], len(rows), dict(sorted(fwd.items(), key=str)), dict(back)))
        return 0
    path = args[0] if args else compile_product()
    bad = 0
    total = 0
    for name, body in sorted(kernels(path).items()):
        for fwd, _back, nxt in analyse(body):
            total += 1
            if fwd == 0:
                bad += 1
                print("%s: a %s reads the result of v_cvt_pk_bf16_f32 in the next issue slot" % (name, nxt))
    print("%d v_cvt_pk_bf16_f32 in %s, %d with a reader in the next issue slot" % (total, path, bad))
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
