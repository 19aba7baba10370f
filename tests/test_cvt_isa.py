"""Build-time tripwire for the v_cvt_pk_bf16_f32 miscompare (profiles/r03_cvt_hazard.md): in the ISA of the product's kernels no instruction may read
the result of a v_cvt_pk_bf16_f32 in the next issue slot.  Every build in which one does loses candidates at random on the MI355X, hipcc has no
hazard rule for the opcode, and the GPU-side tripwire (tests/test_gpu_repeatability.py) needs a render to notice: this one fails in the CPU suite,
on the listing hipcc produces for gfx950 (device-only -S of csrc/rt3_device.hip, ~15 s), when a toolchain or a source change brings the pair back."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_reader_in_the_issue_slot_behind_a_bf16_conversion():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cvt_isa_check.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert int(last.split()[0]) >= 100 and " 0 with a reader in the next issue slot" in last, last      # the conversions are there, none is followed at once
