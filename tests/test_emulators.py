"""CPU: the lane-level numpy restatements of the kernels' ADDRESSING (tools/*_emulate.py: LDS images and swizzles, direct-to-LDS lane ->
source maps, fragment reads, accumulator ownership, partial-sum exchanges, output addresses -- every formula the kernel's, compared with
torch's conv3d / conv_transpose3d) still reproduce the reference operators.  They are what a kernel's index arithmetic is debugged with
before a GPU is involved; running them here keeps them in step with the kernels' geometry rules (e.g. the macro-tile kernel's column
bands of round 6).  Reference ops: nn.Conv3d / nn.ConvTranspose3d as Generic_UNet uses them (test_MultiHead_Module.py:346-426)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool", ["mt_emulate.py", "v9_emulate.py", "down2s_emulate.py", "gen_emulate.py"])
def test_kernel_addressing_emulators_reproduce_the_reference_ops(tool):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_macro_tile_geometry_keeps_full_width_bands_on_the_baseline_plan():
    """mt_geometry (csrc/igemm_conv_mt.hip, restated in tools/mt_emulate.py): every level of the 160x192x160 plan keeps bands of ALL
    columns (the measured configuration); wide planes of anisotropic plans are cut into column bands that fill the MFMA columns."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import mt_emulate as m
    finally:
        sys.path.pop(0)
    assert m.geometry(24, 20) == (1.0, 5, 8, 20) and m.geometry(48, 40) == (1.0, 5, 4, 40)
    assert m.geometry(12, 10)[1:] == (4, 12, 10) and m.geometry(6, 5)[1:] == (4, 6, 5)
    for hw in ((80, 64), (160, 128), (40, 32)):
        e, wn, ty, tx = m.geometry(*hw)
        assert e == 1.0 and tx == 32 and 4 * (ty + 2) * (tx + 2) <= 1024          # (column bands wherever the plane is wider)
