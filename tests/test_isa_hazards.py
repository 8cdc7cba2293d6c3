"""Static check of the compiled gfx950 ISA for the hazards hipcc cannot see inside inline assembly
(scripts/check_dpp_hazard.py): no v_readlane / DPP read directly behind the VALU write of its source."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_no_readlane_or_dpp_hazard_in_compiled_kernels(tmp_path):
    src = os.path.join(ROOT, "ad_mpc_amd", "csrc", "admpc_kernels.hip")
    out = str(tmp_path / "admpc_kernels.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-DADMPC_WSYNC_FENCE_ONLY", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, capture_output=True, cwd=os.path.dirname(src))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_dpp_hazard
    counts = check_dpp_hazard.count_hazards(out)
    assert counts[2] == 0, "VALU write -> v_readlane without a wait state: %r" % (counts,)
    import check_inflight
    assert check_inflight.count(out) == 0, "a register is read while its un-waited ds_read (column-head assembly) is still in flight"
