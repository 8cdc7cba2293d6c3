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
    csrc = os.path.join(ROOT, "ad_mpc_amd", "csrc")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_dpp_hazard
    import check_inflight
    # the translation units exactly as csrc/Makefile builds them (the two persistent condensed kernels with their per-unit flag)
    for name, extra in (("admpc_kernels", []), ("admpc_rowqp", []), ("admpc_quad", []),
                        ("admpc_fused20", ["-mllvm", "-disable-machine-licm"]), ("admpc_seg", ["-mllvm", "-disable-machine-licm"])):
        out = str(tmp_path / (name + ".s"))
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-DADMPC_WSYNC_FENCE_ONLY", "-std=c++17", "-O3", "-S", "--cuda-device-only",
                        "-Wno-bitwise-instead-of-logical"] + extra + ["-o", out, os.path.join(csrc, name + ".hip")], check=True, capture_output=True, cwd=csrc)
        counts = check_dpp_hazard.count_hazards(out)
        assert counts[2] == 0, "%s: VALU write -> v_readlane of the same register / lane without a wait state: %r" % (name, counts)
        assert check_inflight.count(out) == 0, "%s: a register is read while its un-waited ds_read (column-head assembly) is still in flight" % name
