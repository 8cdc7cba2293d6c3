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


def test_checkers_flag_synthetic_hazards(tmp_path):
    """Positive controls: the two static checkers do flag what they are for (and the in-flight check ends at a basic-block boundary)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_dpp_hazard
    import check_inflight
    a = tmp_path / "a.s"
    a.write_text("\tv_fma_f64 v[2:3], v[4:5], v[6:7], v[8:9]\n\tv_readlane_b32 s4, v2, 3\n")
    assert check_dpp_hazard.count_hazards(str(a))[2] == 1
    a.write_text("\tv_fma_f64 v[2:3], v[4:5], v[6:7], v[8:9]\n\ts_nop 0\n\tv_readlane_b32 s4, v2, 3\n")
    assert check_dpp_hazard.count_hazards(str(a))[2] == 0
    b = tmp_path / "b.s"
    b.write_text("\tds_read_b64 v[4:5], v1\n\tds_read_b64 v[6:7], v1 offset:8\n\ts_waitcnt lgkmcnt(1)\n\tv_add_f64 v[8:9], v[4:5], v[6:7]\n")
    assert check_inflight.count(str(b)) == 1                       # v[6:7] is still in flight behind lgkmcnt(1)
    b.write_text("\tds_read_b64 v[4:5], v1\n\tds_read_b64 v[6:7], v1 offset:8\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f64 v[8:9], v[4:5], v[6:7]\n")
    assert check_inflight.count(str(b)) == 0
    b.write_text("\tds_read_b64 v[6:7], v1\n\ts_branch .LBB0_2\n.LBB0_1:\n\tv_add_f64 v[8:9], v[4:5], v[6:7]\n")
    assert check_inflight.count(str(b)) == 0                       # another block: not this read's successor in the text
