"""A 64-lane interpreter for the small gfx950 subset csrc/gen_subst_asm.py emits (TEST INFRASTRUCTURE): EXEC masks by s_bfm_b64,
ds_read_b64 / ds_write_b64 with immediate offsets, v_fmac_f64_dpp row_newbcast, v_mov_b64, v_add_f64.  It checks WHAT the generated
blocks compute (indices, masks, offsets) on the CPU; timing hazards are checked on the compiled ISA by tests/test_isa_hazards.py."""
import re

import numpy as np

WAVE = 64


class Wave:
    def __init__(self):
        self.v = {}            # register number -> np.array(64) (float64 for pairs keyed by their low register, int for 32-bit)
        self.lds = {}          # byte address -> float
        self.exec = np.ones(WAVE, dtype=bool)
        self.saved = None

    def run(self, lines, operands=None):
        for raw in lines:
            ln = raw.strip()
            if operands is not None:
                ln = re.sub(r"%(\d+)", lambda m: operands[int(m.group(1))], ln)
            if not ln or ln.startswith("s_waitcnt") or ln.startswith("s_nop"):
                continue
            m = re.match(r"s_mov_b64 s\[46:47\], exec", ln)
            if m:
                self.saved = self.exec.copy(); continue
            if ln == "s_mov_b64 exec, s[46:47]":
                self.exec = self.saved.copy(); continue
            if ln == "s_mov_b64 exec, -1":
                self.exec[:] = True; continue
            m = re.match(r"s_bfm_b64 exec, (\d+), (\d+)", ln)
            if m:
                cnt, off = int(m.group(1)), int(m.group(2))
                self.exec = np.array([off <= l < off + cnt for l in range(WAVE)]); continue
            m = re.match(r"ds_read_b64 v\[(\d+):\d+\], v(\d+) offset:(\d+)", ln)
            if m:
                d, a, off = int(m.group(1)), int(m.group(2)), int(m.group(3))
                reg = self.v.setdefault(d, np.zeros(WAVE))
                for l in np.nonzero(self.exec)[0]:
                    reg[l] = self.lds.get(int(self.v[a][l]) + off, np.nan)
                continue
            m = re.match(r"ds_write_b64 v(\d+), v\[(\d+):\d+\] offset:(\d+)", ln)
            if m:
                a, s, off = int(m.group(1)), int(m.group(2)), int(m.group(3))
                for l in np.nonzero(self.exec)[0]:
                    self.lds[int(self.v[a][l]) + off] = float(self.v[s][l])
                continue
            m = re.match(r"v_fmac_f64_dpp v\[(\d+):\d+\], v\[(\d+):\d+\], (-?)v\[(\d+):\d+\] row_newbcast:(\d+)", ln)
            if m:
                d, s0, neg, s1, bc = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4)), int(m.group(5))
                src = self.v[s0].copy()
                for l in np.nonzero(self.exec)[0]:
                    sl = (l & ~15) + bc
                    assert self.exec[sl], "DPP source lane %d disabled (target %d): %s" % (sl, l, ln)
                    b = -self.v[s1][l] if neg else self.v[s1][l]
                    self.v[d][l] = self.v[d][l] + src[sl] * b
                continue
            m = re.match(r"v_mov_b64 v\[(\d+):\d+\], 0", ln)
            if m:
                reg = self.v.setdefault(int(m.group(1)), np.zeros(WAVE)); reg[self.exec] = 0.0; continue
            m = re.match(r"v_add_f64 v\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", ln)
            if m:
                d, a, b = int(m.group(1)), int(m.group(2)), int(m.group(3))
                reg = self.v.setdefault(d, np.zeros(WAVE))
                reg[self.exec] = self.v[a][self.exec] + self.v[b][self.exec]; continue
            raise ValueError("asm_emu: unknown instruction: " + ln)
