#!/usr/bin/env python3
"""Static hazard check of a hipcc -S listing (the assembler and hipcc's hazard recogniser cannot see inside inline assembly).
Rules (wait state = one issued VALU or SALU instruction, s_nop N = N + 1; LDS / memory instructions count 0 -- measured):
  R1  VALU writes VGPR  -> DPP source (src0 of v_*_dpp) reads it          : 2 wait states   (informational: hipcc itself emits
      such pairs back to back on gfx950, so this is not a hazard there)
  R2  VALU writes VGPR  -> v_readlane / v_readfirstlane reads it          : 1 wait state    (REAL: measured with
      scripts/probes/subst_probe.hip -- wrong, non-repeatable results; the exit code reflects this rule only.
      v_writelane lane L -> v_readlane lane L back to back returns the OLD value 99 % of the time, a different lane is
      safe: scripts/probes/wlane_probe.hip; hipcc's SGPR spill code uses exactly these two instructions)
  R3  VALU writes SGPR (v_readlane, v_cmp) -> VALU reads that SGPR        : 2 wait states   (informational: hipcc emits them)
Usage: check_dpp_hazard.py file.s"""
import re, sys

def vregs(tok):
    tok = tok.strip().lstrip('-|').rstrip('|')
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

def sregs(tok):
    tok = tok.strip().lstrip('-|').rstrip('|')
    m = re.match(r's\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r's(\d+)$', tok)
    if m: return {int(m.group(1))}
    return {'vcc'} if tok.startswith('vcc') else set()

def count_hazards(path, verbose=False):
    hist = []          # (kind, vgpr_written, sgpr_written, waitstates, text)
    bad = {1: 0, 2: 0, 3: 0}; n = 0
    for line in open(path):
        t = line.split(';')[0].strip()
        if not t or t[0] == '.' or t.endswith(':') or t.startswith('//'): continue
        op = t.split()[0]
        rest = t[len(op):].strip()
        args = [a.strip() for a in rest.split(',')] if rest else []
        if args: args[-1] = args[-1].split()[0] if args[-1] else args[-1]
        is_valu = op.startswith('v_')
        def lookback(limit, pred, rule):
            ws = 0
            for kind, vw, sw, w, txt in reversed(hist[-12:]):
                if ws >= limit: return
                if pred(kind, vw, sw):
                    bad[rule] += 1
                    if verbose and bad[rule] <= 8: print("R%d: %-60s -> %s" % (rule, txt, t))
                    return
                ws += w
        if is_valu:
            n += 1
            if '_dpp' in op and len(args) >= 2:
                src = vregs(args[1]); lookback(2, lambda k, vw, sw: k == 'valu' and vw & src, 1)
            if op.startswith(('v_readlane', 'v_readfirstlane')) and len(args) >= 2:
                src = vregs(args[1])
                rl = args[2] if op.startswith('v_readlane') and len(args) >= 3 else None
                def writes_what_is_read(k, vw, sw, txt):
                    if not (k == 'valu' and vw & src): return False
                    if txt.startswith('v_writelane') and rl is not None and rl.isdigit():
                        wl = txt.split(',')[-1].split()[0].strip()
                        if wl.isdigit() and wl != rl: return False       # another lane: measured safe (probes/wlane_probe.hip)
                    return True
                ws = 0
                for kind, vw, sw, w, txt in reversed(hist[-12:]):
                    if ws >= 1: break
                    if writes_what_is_read(kind, vw, sw, txt):
                        bad[2] += 1
                        if verbose and bad[2] <= 8: print("R2: %-60s -> %s" % (txt, t))
                        break
                    ws += w
            ss = set()
            for a in args[1:]: ss |= sregs(a)
            if op.startswith('v_cndmask') and len(args) == 3: ss |= {'vcc'}
            if ss: lookback(2, lambda k, vw, sw: k == 'valu' and sw & ss, 3)
        if op.startswith('s_nop'):
            hist.append(('salu', set(), set(), int(args[0]) + 1, t))
        elif is_valu:
            vw, sw = set(), set()
            if op.startswith(('v_readlane', 'v_readfirstlane')): sw = sregs(args[0])
            elif op.startswith('v_cmp'): sw = sregs(args[0]) if args and (args[0].startswith('s') or args[0].startswith('vcc')) else {'vcc'}
            elif 'swap' in op: vw = vregs(args[0]) | vregs(args[1])
            else: vw = vregs(args[0]) if args else set()
            hist.append(('valu', vw, sw, 1, t))
        elif op.startswith('s_'):
            hist.append(('salu', set(), set(), 1, t))
        else:
            hist.append(('mem', set(), set(), 0, t))
    if verbose:
        print("checked %d VALU instructions: R1 (VALU->DPP) %d, R2 (VALU->readlane) %d, R3 (VALU sgpr->VALU) %d" % (n, bad[1], bad[2], bad[3]))
    return bad


def main(path):
    bad = count_hazards(path, verbose=True)
    return 1 if bad[2] else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
