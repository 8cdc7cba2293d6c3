#!/usr/bin/env python3
"""Static check of a hipcc -S listing for the hand-written assembly that issues LDS reads WITHOUT waiting for them (the
factorisation's column head: the reads are meant to land later, behind a counted s_waitcnt lgkmcnt(N)).  hipcc does not know
those registers are in flight; a register copy it inserts, or a miscounted N, would silently use stale data.
Model: the LDS operations of a wave complete in order, so after s_waitcnt lgkmcnt(N) only the N youngest can be in flight;
flag every instruction that reads a VGPR whose ds_read is still in that set.   Usage: check_inflight.py file.s"""
import re, sys


def vregs(tok):
    tok = tok.strip().lstrip('-|').rstrip('|')
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def count(path, verbose=False):
    queue = []          # one entry per LDS instruction in issue order: destination VGPRs (empty set for stores)
    bad = 0
    for line in open(path):
        t = line.split(';')[0].strip()
        if t.endswith(':') and t.startswith('.LBB'): queue = []     # a basic-block entry: the text above is not (only) its predecessor.  The un-waited
        if not t or t[0] == '.' or t.endswith(':'): continue        # reads this check is about live in straight-line, fully unrolled assembly
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(',')]
        if op.startswith('s_waitcnt'):
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1)); queue = queue[len(queue) - n:] if n else []
            continue
        if op.startswith(('s_endpgm', 's_barrier', 's_branch', 's_setpc')): queue = []; continue
        is_store = op.startswith(('ds_write', 'global_store', 'scratch_store', 'buffer_store'))
        srcs = set()
        for a in (args if is_store else args[1:]):
            srcs |= vregs(a.split()[0] if a else a)
        flight = set().union(*queue) if queue else set()
        if srcs & flight:
            bad += 1
            if verbose and bad <= 10: print("IN-FLIGHT READ of v%s: %s" % (sorted(srcs & flight), t))
        if op.startswith('ds_'):
            queue.append(vregs(args[0]) if op.startswith('ds_read') and args else set())
        elif args and op.startswith(('v_', 'scratch_load', 'global_load', 'buffer_load')):
            w = vregs(args[0].split()[0])
            queue = [q - w for q in queue]           # overwritten by the VALU (or by a spill reload / memory load, which hipcc orders behind
                                                     # the LDS read with its own s_waitcnt): not the loaded value any more
    return bad


if __name__ == "__main__":
    b = count(sys.argv[1], verbose=True)
    print("in-flight reads: %d" % b)
    sys.exit(1 if b else 0)
