#!/usr/bin/env python3
"""Scan a hipcc -S listing for the pattern tools/isa_chain.py showed in the PPO minibatch kernels: an inner loop (or straight-line
run) in which global loads are waited for one at a time (`global_load ... s_waitcnt vmcnt(0)` with <= 2 loads in flight), i.e. a
chain of dependent L2 round trips the source does not show.  Prints kernel, block label, loads in the block, waits to zero.

  python tools/isa_serial_loops.py /tmp/isa/*.s
"""
import re
import sys


def scan(path):
    kern, blk, loads, zero_waits, inflight, header = None, None, 0, 0, 0, ""
    out = []

    def flush():
        if kern and blk and loads >= 1 and zero_waits >= 1 and "Loop" in header and loads <= 2 * zero_waits:
            out.append((kern, blk, loads, zero_waits))

    for l in open(path):
        s = l.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m:
            flush()
            kern, blk, loads, zero_waits, header = m.group(1), None, 0, 0, ""
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", s)
        if m:
            flush()
            blk, header, loads, zero_waits = m.group(1), m.group(2), 0, 0
            continue
        if s.startswith("global_load") or s.startswith("buffer_load"):
            loads += 1
        elif s.startswith("s_waitcnt") and "vmcnt(0)" in s:
            zero_waits += 1
    flush()
    return out


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for k, b, n, z in scan(p):
            print(f"{p.split('/')[-1]:14s} {k[:70]:70s} {b:12s} loads {n:2d}  waits-to-zero {z}")
