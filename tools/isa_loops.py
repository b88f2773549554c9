#!/usr/bin/env python3
"""The MFMA loops of one kernel in a `hipcc -S` listing, compacted: runs of MFMAs collapsed to counts, every other instruction by
mnemonic (waits spelled out) -- to check that a hand-pipelined loop kept its LDS reads / DMA issues in the MFMA shadows.
    python tools/isa_loops.py k.s <mangled-name-substring> [min_mfma_per_block]"""
import re
import sys

txt = open(sys.argv[1]).read()
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):.*$", txt, re.M)
start = m.start()
end = txt.index(".end_amdhsa_kernel", start)
blocks, cur, lab = [], [], m.group(1)
for l in txt[start:end].split("\n")[1:]:
    l = l.strip()
    if not l or l.startswith(";"):
        continue
    if re.match(r"^[.\w$]+:$", l):
        blocks.append((lab, cur))
        lab, cur = l, []
        continue
    if l.startswith("."):
        continue
    cur.append(l)
blocks.append((lab, cur))
for lab, cur in blocks:
    nm = sum(1 for o in cur if o.startswith("v_mfma"))
    if nm < min_mfma:
        continue
    out, run = [], 0
    for o in cur:
        if o.startswith("v_mfma"):
            run += 1
            continue
        if run:
            out.append(f"[{run} mfma]")
            run = 0
        op = o.split()[0]
        out.append(o.replace(" ", "") if op.startswith("s_waitcnt") else op)
    if run:
        out.append(f"[{run} mfma]")
    print(lab, "instructions", len(cur), "mfma", nm)
    print("  " + " ".join(out))
