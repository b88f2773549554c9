#!/usr/bin/env python3
"""Latency-chain view of one kernel in a hipcc -S listing: only the instructions that start or
end a memory round trip (global / LDS loads and stores, s_waitcnt, barriers), with runs of
MFMA / VALU instructions collapsed to counts.  A GPU-less proxy for "how many dependent L2
round trips sit between launch and the last store" (DESIGN: the minibatch kernels are latency chains).

  hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o k.s file.hip
  python tools/isa_chain.py k.s <mangled-name-substring>
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    run = {}
    def flush():
        if run:
            print("      ... " + ", ".join(f"{v} x {k}" for k, v in run.items()))
            run.clear()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"\.LBB\d+_\d+:", s):
                flush()
                print(s)
            continue
        op = s.split()[0]
        if re.match(r"(global_|buffer_|flat_|ds_|s_waitcnt|s_barrier|s_load|s_endpgm|s_cbranch|s_branch|scratch_)", op):
            flush()
            print("  " + s.split(";")[0].strip())
        else:
            k = "mfma" if "mfma" in op else ("valu" if op.startswith("v_") else "salu")
            run[k] = run.get(k, 0) + 1
    flush()


if __name__ == "__main__":
    main()
