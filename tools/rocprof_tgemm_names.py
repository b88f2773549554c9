#!/usr/bin/env python3
"""Put the call-site names back on the grouped GEMM engine's kernels in a rocprofv3 --kernel-trace --stats summary.

The engine's kernels are instantiated once per call site (csrc/jh_tgemm.h: JH_TGEMM_TAGS), so rocprofv3 sees
`jh_tgemm_kernel<TM, TN, ID, EPI>` (register-staged operands) and `jh_tgemm_dma_kernel<TM, TN, ID, EPI, NB>` (LDS-DMA operands; `<ID, EPI, NB>` in the summaries of rounds 3-5); this prints the
CSV with `jh_tgemm_<site>[TMxTN|dma]` in their place:  python tools/rocprof_tgemm_names.py profiles/r03_bench_kernel_stats.csv
"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tags():
    src = open(os.path.join(ROOT, "jorldy_amd", "csrc", "jh_tgemm.h")).read()
    block = src[src.index("#define JH_TGEMM_TAGS(X)"):]
    block = block[: block.index("\n//")]
    return {int(i): n for n, i in re.findall(r"X\((\w+), (\d+)\)", block)}


def rename(name, t=None):
    t = t or tags()
    m = re.search(r"jh_tgemm_kernel<(\d+), ?(\d+), ?(\d+)[,>]", name)
    if m:
        return f"jh_tgemm_{t.get(int(m.group(3)), m.group(3))}[{32 * int(m.group(1))}x{32 * int(m.group(2))}]"
    m = re.search(r"jh_tgemm_dma_kernel<(\d+), ?(\d+), ?(\d+), ?(?:true|false)", name)  # round 6: <TM, TN, TAG, EPI, NB>
    if m:
        return f"jh_tgemm_{t.get(int(m.group(3)), m.group(3))}[dma {32 * int(m.group(1))}x{32 * int(m.group(2))}]"
    m = re.search(r"jh_tgemm_dma_kernel<(\d+)[,>]", name)  # rounds 3-5: <TAG, EPI, NB>
    if m:
        return f"jh_tgemm_{t.get(int(m.group(1)), m.group(1))}[dma]"
    return name


if __name__ == "__main__":
    t = tags()
    w = csv.writer(sys.stdout)
    with open(sys.argv[1]) as f:
        for i, row in enumerate(csv.reader(f)):
            if i and row:
                row[0] = rename(row[0], t)
            w.writerow(row)
