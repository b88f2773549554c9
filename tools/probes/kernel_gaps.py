#!/usr/bin/env python3
"""Gaps between consecutive kernel dispatches of a rocprofv3 --kernel-trace CSV (steady state = the last `tail` dispatches): how long the queue sits between the
end of one kernel and the start of the next, by the name of the kernel that FOLLOWS the gap."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-tail:]
gaps = collections.defaultdict(list)
tot_k = tot_g = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps[b["Kernel_Name"][:70]].append(g)
    tot_g += max(g, 0)
    tot_k += int(b["End_Timestamp"]) - int(b["Start_Timestamp"])
print(f"dispatches {len(rows)}  kernel time {tot_k / 1e3:.1f} us  gaps {tot_g / 1e3:.1f} us  = {tot_g / max(1, len(rows) - 1) / 1e3:.2f} us per dispatch")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:24]:
    v = sorted(v)
    print(f"{k:70s} n={len(v):5d} median {v[len(v) // 2] / 1e3:6.2f} us  p90 {v[int(len(v) * 0.9)] / 1e3:6.2f}  mean {sum(v) / len(v) / 1e3:6.2f}")
