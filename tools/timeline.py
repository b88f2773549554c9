#!/usr/bin/env python3
"""Per-iteration GPU timeline of the PPO bench from a rocprofv3 --kernel-trace CSV (kernel start / end timestamps):
where the step's wall time goes -- the acting kernel, the learner's kernels, and the idle gaps between them.

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python bench.py --steps 30 --warmup 10 --no-rainbow --no-roofline --no-cpu-baseline
  python tools/timeline.py <dir> [out.json]
"""
import csv
import glob
import json
import sys
from collections import OrderedDict, defaultdict


def short(n):
    n = n.replace("void ", "")
    for cut in ("(", "<"):
        if cut in n:
            n = n[: n.index(cut)]
    return n


def main():
    d = sys.argv[1]
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # iterations are delimited by the persistent acting kernel
    starts = [i for i, r in enumerate(rows) if r[2] == "jh_act_persist_kernel"]
    its = []
    for a, b in zip(starts[:-1], starts[1:]):
        its.append(rows[a:b + 1])  # includes the NEXT acting kernel's start as the end marker
    its = its[len(its) // 3:]  # steady state
    agg = defaultdict(lambda: [0, 0.0])
    gap_after = defaultdict(lambda: [0, 0.0])
    tot = {"iter_us": 0.0, "acting_us": 0.0, "learner_kernels_us": 0.0, "gaps_us": 0.0, "gap_act_to_first_us": 0.0, "gap_last_to_act_us": 0.0}
    for it in its:
        t0, t_end = it[0][0], it[-1][0]
        tot["iter_us"] += (t_end - t0) / 1e3
        tot["acting_us"] += (it[0][1] - it[0][0]) / 1e3
        body = it[1:-1]
        prev_end, prev_name = it[0][1], "jh_act_persist_kernel"
        for s, e, n in body:
            agg[n][0] += 1
            agg[n][1] += (e - s) / 1e3
            g = max(0.0, (s - prev_end) / 1e3)
            gap_after[prev_name][0] += 1
            gap_after[prev_name][1] += g
            tot["gaps_us"] += g
            if prev_name == "jh_act_persist_kernel":
                tot["gap_act_to_first_us"] += g
            tot["learner_kernels_us"] += (e - s) / 1e3
            prev_end, prev_name = max(prev_end, e), n
        g = max(0.0, (t_end - prev_end) / 1e3)
        tot["gap_last_to_act_us"] += g
        tot["gaps_us"] += g
    n = max(1, len(its))
    out = OrderedDict(iterations=len(its))
    out.update({k: round(v / n, 2) for k, v in tot.items()})
    out["kernels_per_iter"] = OrderedDict((k, {"launches": round(c / n, 2), "avg_us": round(t / c, 2), "us_per_iter": round(t / n, 1)})
                                          for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]))
    out["avg_gap_after_us"] = OrderedDict((k, round(t / c, 2)) for k, (c, t) in sorted(gap_after.items(), key=lambda kv: -kv[1][1]))
    js = json.dumps(out, indent=1)
    print(js)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js)


if __name__ == "__main__":
    main()
