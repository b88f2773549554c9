#!/usr/bin/env python3
"""Fold the margin ledgers of several GPU-suite runs (tests/margins.py -> gpurun_out/margins_*.json, one per box) into one summary:
per test the worst achieved / allowed over the boxes, the spread between boxes, and the list of tests above 0.5.

    python tools/margins_merge.py gpurun_out/r04_margins_box*.json > profiles/r04_margins.json
"""
import json
import sys

runs = [json.load(open(p)) for p in sys.argv[1:]]
tests = {}
for i, r in enumerate(runs):
    for k, v in r["tests"].items():
        e = tests.setdefault(k, {"ratios": [None] * len(runs), "what": v["what"]})
        e["ratios"][i] = v["worst_ratio"]
        if v["worst_ratio"] >= max(x for x in e["ratios"] if x is not None):
            e["what"] = v["what"]
out = {"boxes": [{"file": p, "host": r.get("host"), "gpu": r.get("gpu"), "n_tests": r["n_tests"], "max_ratio": r["max_ratio"]} for p, r in zip(sys.argv[1:], runs)],
       "n_tests_with_tolerance_asserts": len(tests)}
rows = []
for k, e in tests.items():
    rs = [x for x in e["ratios"] if x is not None]
    rows.append({"test": k, "worst_ratio": max(rs), "min_ratio": min(rs), "same_on_every_box": len(set(rs)) == 1 and len(rs) == len(runs), "ratios": e["ratios"], "what": e["what"]})
rows.sort(key=lambda r: -r["worst_ratio"])
out["max_ratio"] = rows[0]["worst_ratio"] if rows else None
out["over_half"] = [r for r in rows if r["worst_ratio"] > 0.5]
out["over_half_and_box_dependent"] = [r["test"] for r in rows if r["worst_ratio"] > 0.5 and not r["same_on_every_box"]]
out["tests"] = rows
json.dump(out, sys.stdout, indent=1)
