#!/bin/bash
# round 6, validation of the committed tree (the in-tree library is built from it): the whole GPU suite with the margins ledger, smoke(), the driver's bench line
mkdir -p gpurun_out
JH_MARGINS_OUT=gpurun_out/r06_margins.json timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_final_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_final_tests.txt
tail -4 gpurun_out/r06_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r06_bench_final.json") if l.startswith("{")][-1])
    print(json.dumps(d["legs"], indent=0))
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("torch_threads"), d["cpu_baseline"].get("learn_ms_by_torch_threads"), "ms/step", d["ms_per_step"], "value", d["value"])
    r = d["roofline"]
    print("roofline", r["kernel"], r["frac"], r["avg_us"], r.get("rocprof_avg_us"), r.get("traffic"), r.get("algorithmic_bytes"), r.get("traffic_ratio"))
    for leg in ("rainbow", "hopper", "apex"):
        rr = d.get(leg, {}).get("roofline") or {}
        print(leg, rr.get("kernel"), rr.get("frac"), rr.get("avg_us"), rr.get("rocprof_avg_us"), rr.get("rocprof_summary"))
except Exception as e:
    print("parse failed", e)
PY
