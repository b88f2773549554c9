#!/bin/bash
# round 5, final validation of the committed tree (the in-tree library is built from it): the whole GPU suite with the margins ledger, smoke(), the driver's
# bench line, rocprofv3 kernel statistics of the same command, and kernel statistics + PMC of the Rainbow and Ape-X learners
mkdir -p gpurun_out
JH_MARGINS_OUT=gpurun_out/r05_margins.json timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r05_final_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_final_tests.txt
tail -4 gpurun_out/r05_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05_bench_final.json") if l.startswith("{")][-1])
    print(json.dumps(d["legs"], indent=0))
    print("cpu", d["cpu_baseline"]["value"], "ms/step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_us"])
except Exception as e:
    print("parse failed", e)
PY




