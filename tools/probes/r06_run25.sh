#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r25_bench_default.json 2> gpurun_out/r25_bench_default.err; echo "bench rc $?"
t1=$(date +%s); echo "default bench wall seconds: $((t1-t0))"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r25_bench_default.json") if l.startswith("{")][-1])
print(d["steps"], d["warmup"], d["value"], d["ms_per_step"], json.dumps(d["legs"])[:600])
PY
timeout 300 python -m pytest tests/test_ppo_cnn_gpu.py -x -q -k act 2>&1 | tail -2
