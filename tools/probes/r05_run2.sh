#!/bin/bash
# round 5, GPU call 2: the new tests (wide heads, compatibility table, bench contract with the new legs), then the full bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_compat_gpu.py tests/test_baseline_width_gpu.py tests/test_agents_gpu.py tests/test_kernels_gpu.py -x -q -k "compat or constructs or raises or container or ppo or more_than_8 or 32_workers or pponet" > gpurun_out/r05_run2_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run2_tests.txt
tail -15 gpurun_out/r05_run2_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_run2_bench.json 2> gpurun_out/r05_run2_bench.err; echo "bench rc $?"
tail -3 gpurun_out/r05_run2_bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05_run2_bench.json") if l.startswith("{")][-1])
    print(json.dumps(d["legs"], indent=0))
    print("dqn", json.dumps({k: d["dqn"].get(k) for k in ("value", "us_per_step", "exploration_phase_env_steps_per_s", "x_cpu_reference", "error")}))
    print("rainbow single_mode", json.dumps(d["rainbow"].get("single_mode")))
    print("variants", json.dumps(d.get("variants")))
    print("hopper e2e", json.dumps(d["hopper"].get("end_to_end")))
    print("cpu", d["cpu_baseline"]["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_us"])
except Exception as e:
    print("parse failed", e)
PY
