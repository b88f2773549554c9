#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_agents_gpu.py tests/test_kernels_gpu.py tests/test_baseline_width_gpu.py tests/test_learning_curve_gpu.py -x -q > gpurun_out/r35.log 2>&1; echo "rc=$?" >> gpurun_out/r35.log
tail -4 gpurun_out/r35.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-rainbow --no-apex --no-dqn --no-ppo-atari --no-variants > gpurun_out/r35_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r35_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["legs"]["hopper_transitions_s"], d["legs"]["hopper_end_to_end_env_transitions_s"], d["legs"]["hopper_per_gpu_share_of_8_env_transitions_s"])
PY
