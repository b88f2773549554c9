#!/bin/bash
# round 5, GPU call 14: act()'s network branch through jh_value_act + device-mapped actions (value agents), conv1 forward at one tile per wave for 64-191 frames
# (lib tg6 vs tg5): the tests that touch them, the Rainbow A/B, then the driver's bench line of this tree
mkdir -p gpurun_out
cp ab/lib_tg6.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1500 python -m pytest tests/test_actors_gpu.py tests/test_rbnet_gpu.py tests/test_compat_gpu.py tests/test_agents_gpu.py tests/test_learning_curve_gpu.py -x -q > gpurun_out/r05_run14_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run14_tests.txt
tail -4 gpurun_out/r05_run14_tests.txt
bash tools/probes/ab_rb_lib.sh 2 tg5 tg6 2>&1 | tee gpurun_out/r05_run14_ab.txt
cp ab/lib_tg6.so jorldy_amd/csrc/libjorldy_hip.so
python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tg6', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4), {k.replace('jh_',''):v for k,v in d['lib_kernel_avg_us'].items() if 'tgemm' not in k})" | tee -a gpurun_out/r05_run14_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final2.json 2> gpurun_out/r05_bench_final2.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05_bench_final2.json") if l.startswith("{")][-1])
    print(json.dumps(d["legs"], indent=0))
    print("cpu", d["cpu_baseline"]["value"], "ms/step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_us"])
except Exception as e:
    print("parse failed", e)
PY
