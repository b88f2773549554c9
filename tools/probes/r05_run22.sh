#!/bin/bash
# round 5, GPU call 22: split-K arrival counters one per 128-byte line (lib tk2) against tk (two-level optimizer ticket only) and pw3: tgemm tests, Rainbow / Ape-X / Hopper
mkdir -p gpurun_out
cp ab/lib_tk2.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_baseline_width_gpu.py -x -q > gpurun_out/r05_run22_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run22_tests.txt
tail -3 gpurun_out/r05_run22_tests.txt
{
bash tools/probes/ab_rb_lib.sh 2 pw3 tk tk2
bash tools/probes/ab_apex_lib.sh 2 tk tk2
for rep in 1 2; do for v in tk tk2; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"; done; done
} 2>&1 | tee gpurun_out/r05_run22_ab.txt
cp ab/lib_tk2.so jorldy_amd/csrc/libjorldy_hip.so
