#!/bin/bash
# round 5, GPU call 10: lib tg2 (epilogue operands behind the MFMA loop in shared slots, conv1 forward / col2im / C51 / PER-sample / noise fetch order)
# against r8 (before the tgemm work) and tg (call 9): tests, then Rainbow, Ape-X, Hopper
mkdir -p gpurun_out
cp ab/lib_tg2.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1500 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_agents_gpu.py tests/test_kernels_gpu.py tests/test_capture_gpu.py tests/test_baseline_width_gpu.py -x -q > gpurun_out/r05_run10_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run10_tests.txt
tail -5 gpurun_out/r05_run10_tests.txt
{
bash tools/probes/ab_rb_lib.sh 2 r8 tg tg2
bash tools/probes/ab_apex_lib.sh 2 r8 tg tg2
for rep in 1 2; do for v in r8 tg2; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items()})
"; done; done
} 2>&1 | tee gpurun_out/r05_run10_ab.txt
cp ab/lib_tg2.so jorldy_amd/csrc/libjorldy_hip.so
