#!/bin/bash
# round 5, GPU call 5: interleaved ds_read_b64 for x-contiguous GEMM operands -- tgemm tests, then old vs new on the Ape-X learner (B = 512),
# the Hopper learner (minibatch 2048) and the Rainbow learner (B = 32)
mkdir -p gpurun_out
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
timeout 600 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_baseline_width_gpu.py -x -q > gpurun_out/r05_run5_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run5_tests.txt
tail -4 gpurun_out/r05_run5_tests.txt
timeout 300 tools/probes/ab_apex_lib.sh 2 old new 2>&1 | tee gpurun_out/r05_run5_ab_apex.txt
for rep in 1 2; do for v in old new; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_tgemm_',''):(v['avg_us'], v.get('TFLOP/s')) for n,v in k.items() if 'tgemm' in n})
"; done; done 2>&1 | tee gpurun_out/r05_run5_ab_hopper.txt
timeout 200 tools/probes/ab_rb_lib.sh 2 old new 2>&1 | tee gpurun_out/r05_run5_ab_rb.txt
