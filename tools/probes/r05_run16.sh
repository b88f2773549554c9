#!/bin/bash
# round 5, GPU call 16: the LDS-DMA GEMM loop with its MFMA operands double-buffered in registers (lib tg7) against tg6: tests, then Ape-X / Hopper / Rainbow
mkdir -p gpurun_out
cp ab/lib_tg7.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_baseline_width_gpu.py -x -q > gpurun_out/r05_run16_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run16_tests.txt
tail -4 gpurun_out/r05_run16_tests.txt
{
bash tools/probes/ab_apex_lib.sh 2 tg6 tg7
for rep in 1 2; do for v in tg6 tg7; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"; done; done
bash tools/probes/ab_rb_lib.sh 2 tg6 tg7
} 2>&1 | tee gpurun_out/r05_run16_ab.txt
cp ab/lib_tg7.so jorldy_amd/csrc/libjorldy_hip.so
