#!/bin/bash
# round 6, GPU call 8: the engine after the prologue / epilogue / split work: its tests, then r5 vs new on the Ape-X, Hopper and Rainbow learners
cp ab/lib_fix2.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1500 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_baseline_width_gpu.py -x -q > gpurun_out/r06_run8_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_run8_tests.txt
tail -3 gpurun_out/r06_run8_tests.txt
for rep in 1 2; do for v in r5 fix2; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v apex', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_tgemm_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"
python tools/bench_rainbow.py 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v rainbow', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('learner_updates_per_s','ms_per_learn_only','updates_per_s','ms_per_update')})
"
done; done 2>&1 | tee gpurun_out/r06_run8_ab.txt
cp ab/lib_fix2.so jorldy_amd/csrc/libjorldy_hip.so
