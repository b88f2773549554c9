#!/bin/bash
# round 6, GPU call 11: operand B through registers (JH_TGEMM_BREG=1) against the all-DMA loop: engine tests, then the learners
JH_TGEMM_BREG=1 timeout 1200 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
JH_TGEMM_BREG=$v python tools/bench_apex.py --updates 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('breg=$v apex', 'learn_ms', round(d['ms_per_learn_only'],4), {n.replace('jh_tgemm_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"
JH_TGEMM_BREG=$v timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('breg=$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items() if 'tgemm' in n})
"
JH_TGEMM_BREG=$v python tools/bench_rainbow.py 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('breg=$v rainbow', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('learner_updates_per_s','ms_per_learn_only')})
"
done; done 2>&1 | tee gpurun_out/r06_run11_breg.txt
