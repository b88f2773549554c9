#!/bin/bash
# round 5, GPU call 8: multi-workgroup PPO loss (batched fetches, one LDS exchange) + dW1 column reduction (fetch order): bits old vs new, tests, Hopper A/B
mkdir -p gpurun_out
for v in old new; do cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so; timeout 300 python tools/probes/r05_loss_bits.py /tmp/bits_$v.npz 2>&1 | tail -2; done
python - <<'PY' 2>&1 | tee gpurun_out/r05_run8_bits.txt
import numpy as np
a, b = np.load("/tmp/bits_old.npz"), np.load("/tmp/bits_new.npz")
bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint8) if a[k].dtype != object else a[k], b[k].view(np.uint8) if b[k].dtype != object else b[k])]
print("arrays", len(a.files), "differing", len(bad), bad[:10])
for k in bad[:10]:
    print(k, np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
PY
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_baseline_width_gpu.py tests/test_kernels_gpu.py tests/test_dp_two_ranks_gpu.py -x -q > gpurun_out/r05_run8_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run8_tests.txt
tail -5 gpurun_out/r05_run8_tests.txt
for rep in 1 2; do for v in old new; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items()})
"; done; done 2>&1 | tee gpurun_out/r05_run8_ab_hopper.txt
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
