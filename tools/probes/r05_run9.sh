#!/bin/bash
# round 5, GPU call 9: tgemm prologue / fetch / epilogue order (lib tg) against the previous build (lib r8): tests, then Rainbow, Ape-X and Hopper A/B;
# plus the bit comparison of call 8 (old = call 7's library, r8 = call 8's) that call 8's script tripped over
mkdir -p gpurun_out
for v in old r8 tg; do cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so; timeout 300 python tools/probes/r05_loss_bits.py /tmp/bits_$v.npz 2>&1 | tail -1; done
python - <<'PY' 2>&1 | tee gpurun_out/r05_run9_bits.txt
import numpy as np
a = np.load("/tmp/bits_old.npz")
for other in ("r8", "tg"):
    b = np.load(f"/tmp/bits_{other}.npz")
    bad = [k for k in a.files if np.ascontiguousarray(a[k]).tobytes() != np.ascontiguousarray(b[k]).tobytes()]
    print("old vs", other, ": arrays", len(a.files), "differing", len(bad), bad[:10])
    for k in bad[:10]:
        print("  ", k, np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
PY
cp ab/lib_tg.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1200 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_agents_gpu.py tests/test_baseline_width_gpu.py tests/test_kernels_gpu.py tests/test_capture_gpu.py -x -q > gpurun_out/r05_run9_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_run9_tests.txt
tail -5 gpurun_out/r05_run9_tests.txt
{
bash tools/probes/ab_rb_lib.sh 2 r8 tg
bash tools/probes/ab_apex_lib.sh 2 r8 tg
for rep in 1 2; do for v in r8 tg; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
timeout 120 python tools/bench_hopper.py --iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['lib_kernels']
print('$v hopper', 'ms_per_iter', round(d['ms_per_iteration'],2), round(d['learner_transitions_per_s']), {n.replace('jh_',''):v['avg_us'] for n,v in k.items()})
"; done; done
} 2>&1 | tee gpurun_out/r05_run9_ab.txt
cp ab/lib_tg.so jorldy_amd/csrc/libjorldy_hip.so
