#!/bin/bash
# round 6, GPU call 6: launch header in preloaded SGPRs, no dummy lookups / epilogue fetches, bias under the loop, quantisation-aware splits: tests + K slopes + sweeps
cp ab/lib_fix.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1500 python -m pytest tests/test_0_tgemm_gpu.py tests/test_rbnet_gpu.py tests/test_baseline_width_gpu.py -x -q > gpurun_out/r06_run6_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_run6_tests.txt
tail -5 gpurun_out/r06_run6_tests.txt
for v in r5 fix; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v"
python tools/probes/tgemm_kslope.py 2048 512 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 2048 512 "" 1 0 3 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 6144 1024 "" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06_run6_kslope.txt 2>&1
cat gpurun_out/r06_run6_kslope.txt
for v in r5 fix; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v"
timeout 600 python tools/tgemm_sweep.py --apex --hopper --cfg "" 2>/dev/null | cut -c1-1000
JH_TGEMM_QUANT=0 timeout 600 python tools/tgemm_sweep.py --apex --hopper --cfg "" 2>/dev/null | cut -c1-1000
done > gpurun_out/r06_run6_sweep.txt 2>&1
cat gpurun_out/r06_run6_sweep.txt
cp ab/lib_fix.so jorldy_amd/csrc/libjorldy_hip.so
