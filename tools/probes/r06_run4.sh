#!/bin/bash
# round 6, GPU call 4: the hand-pipelined LDS-DMA loop: engine tests, phase traces (new loop at 2 / 3 / 4 buffers, without DMA), sweeps per library
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
timeout 1200 python -m pytest tests/test_0_tgemm_gpu.py -x -q > gpurun_out/r06_run4_tgemm_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_run4_tgemm_tests.txt
tail -5 gpurun_out/r06_run4_tgemm_tests.txt
{
for v in trace nb3t nb4t skipab; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v"
python tools/probes/tgemm_trace.py 2048 512 512 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_trace.py 1536 1024 3136 "0:2x2:s2" 1 1 2 2>&1 | grep -v amdgpu.ids | grep -v "hand-off\|stores"
python tools/probes/tgemm_trace.py 1536 1024 3136 "0:4x2:s4" 1 1 2 2>&1 | grep -v amdgpu.ids | grep -v "hand-off\|stores"
done
} > gpurun_out/r06_run4_trace.txt 2>&1
cat gpurun_out/r06_run4_trace.txt
for v in r5 new nb3 nb4; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v"
timeout 600 python tools/tgemm_sweep.py --apex --hopper --cfg "" --cfg "*:4x2" --cfg "6:2x2:s2" --cfg "6:4x2:s4" --cfg "6:4x2:s2" --cfg "15:2x2:s2,16:2x2:s2" --cfg "2:4x2,3:4x2" 2>/dev/null | cut -c1-900
done > gpurun_out/r06_run4_sweep.txt 2>&1
cat gpurun_out/r06_run4_sweep.txt
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
