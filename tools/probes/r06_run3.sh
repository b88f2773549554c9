#!/bin/bash
# round 6, GPU call 3: is the LDS-DMA stream the loop's limit?  trace builds without operand B's / both operands' DMA (wrong results), 3 and 4 LDS buffers
cp jorldy_amd/csrc/libjorldy_hip.so /tmp/keep.so
{
for v in trace skipb skipab nb3 nb4; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v"
python tools/probes/tgemm_trace.py 2048 512 512 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_trace.py 1536 1024 3136 "0:2x2:s2" 1 1 2 2>&1 | grep -v amdgpu.ids | grep -v "hand-off\|stores"
done
for v in trace skipb skipab; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v 4x2"
python tools/probes/tgemm_trace.py 1536 1024 3136 "0:4x2:s4" 1 1 2 2>&1 | grep -v amdgpu.ids | grep -v "hand-off\|stores"
done
} > gpurun_out/r06_run3_trace.txt 2>&1
cp /tmp/keep.so jorldy_amd/csrc/libjorldy_hip.so
cat gpurun_out/r06_run3_trace.txt
