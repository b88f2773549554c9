#!/bin/bash
# round 6, GPU call 5: fixed vs per-chunk cost of the GEMM launches, old loop / new loop / 3 buffers, several shapes and tiles
for v in r5 new nb3; do
cp ab/lib_$v.so jorldy_amd/csrc/libjorldy_hip.so
echo "=== $v"
python tools/probes/tgemm_kslope.py 2048 512 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 2048 512 "0:4x2" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 2048 512 "0:2x2:s2" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 1536 1024 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 3072 1024 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 3072 1024 "0:4x2" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 6144 1024 "0:4x2" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_kslope.py 6144 1024 "" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06_run5_kslope.txt 2>&1
cat gpurun_out/r06_run5_kslope.txt
cp ab/lib_new.so jorldy_amd/csrc/libjorldy_hip.so
