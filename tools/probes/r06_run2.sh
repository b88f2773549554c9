#!/bin/bash
# round 6, GPU call 2: phase timeline of the LDS-DMA GEMM (trace build) on the Hopper / Ape-X shapes
cp jorldy_amd/csrc/libjorldy_hip.so /tmp/keep.so
cp ab/lib_trace.so jorldy_amd/csrc/libjorldy_hip.so
{
python tools/probes/tgemm_trace.py 2048 512 512 ""
python tools/probes/tgemm_trace.py 2048 512 512 "0:2x2:s2"
python tools/probes/tgemm_trace.py 2048 512 512 "0:4x2"
python tools/probes/tgemm_trace.py 1536 1024 3136 "" 1 1 2
python tools/probes/tgemm_trace.py 1536 1024 3136 "0:2x2:s2" 1 1 2
python tools/probes/tgemm_trace.py 1536 1024 3136 "0:4x2:s4" 1 1 2
python tools/probes/tgemm_trace.py 2048 512 512 "" 1 0 3
} > gpurun_out/r06_run2_trace.txt 2>&1
cp /tmp/keep.so jorldy_amd/csrc/libjorldy_hip.so
cat gpurun_out/r06_run2_trace.txt
