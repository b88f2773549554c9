#!/bin/bash
cp ab/lib_fixt.so jorldy_amd/csrc/libjorldy_hip.so
python tools/probes/tgemm_trace.py 2048 512 512 "" 2>&1 | grep -v amdgpu.ids
python tools/probes/tgemm_trace.py 2048 512 512 "" 1 0 3 2>&1 | grep -v amdgpu.ids
cp ab/lib_fix.so jorldy_amd/csrc/libjorldy_hip.so
