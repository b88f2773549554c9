#!/bin/bash
# Repeat the GPU suite up to the test in which the intermittent "capture invalidated" failure shows up, with the HIP runtime's error log on, until it fails.
n=${1:-8}
for i in $(seq 1 $n); do
  AMD_LOG_LEVEL=${LOGLEVEL:-1} timeout 300 python -m pytest tests -m gpu -q -x -s > gpurun_out/repro_$i.log 2>&1
  if grep -q "capture of learn() failed" gpurun_out/repro_$i.log; then echo "FAILED at run $i"; grep -n ":1:\|hipError\|capture" gpurun_out/repro_$i.log | grep -v "Search for" | head -60; exit 0; fi
  tail -1 gpurun_out/repro_$i.log
  rm -f gpurun_out/repro_$i.log
done
echo "no failure in $n runs"
