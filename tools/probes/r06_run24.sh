#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for t in ppo_atari hopper rainbow; do
  case $t in
    ppo_atari) cmd="python $repo/tools/bench_ppo_atari.py --iters 3";;
    hopper) cmd="python $repo/tools/bench_hopper.py --iters 2";;
    rainbow) cmd="python $repo/tools/bench_rainbow.py";;
  esac
  rm -rf /tmp/kt_$t; ( cd $repo && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$t -o run -- $cmd > /dev/null 2>&1 )
  f=$(find /tmp/kt_$t -name "*kernel_trace.csv" | head -1)
  echo "== $t"; python $repo/tools/probes/kernel_gaps.py $f 3000
done > $repo/gpurun_out/r24_gaps.txt 2>&1
cat $repo/gpurun_out/r24_gaps.txt | cut -c1-150
