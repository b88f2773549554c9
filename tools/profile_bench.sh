#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag> [bench.py args...]
# -> gpurun_out/<tag>_kernel_stats.csv (+ the bench line it produced, <tag>_under_rocprof.json)
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python $repo/bench.py "$@" > $repo/gpurun_out/${tag}_under_rocprof.json 2> $repo/gpurun_out/${tag}_rocprof.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $repo/gpurun_out/${tag}_kernel_stats.csv
head -25 $repo/gpurun_out/${tag}_kernel_stats.csv | cut -c1-170
