#!/bin/bash
# rocprofv3 kernel-trace statistics of an arbitrary command, ONE pass (tools/profile_cmd.sh adds the two PMC passes):
#   tools/profile_stats_cmd.sh <tag> <command...>  ->  gpurun_out/<tag>_kernel_stats.csv, <tag>_under_rocprof.json
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out
( cd $repo && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- "$@" > $repo/gpurun_out/${tag}_under_rocprof.json 2> $repo/gpurun_out/${tag}_rocprof.err )
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $repo/gpurun_out/${tag}_kernel_stats.csv
head -14 $repo/gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
rm -rf $out
