#!/bin/bash
# rocprofv3 evidence for an arbitrary command (run on the GPU box from the repo root):
#   tools/profile_cmd.sh <tag> <command...>     e.g.  tools/profile_cmd.sh r04_apex python tools/bench_apex.py --e2e 64 --device-feed --frames --updates 400
# Three runs of the command, as MI355X_MICROARCH.md prescribes (counters never share a pass with each other or with --stats):
#   1  --kernel-trace --stats          -> gpurun_out/<tag>_kernel_stats.csv   (per-kernel calls / average duration)
#   2  --pmc FETCH_SIZE --kernel-trace -> \ gpurun_out/<tag>_pmc.json  {kernel: {FETCH_SIZE_KB_mean, WRITE_SIZE_KB_mean, launches_*}}  (raw KiB;
#   3  --pmc WRITE_SIZE --kernel-trace -> /  readers apply the guide's gfx950 correction FETCH_SIZE x 2)
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
( cd $repo && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- "$@" > $repo/gpurun_out/${tag}_under_rocprof.json 2> $repo/gpurun_out/${tag}_rocprof.err )
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $repo/gpurun_out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  o=$repo/gpurun_out/pmc_${tag}_$c
  rm -rf $o; mkdir -p $o
  ( cd $repo && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o -o run -- "$@" > $o/cmd.json 2> $o/err.log )
done
python - "$repo" "$tag" <<'PY'
import csv, glob, json, sys, collections
repo, tag = sys.argv[1], sys.argv[2]
res = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.OrderedDict()
    for f in glob.glob(f"{repo}/gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c:
                continue
            a = acc.setdefault(row["Kernel_Name"][:120], [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    for k, (n, s) in acc.items():
        e = res.setdefault(k, {})
        e[f"{c}_KB_mean"] = round(s / n, 2)
        e[f"launches_{c}"] = n
json.dump(res, open(f"{repo}/gpurun_out/{tag}_pmc.json", "w"), indent=1)
PY
head -16 $repo/gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
