#!/bin/bash
# HBM traffic per kernel of the bench command from rocprofv3 PMC counters, collected the way MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass), --kernel-trace only.
#   tools/pmc_bench.sh <tag> [bench.py args...]      (run on the GPU box from the repo root)
# -> gpurun_out/<tag>_pmc_bench.json  {kernel: {FETCH_SIZE_KB_mean, WRITE_SIZE_KB_mean, launches_*}}  (raw counter
#    units: KiB; bench.py applies the guide's gfx950 correction, FETCH_SIZE x 2, when it reads the summary)
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$repo/gpurun_out/pmc_${tag}_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o run -- python $repo/bench.py "$@" > $out/bench.json 2> $out/err.log
done
python - "$repo" "$tag" <<'PY'
import csv, glob, json, sys, collections
repo, tag = sys.argv[1], sys.argv[2]
res = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{repo}/gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.OrderedDict()
    for f in files:
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
json.dump(res, open(f"{repo}/gpurun_out/{tag}_pmc_bench.json", "w"), indent=1)
for k, v in res.items():
    if k.startswith(("void jh_", "jh_")):
        print(k[:70].ljust(72), v)
PY
