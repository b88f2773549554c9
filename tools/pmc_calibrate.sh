#!/bin/bash
# Calibration of the rocprofv3 HBM counters per access width on this GPU (MI355X_MICROARCH.md asks for it for anything but 16-byte loads):
#   tools/pmc_calibrate.sh          (on the GPU box, from the repo root)  ->  gpurun_out/pmc_calibration.json
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum; do
  out=$repo/gpurun_out/pmc_cal_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o run -- python $repo/tools/pmc_calibrate.py > $out/log.txt 2> $out/err.log || echo "counter $c: rocprofv3 failed (not available?)"
done
python - "$repo" <<'PY'
import csv, glob, json, sys, collections
repo = sys.argv[1]
res = collections.OrderedDict()
for d in sorted(glob.glob(f"{repo}/gpurun_out/pmc_cal_*")):
    c = d.split("pmc_cal_")[1]
    acc = collections.OrderedDict()
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "jh_calib_stream_kernel" not in row["Kernel_Name"]:
                continue
            a = acc.setdefault(row["Kernel_Name"][:60], [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    res[c] = {k: {"launches": n, "mean": s / n, "bytes_per_unit_if_1GiB": (1 << 30) / (s / n) if s else None} for k, (n, s) in acc.items()}
json.dump(res, open(f"{repo}/gpurun_out/pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
