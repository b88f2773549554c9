#!/bin/bash
# SQ counters of the tiled GEMM launches of one learner (run on the GPU box from the repo root):
#   tools/pmc_tgemm.sh <tag> <command...>     e.g.  tools/pmc_tgemm.sh r02_apex python tools/bench_apex.py --updates 10 --warmup 3
# One rocprofv3 pass (8 SQ slots, --kernel-trace only); launches are told apart by (kernel instantiation, grid size).
# -> gpurun_out/<tag>_pmc_tgemm.json
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out/pmc_tgemm_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
( cd $repo && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out -o run -- "$@" > $out/cmd.out 2> $out/err.log )
python - "$repo" "$tag" <<'PY'
import csv, glob, json, sys, collections
repo, tag = sys.argv[1], sys.argv[2]
files = glob.glob(f"{repo}/gpurun_out/pmc_tgemm_{tag}/**/*counter_collection.csv", recursive=True)
acc = collections.OrderedDict()
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "tgemm" not in name and "pmb" not in name:
            continue
        key = (name[:60], row.get("Grid_Size", "?"))
        a = acc.setdefault(key, collections.OrderedDict())
        c = a.setdefault(row["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(row["Counter_Value"])
res = []
for (name, grid), cs in acc.items():
    m = {k: v[1] / v[0] for k, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    res.append({"kernel": name, "grid": grid, "dispatches": next(iter(cs.values()))[0], **{k: round(v, 1) for k, v in m.items()},
                "frac_wait_any": round(m.get("SQ_WAIT_ANY", 0) / wc, 3), "frac_issue_stall": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                "frac_active": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "frac_valu": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                "frac_lds": round(m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)})
res.sort(key=lambda r: -r.get("SQ_WAVE_CYCLES", 0))
json.dump(res, open(f"{repo}/gpurun_out/{tag}_pmc_tgemm.json", "w"), indent=1)
for r in res[:16]:
    print(r["kernel"][:44].ljust(46), str(r["grid"]).rjust(8), "wave_cyc", int(r.get("SQ_WAVE_CYCLES", 0)), "wait", r["frac_wait_any"], "stall", r["frac_issue_stall"],
          "active", r["frac_active"], "valu", r["frac_valu"], "lds", r["frac_lds"], "mfma_busy", int(r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)), "bank_conf", int(r.get("SQ_LDS_BANK_CONFLICT", 0)))
PY
