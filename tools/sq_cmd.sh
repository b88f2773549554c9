#!/bin/bash
# Where do a kernel's wave cycles go?  One rocprofv3 PMC pass with the SQ wait / issue counters + the clock counter (run on the GPU box
# from the repo root):   tools/sq_cmd.sh <tag> <command...>
# -> gpurun_out/<tag>_sq.json  {kernel: {launches, dur_us, <counter>_mean ..., derived: wait_frac, issue_stall_frac, active_frac,
#    mfma_busy_frac_of_simd_cycles, eff_clock_GHz}}
# MI355X_MICROARCH.md (rocprofv3 PMC slots): SQ has 8 slots, GRBM 2; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, summed over waves);
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles; effective clock = GRBM_GUI_ACTIVE / kernel wall time.  Counters only with --kernel-trace (no other trace domain).
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
o=$repo/gpurun_out/pmc_${tag}_sq
rm -rf $o; mkdir -p $o
( cd $repo && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o -o run -- "$@" > $o/cmd.json 2> $o/err.log )
python - "$repo" "$tag" <<'PY'
import csv, glob, json, sys, collections
repo, tag = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
dur = {}
for f in glob.glob(f"{repo}/gpurun_out/pmc_{tag}_sq/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        d = dur.setdefault(row["Kernel_Name"][:120], [0, 0.0])
        d[0] += 1
        d[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
for f in glob.glob(f"{repo}/gpurun_out/pmc_{tag}_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = acc.setdefault(row["Kernel_Name"][:120], collections.OrderedDict())
        c = a.setdefault(row["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(row["Counter_Value"])
res = collections.OrderedDict()
for k, cs in acc.items():
    e = {"launches": max(n for n, _ in cs.values())}
    for c, (n, s) in cs.items():
        e[c + "_mean"] = round(s / n, 1)
    if k in dur and dur[k][0]:
        e["dur_us_under_pmc"] = round(dur[k][1] / dur[k][0], 2)
    wc = e.get("SQ_WAVE_CYCLES_mean")
    if wc:
        for c, name in (("SQ_WAIT_ANY", "wait_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "active_frac"), ("SQ_WAIT_INST_LDS", "lds_issue_stall_frac")):
            if c + "_mean" in e:
                e[name] = round(e[c + "_mean"] / wc, 3)
    if "GRBM_GUI_ACTIVE_mean" in e and e.get("dur_us_under_pmc"):
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (a long MFMA kernel reads 19.3 "GHz" = 8 x 2.41)
        gui = e["GRBM_GUI_ACTIVE_mean"] / 8.0
        e["eff_clock_GHz"] = round(gui / e["dur_us_under_pmc"] / 1e3, 3)  # includes the launch's ramp: short kernels read high
        if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in e:  # busy cycles summed over SIMDs / (1024 SIMDs x kernel cycles)
            e["mfma_busy_frac_of_simd_cycles"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (1024.0 * gui), 3)
    res[k] = e
json.dump(res, open(f"{repo}/gpurun_out/{tag}_sq.json", "w"), indent=1)
top = sorted(res.items(), key=lambda kv: -kv[1].get("dur_us_under_pmc", 0) * kv[1]["launches"])[:12]
for k, e in top:
    print(k[:70], {x: e.get(x) for x in ("launches", "dur_us_under_pmc", "wait_frac", "issue_stall_frac", "active_frac", "lds_issue_stall_frac", "mfma_busy_frac_of_simd_cycles", "eff_clock_GHz")})
PY
tail -3 $o/err.log
rm -rf $o  # the raw per-launch CSVs are tens of MB (gpurun merges at most 64 MB back)
