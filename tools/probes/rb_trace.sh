repo=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rbt; rocprofv3 --kernel-trace --output-format csv -d /tmp/rbt -o run -- python $repo/tools/bench_rainbow.py --buffer 100000 --updates 40 --warmup 8 > $repo/gpurun_out/rb_trace.out 2>&1
f=$(find /tmp/rbt -name "*kernel_trace.csv" | head -1)
python - $f > $repo/gpurun_out/rb_trace_seq.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 3 learn()s worth: find last occurrences of per_sample
idx = [i for i, r in enumerate(rows) if "per_sample" in r["Kernel_Name"]]
import os
if os.environ.get("RB_TRACE_MID"):
    a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
else:
    a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}/{r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}  {r['Kernel_Name'][:90]}")
    prev_end = e
print("launches", b - a, "span us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
PY
