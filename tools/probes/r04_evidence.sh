set -x
cd $GRAFT_REPO_ROOT
# 1. bench line (final code), plain
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
tail -c 600 gpurun_out/r04_bench_final.json
# 2. rocprofv3 kernel stats + PMC of the same command (shorter legs to keep it quick)
tools/profile_bench.sh r04_bench --steps 60 --warmup 10 --no-cpu-baseline --apex-updates 600 --rainbow-updates 200 > gpurun_out/r04_profile_bench.log 2>&1
tools/pmc_bench.sh r04 --steps 40 --warmup 10 --no-cpu-baseline --no-apex --no-hopper --rainbow-updates 100 > gpurun_out/r04_pmc_bench.log 2>&1
# 3. apex: rocprof + pmc
tools/profile_cmd.sh r04_apex python tools/bench_apex.py --e2e 64 --device-feed --frames --updates 500 --warmup 30 > gpurun_out/r04_profile_apex.log 2>&1
# 4. in-kernel stamps of the acting kernel, both forms
for la in 2 1; do JH_COLLECT_LOOKAHEAD=$la JH_PERSIST_DEBUG=1 python bench.py --steps 60 --warmup 10 --no-rainbow --no-apex --no-hopper --no-cpu-baseline --no-roofline 2> gpurun_out/tmp.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lookahead $la', round(d['value']), round(d['ms_per_step'],4), d['collector_host_us_per_timestep'])"; grep jh_persist gpurun_out/tmp.err | tail -2; done > gpurun_out/r04_acting_in_kernel.txt 2>&1
cat gpurun_out/r04_acting_in_kernel.txt
# 5. scaled rooflines
python tools/roofline_scaled.py --out gpurun_out/r04_roofline_scaled.json > gpurun_out/r04_roofline_scaled.log 2>&1
grep gae gpurun_out/r04_roofline_scaled.log | cut -c1-200
