#!/bin/bash
# Which host cores are close to the GPU?  Acting is two PCIe crossings per timestep: run the bench pinned to each NUMA node.
echo "nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)"; nproc
for f in /sys/class/drm/card*/device/numa_node; do echo "$f: $(cat $f 2>/dev/null)"; done
for f in /sys/class/kfd/kfd/topology/nodes/*/properties; do echo "$f: $(grep -E 'simd_count|cpu_cores_count' $f | tr '\n' ' ')"; done 2>/dev/null | head -12
for n in /sys/devices/system/node/node*; do
  cpus=$(cat $n/cpulist); echo "== $(basename $n) cpus $cpus"
  for rep in 1 2; do
    JH_PERSIST_GROUPS=1 taskset -c $cpus timeout 120 python bench.py --no-cpu-baseline --no-rainbow --no-roofline 2>/dev/null > /tmp/nb.json
    python -c "
import json
d=json.loads(open('/tmp/nb.json').read().strip().splitlines()[-1]); print('  ', round(d['value']), round(d['ms_per_step'],3), d['collector_host_us_per_timestep'])"
  done
done
