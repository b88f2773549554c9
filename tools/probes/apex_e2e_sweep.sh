#!/bin/bash
# End-to-end Ape-X (tools/bench_apex.py --e2e N --device-feed --frames) for several actor counts
for n in "$@"; do
  timeout 500 python tools/bench_apex.py --e2e $n --updates 300 --device-feed --frames 2>/dev/null | grep "^{" > /tmp/l.json
  python - $n <<'PY'
import json, sys
d = json.loads(open("/tmp/l.json").readline())
e = d["end_to_end"]
print("actors", sys.argv[1], "env_steps_per_s", round(e["env_steps_per_s"]), "ticks_per_s", round(e["actor_ticks_per_s"]), "learner_updates_per_s", round(d["learner_updates_per_s"], 1),
      "act_ms", round(e["act_ms_per_tick"], 3), "host_ms", round(e["host_ms_per_tick"], 3), "pool_KB_per_row", round(e["pool_bytes_per_buffer_row"] / 1024, 1))
PY
done
