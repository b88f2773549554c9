for cfg in "4096 8" "2048 32" "2048 64" "1024 64" "4096 8" "2048 32" "2048 64" "1024 64"; do set -- $cfg
JH_TGEMM_SMALL_TILE_K=$1 JH_TGEMM_SMALL_TILE_MAXTILES=$2 python tools/bench_rainbow.py --updates 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('k>=$1 tiles<=$2', round(d['learner_updates_per_s']), round(d['ms_per_learn_only'],4))
"; done
