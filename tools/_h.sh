for h in 0 1 0 1; do echo "== GR_TILE_HISTORY=$h"; GR_TILE_HISTORY=$h python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('frame_one_at_a_time_ms'))"; done
for h in 0 1; do GR_TILE_HISTORY=$h TILE_HISTORY_INFLIGHT=3 python tools/tile_history_probe.py 0.45 2>&1 | grep "in flight"; done
