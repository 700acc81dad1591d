export TRON_TUNING=1
for z in 4 5 3 6 8 16 4 5; do
  echo -n "zper=$z: "; TRON_ARC_ZPER=$z python bench.py --coils 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']), round(j['sustained_slices_per_s']), j['config'].get('workload','')[:60])"
done
