# usage (GPU box): bash tools/knob_sweep.sh "bench args" VAR v1 v2 ...  -- bench.py's short and sustained figures per value of one tuning variable ("-" = unset)
export TRON_TUNING=1
A=$1; V=$2; shift 2
for x in "$@"; do
  if [ "$x" = "-" ]; then unset $V; else export $V=$x; fi
  echo -n "$V=$x: "; python bench.py $A --steps 20 --warmup 5 --cpu-slices 0 --no-irt --no-check 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']), round(j.get('sustained_slices_per_s') or 0), j['roofline']['frac'])"
done
