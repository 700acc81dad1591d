# usage (GPU box): bash tools/bench_ab.sh -- bench.py value under a few switches, same box (no CPU baselines)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'ms/step', d['ms_per_step'], 'grid', r['launch_ms'], 'frac', r['frac'], 'alg', d.get('algorithmic_frac_of_peak'), 'parity', d.get('parity_rel_l2_vs_oracle'))"; }
B="python bench.py --cpu-slices 0 --no-irt"
run "arc 8 coils      " $B
run "arc 1 coil       " $B --coils 1
run "binned 1 coil    " TRON_GRID_KERNEL=binned $B --coils 1
run "arc half         " $B --half
run "binned half      " TRON_GRID_KERNEL=binned $B --half
run "arc half 4 coils " $B --half --coils 4
run "linear 1 coil    " $B --coils 1 --linear
