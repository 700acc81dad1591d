export TRON_TUNING=1
NI="--cpu-slices 0 --no-irt"
for a in "--coils 1" "--coils 1 --half" "--coils 1 --linear" "--coils 1 --slices 32"; do echo -n "bench $a: "; python bench.py $NI $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['sustained_slices_per_s'], d['parity_rel_l2_vs_oracle'], d['roofline']['kernel'][:24], d['roofline']['frac'])"; done
