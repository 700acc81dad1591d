export TRON_TUNING=1
timeout 1500 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu -k "vs_oracle_and_arc" 2>&1 | tail -3
NI="--cpu-slices 0 --no-irt --sustain 0"
echo -n "bench nc1 804: "; python bench.py $NI --coils 1 --spokes 804 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['parity_rel_l2_vs_oracle'], d['roofline']['kernel'][:24])"
