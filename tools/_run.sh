export TRON_TUNING=1
timeout 1500 python -m pytest tests/test_gpu_arc.py tests/test_gpu_headline.py -x -q -m gpu -k "half or shapes_the_arc" 2>&1 | tail -8
NI="--cpu-slices 0 --no-irt --sustain 0"
for a in "--half --coils 6" "--coils 6" "--half --coils 8" ; do echo -n "$a: "; python bench.py $NI $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['parity_rel_l2_vs_oracle'], d['roofline']['kernel'], d['roofline']['frac'])"; done
