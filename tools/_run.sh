export TRON_TUNING=1
python tools/_chk.py 2>&1 | grep "rel l2"
timeout 1200 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu 2>&1 | tail -5
for a in "1 128" "2 128"; do WARM=20 python tools/gridbench.py $a fast 5 2>&1 | tail -1; done
