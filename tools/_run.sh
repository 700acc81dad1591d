export TRON_TUNING=1
timeout 1500 python -m pytest tests/test_gpu_scatter.py -x -q -m gpu -k "random_shapes" 2>&1 | tail -8
