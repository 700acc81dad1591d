#!/bin/bash
# usage (GPU box): bash tools/flaky.sh [n]  -- the same parity tests n times in fresh processes (intermittent races / faults show up as differing outcomes)
n=${1:-6}
for i in $(seq $n); do
  python -m pytest tests/test_gpu_arc.py -x -q -k "workers or half or vs_oracle" 2>&1 | grep -v "^  File" | tail -2 | tr '\n' ' '; echo
done
echo "== centre kernel off (binned inner tile)"
for i in $(seq $n); do
  TRON_TUNING=1 TRON_CENTRE_KERNEL=binned python -m pytest tests/test_gpu_arc.py -x -q -k "workers or half or vs_oracle" 2>&1 | grep -v "^  File" | tail -2 | tr '\n' ' '; echo
done
