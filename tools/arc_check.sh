# usage (GPU box): bash tools/arc_check.sh  -- arc kernel vs binned kernel: timing (gridbench) and headline parity tests
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
export TRON_ARC_DEBUG=1
echo "== gridbench arc"; python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== gridbench binned"; TRON_GRID_KERNEL=binned python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== gridbench arc"; python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== gridbench binned"; TRON_GRID_KERNEL=binned python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
for nc in 6 4 2; do
echo "== nc=$nc arc"; python tools/gridbench.py $nc 64 fast 5 2>&1 | tail -1
echo "== nc=$nc binned"; TRON_GRID_KERNEL=binned python tools/gridbench.py $nc 64 fast 5 2>&1 | tail -1
done
echo "== headline tests"; timeout 900 python -m pytest tests/test_gpu_headline.py -x -q 2>&1 | tail -15
