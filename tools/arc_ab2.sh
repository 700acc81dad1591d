export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
for i in 1 2; do
echo "== arc default"; python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== inner in front"; TRON_ARC_INNER_STREAM=0 python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== zper 1"; TRON_ARC_ZPER=1 python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== zper 2"; TRON_ARC_ZPER=2 python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== zper 8"; TRON_ARC_ZPER=8 python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== binned"; TRON_GRID_KERNEL=binned python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
done
for nc in 6 4 2; do
echo "== nc=$nc arc"; python tools/gridbench.py $nc 64 fast 5 2>&1 | tail -1
echo "== nc=$nc binned"; TRON_GRID_KERNEL=binned python tools/gridbench.py $nc 64 fast 5 2>&1 | tail -1
done
echo "== 32 slices arc"; python tools/gridbench.py 8 32 fast 5 2>&1 | tail -1
echo "== 32 slices binned"; TRON_GRID_KERNEL=binned python tools/gridbench.py 8 32 fast 5 2>&1 | tail -1
echo "== headline tests"; timeout 900 python -m pytest tests/test_gpu_headline.py -x -q 2>&1 | tail -5
