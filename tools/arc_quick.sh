# usage (GPU box): bash tools/arc_quick.sh  -- arc kernel: timing at 8 and 2 coils (+ binned for reference), phase clock, headline parity
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
for i in 1 2; do
echo "== arc 8";  python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== binned 8"; TRON_GRID_KERNEL=binned python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
done
echo "== arc 8 zper=4";  TRON_ARC_ZPER=4 python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
echo "== arc 2";  python tools/gridbench.py 2 64 fast 5 2>&1 | tail -1
echo "== arc 2 zper=4";  TRON_ARC_ZPER=4 python tools/gridbench.py 2 64 fast 5 2>&1 | tail -1
cp tron_amd/lib/libtronhip.so /tmp/orig.so; cp tron_amd/lib/libtronhip_aprof.so tron_amd/lib/libtronhip.so
python tools/arcprof.py 8 64 2>&1 | tail -16
cp /tmp/orig.so tron_amd/lib/libtronhip.so
echo "== headline tests"; timeout 900 python -m pytest tests/test_gpu_headline.py -x -q 2>&1 | tail -5
