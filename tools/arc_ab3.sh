export TRON_TUNING=1
for i in 1 2; do
echo "== new";  python tools/gridbench.py 8 128 fast 5 2>&1 | tail -1
echo "== old"; cp tron_amd/lib/libtronhip.so /tmp/new.so; cp tron_amd/lib/libtronhip_old.so tron_amd/lib/libtronhip.so; python tools/gridbench.py 8 128 fast 5 2>&1 | tail -1; cp /tmp/new.so tron_amd/lib/libtronhip.so
echo "== new r0=10";  TRON_INNER_R0=10 python tools/gridbench.py 8 128 fast 5 2>&1 | tail -1
echo "== new r0=8";  TRON_INNER_R0=8 python tools/gridbench.py 8 128 fast 5 2>&1 | tail -1
echo "== new r0=6";  TRON_INNER_R0=6 python tools/gridbench.py 8 128 fast 5 2>&1 | tail -1
done
echo "== 32 slices new";  python tools/gridbench.py 8 32 fast 5 2>&1 | tail -1
echo "== 32 slices r0=8";  TRON_INNER_R0=8 python tools/gridbench.py 8 32 fast 5 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -k "metric or config4" 2>&1 | tail -2
TRON_INNER_R0=8 timeout 600 python -m pytest tests/test_gpu_headline.py -x -q -k "metric or config4" 2>&1 | tail -2
