export TRON_TUNING=1
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for r in 1 2; do
for v in /tmp/orig.so tron_amd/lib/libtronhip_s41.so tron_amd/lib/libtronhip_s42.so tron_amd/lib/libtronhip_s44.so tron_amd/lib/libtronhip_r5.so tron_amd/lib/libtronhip_r4.so; do cp $v tron_amd/lib/libtronhip.so; echo -n "$(basename $v) "; WARM=20 python tools/gridbench.py 1 128 fast 5 2>&1 | tail -1; done
done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
python -m pytest tests/test_irt.py -x -q -m gpu 2>&1 | tail -3
