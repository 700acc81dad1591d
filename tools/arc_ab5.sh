export TRON_TUNING=1
for i in 1 2; do
for nz in 32 16 48; do
echo -n "new nz=$nz: "; python tools/gridbench.py 8 $nz fast 5 2>&1 | tail -1
cp tron_amd/lib/libtronhip.so /tmp/new.so; cp tron_amd/lib/libtronhip_old.so tron_amd/lib/libtronhip.so
echo -n "old nz=$nz: "; python tools/gridbench.py 8 $nz fast 5 2>&1 | tail -1
cp /tmp/new.so tron_amd/lib/libtronhip.so
done
done
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_headline.py -x -q 2>&1 | tail -2
