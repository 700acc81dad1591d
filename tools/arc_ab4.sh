export TRON_TUNING=1
for r in 1 2; do
for f in /tmp/orig.so tron_amd/lib/libtronhip_w6.so tron_amd/lib/libtronhip_w5.so; do
  cp $f tron_amd/lib/libtronhip.so
  for nc in 1 2; do echo -n "$(basename $f) nc=$nc: "; python tools/gridbench.py $nc 256 fast 5 2>&1 | tail -1; done
done
done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
