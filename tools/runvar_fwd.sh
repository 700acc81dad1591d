export TRON_TUNING=1
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for r in 1 2; do
  for f in /tmp/orig.so tron_amd/lib/libtronhip_cb4.so tron_amd/lib/libtronhip_cb16.so tron_amd/lib/libtronhip_cb32.so; do
    cp $f tron_amd/lib/libtronhip.so; echo -n "$(basename $f) : "; python tools/fwdbench.py 8 64 fast | tail -1
  done
done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
