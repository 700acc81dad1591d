cp tron_amd/lib/libtronhip.so /tmp/new.so
for i in 1 2 3; do
cp tron_amd/lib/libtronhip_old.so tron_amd/lib/libtronhip.so; echo old; python tools/gridbench.py 8 128 fast 3
cp /tmp/new.so tron_amd/lib/libtronhip.so; echo new; python tools/gridbench.py 8 128 fast 3
done
