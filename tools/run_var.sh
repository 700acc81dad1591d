# usage (GPU box): bash tools/run_var.sh [gridbench args]  -- times every tron_amd/lib/libtronhip_*.so variant, then the default library
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for f in tron_amd/lib/libtronhip_*.so; do cp $f tron_amd/lib/libtronhip.so; echo "variant $f"; python tools/gridbench.py ${@:-8 128 fast 2}; TRON_DUAL_STREAM=1 python tools/gridbench.py ${@:-8 128 fast 2}; done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
echo default; python tools/gridbench.py ${@:-8 128 fast 2}; TRON_DUAL_STREAM=1 python tools/gridbench.py ${@:-8 128 fast 2}
