# usage (GPU box): bash tools/run_var_fwd.sh [rounds] [fwdbench args]  -- tools/run_var.sh for the forward direction
export TRON_TUNING=1
R=${1:-3}; shift; A=${@:-8 64 fast}
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for r in $(seq $R); do
  for f in /tmp/orig.so tron_amd/lib/libtronhip_*.so; do
    cp $f tron_amd/lib/libtronhip.so; echo -n "$(basename $f) : "; python tools/fwdbench.py $A 2>&1 | grep -v "^W\|amdgpu" | tail -1
  done
done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
