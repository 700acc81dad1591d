# usage (GPU box): bash tools/run_var.sh [rounds] [gridbench args]  -- interleaved rounds over every tron_amd/lib/libtronhip_*.so variant and the
# default library (A/B deltas from interleaved rounds on ONE box; box-to-box variance is larger than most single optimisations)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
R=${1:-3}; shift; A=${@:-8 64 fast 5}
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for r in $(seq $R); do
  for f in /tmp/orig.so tron_amd/lib/libtronhip_*.so; do
    cp $f tron_amd/lib/libtronhip.so; echo -n "$(basename $f) : "; python tools/gridbench.py $A 2>&1 | grep -v "^W\|amdgpu" | tail -1
  done
done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
