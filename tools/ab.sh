# usage (GPU box): bash tools/ab.sh [script args...]   -- A/B on ONE box: tron_amd/lib/libtronhip_old.so vs the current library,
# three alternating runs each (box-to-box variance is +-3 %, larger than most single optimisations)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
S=${1:-tools/gridbench.py}; shift; A=${@:-8 128 fast 3}
cp tron_amd/lib/libtronhip.so /tmp/new.so
for i in 1 2 3; do
cp tron_amd/lib/libtronhip_old.so tron_amd/lib/libtronhip.so; echo old; python $S $A
cp /tmp/new.so tron_amd/lib/libtronhip.so; echo new; python $S $A
done
