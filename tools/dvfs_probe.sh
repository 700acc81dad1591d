#!/bin/bash
# usage (GPU box): bash tools/dvfs_probe.sh  -- is the gridding kernel held down by power?  Times every tron_amd/lib/libtronhip_*.so variant and the
# default library on random and on all-zero k-space (same instruction stream, no toggling operands), then reads the in-kernel clock
# (s_memtime / s_memrealtime) of the -DTRON_ARC_PROFILE build on both.
export TRON_TUNING=1
cp tron_amd/lib/libtronhip.so /tmp/orig.so
[ -f tron_amd/lib/libtronhip_aprof.so ] && mv tron_amd/lib/libtronhip_aprof.so /tmp/aprof.so
for r in 1 2; do
  for f in /tmp/orig.so tron_amd/lib/libtronhip_*.so; do
    for d in random zeros; do
      cp $f tron_amd/lib/libtronhip.so; echo -n "$(basename $f) $d : "; DATA=$d python tools/gridbench.py 8 128 fast 8 2>&1 | grep -v "^W\|amdgpu" | tail -1
    done
  done
done
if [ -f /tmp/aprof.so ]; then
  cp /tmp/aprof.so tron_amd/lib/libtronhip.so
  for d in random zeros; do echo "== phase clock, $d data"; DATA=$d python tools/arcprof.py 8 128 2>&1 | grep -v "^W\|amdgpu"; done
fi
cp /tmp/orig.so tron_amd/lib/libtronhip.so
