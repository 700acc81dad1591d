#!/bin/bash
# usage (GPU box): bash tools/bisect.sh [coils] [slices]  -- grid_binned_kernel phase bisection (TRON_DEBUG_SKIP 0..4), unprofiled
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
NC=${1:-8}; NZ=${2:-64}
# needs a library built with the knob: tools/build_variants.sh debug:"-DTRON_DEBUG_KNOBS" && cp tron_amd/lib/libtronhip_debug.so tron_amd/lib/libtronhip.so
for s in 0 1 2 3 4 0; do TRON_DEBUG_SKIP=$s python tools/gridbench.py $NC $NZ fast 5; done
