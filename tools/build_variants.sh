#!/bin/bash
# usage: tools/build_variants.sh name:"-DFLAG ..."[:file.hip] ...  -- builds tron_amd/lib/libtronhip_<name>.so with extra flags for one
# kernel file (default tron_grid_binned.hip); tools/run_var.sh / tools/ab.sh time the variants on one GPU box (kernel experiments)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
set -e
cd "$(dirname "$0")/.."
make -j8 >/dev/null
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; flags=${rest%%:*}; file=tron_grid_binned.hip
  if [[ "$rest" == *:* ]]; then file=${rest#*:}; fi
  base=${file%.hip}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -Iinclude --offload-arch=gfx950 $flags \
      -c tron_amd/csrc/$file -o build/var_${base}_$name.o
  objs=$(ls build/*.o | grep -v "tron_main.o\|build/var_\|build/$base.o" | tr '\n' ' ')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/var_${base}_$name.o -o tron_amd/lib/libtronhip_$name.so \
      -L/opt/rocm/lib -lrocfft -lamdhip64 -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=tron_amd/csrc/exports.map
  echo "built libtronhip_$name.so ($file: $flags)"
done
