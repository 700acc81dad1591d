#!/bin/bash
# usage: tools/build_variants.sh name1:"-DFLAG ..." name2:"..."  -- builds tron_amd/lib/libtronhip_<name>.so with extra flags for
# tron_grid_binned.hip (kernel experiments; tools/run_var.sh / tools/ab.sh time them on one GPU box)
set -e
cd "$(dirname "$0")/.."
make -j8 >/dev/null
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -Iinclude --offload-arch=gfx950 $flags \
      -c tron_amd/csrc/tron_grid_binned.hip -o build/tron_grid_binned_$name.o
  objs=$(ls build/*.o | grep -v "tron_main.o\|tron_grid_binned" | tr '\n' ' ')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/tron_grid_binned_$name.o -o tron_amd/lib/libtronhip_$name.so \
      -L/opt/rocm/lib -lrocfft -lamdhip64 -Wl,-rpath,/opt/rocm/lib
  echo "built libtronhip_$name.so ($flags)"
done
