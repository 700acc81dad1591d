# usage (GPU box): bash tools/arc_skip.sh -- VALU instructions and time of the arc kernel with phases compiled out (variants skipi/skipo/skipd)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for v in orig skipi skipo skipd; do
  if [ $v = orig ]; then cp /tmp/orig.so tron_amd/lib/libtronhip.so; else cp tron_amd/lib/libtronhip_$v.so tron_amd/lib/libtronhip.so; fi
  echo "== $v"; python tools/gridbench.py 8 64 fast 5 2>&1 | tail -1
  bash tools/pmcx.sh arc_skip_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU" -- tools/gridbench.py 8 64 fast 3 > /dev/null 2>&1
  grep -A1 "grid_arc_kernel" gpurun_out/arc_skip_$v/summary.txt
done
cp /tmp/orig.so tron_amd/lib/libtronhip.so
cp tron_amd/lib/libtronhip_aprof.so tron_amd/lib/libtronhip.so; python tools/arcprof.py 8 64 2>&1 | tail -12; cp /tmp/orig.so tron_amd/lib/libtronhip.so
