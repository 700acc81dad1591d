#!/bin/bash
# usage (GPU box): bash tools/kstat.sh <script.py> [args...]  -- per-kernel average durations (rocprofv3 --kernel-trace --stats), old library then new
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cp $R/tron_amd/lib/libtronhip.so /tmp/new.so
for v in old new; do
  if [ $v = old ]; then cp $R/tron_amd/lib/libtronhip_old.so $R/tron_amd/lib/libtronhip.so; else cp /tmp/new.so $R/tron_amd/lib/libtronhip.so; fi
  rm -rf /tmp/ks_$v; rocprofv3 --kernel-trace --stats -d /tmp/ks_$v --output-format csv -- python3 $R/"$@" > /tmp/ks_$v.log 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
for f in glob.glob('/tmp/ks_$v/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:7]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
done
