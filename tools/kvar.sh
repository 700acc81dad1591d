#!/bin/bash
# usage (GPU box): bash tools/kvar.sh <kernel-name-pattern> <script.py> [args...]  -- average duration of the kernels matching the pattern
# (rocprofv3 --kernel-trace --stats) under the default library and every tron_amd/lib/libtronhip_*.so variant (tools/build_variants.sh)
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
pat=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
cp $R/tron_amd/lib/libtronhip.so /tmp/orig.so
for f in /tmp/orig.so $R/tron_amd/lib/libtronhip_*.so; do
  cp $f $R/tron_amd/lib/libtronhip.so
  rm -rf /tmp/kv; rocprofv3 --kernel-trace --stats -d /tmp/kv --output-format csv -- python3 $R/"$@" > /tmp/kv.log 2>&1
  python3 - "$pat" "$(basename $f)" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/kv/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Name']:
            print(f"{sys.argv[2]:28s} {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
done
cp /tmp/orig.so $R/tron_amd/lib/libtronhip.so
