#!/bin/bash
# usage (on the GPU box): tools/pmcx.sh <outdir> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <python script + args...>
# ad-hoc counter passes (one rocprofv3 --pmc run each) with per-kernel sums; `rocprofv3 --list-avail` names the counters
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
passes=(); while [ "$1" != "--" ]; do passes+=("$1"); shift; done; shift
i=0
for pass in "${passes[@]}"; do
  i=$((i+1)); rocprofv3 --pmc $pass --kernel-trace -d $out/p$i --output-format csv -- python3 $R/"$@" > $out/p$i.log 2>&1
done
python3 - <<PY | tee $out/summary.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$out/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if any(t in k for t in ("grid","post","fft","pre")) and 'warm' not in k:
        print(k); print('   '+'  '.join(f"{n}={val:.4g}" for n,val in sorted(v.items())))
PY
