#!/bin/bash
# usage (GPU box): tools/pmcx.sh <outdir> "<counter list pass 1>;<pass 2>;..." <script + args>   -- generic PMC passes, per-kernel sums
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; passes=$2; shift 2; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
IFS=';' read -ra P <<< "$passes"
for pass in "${P[@]}"; do
  i=$((i+1)); rocprofv3 --pmc $pass --kernel-trace -d $out/p$i --output-format csv -- python3 $R/"$@" > $out/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$out/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-44:]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'warm' in k: continue
    print(k); print('   '+'  '.join(f"{n}={val:.4g}" for n,val in sorted(v.items())))
PY
