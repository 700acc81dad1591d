#!/bin/bash
# usage (GPU box): tools/traffic.sh <outdir> <script + args>  -- FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c --output-format csv -- python3 $R/"$@" > $out/$c.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob('$out/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-44:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])].add(r['Dispatch_Id'])
for k,v in agg.items():
    if any(t in k for t in ('grid','fft','post','pre')):
        print(k, {c: (val, len(n[(k,c)])) for c,val in v.items()})
PY
