#!/bin/bash
# usage (GPU box): tools/traffic.sh <outdir> <script + args>  -- FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots);
# writes gpurun_out/<outdir>/traffic.json stamped with the kernel-source hash (copy it to profiles/traffic_current.json)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c --output-format csv -- python3 $R/"$@" > $out/$c.log 2>&1
done
CMD="$*" OUT=$out python3 - <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tron_amd.buildinfo import kernel_source_hash
out = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].split('::')[-1]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
res = {"source_hash": kernel_source_hash(), "command": os.environ["CMD"],
       "coil_slices_per_launch": int(os.environ.get("COIL_SLICES_PER_LAUNCH", "512")),   # gridbench 8 coils x 64-slice launches
       "units": "KiB summed over dispatches; gfx950: FETCH_SIZE counts half the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM)",
       "kernels": {}}
for k, v in agg.items():
    if any(t in k for t in ('grid', 'fft', 'post', 'pre', 'reduce')) and 'warm' not in k:
        res["kernels"][k] = {c: {"kib": val, "dispatches": len(n[(k, c)])} for c, val in v.items()}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
