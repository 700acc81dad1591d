#!/bin/bash
# usage (GPU box): tools/traffic.sh <key> [bench.py arguments]  -- HBM traffic of one bench.py workload: FETCH_SIZE and WRITE_SIZE
# in separate rocprofv3 passes (TCC slots) over `bench.py --steps 2 --warmup 1 <arguments>`, the very launches bench.py times.
# Writes gpurun_out/traffic/traffic_<key>.json stamped with the kernel-source hash and the launch size; copy it to profiles/.
# <key> must be bench.py's workload_key() of the arguments (bench.py prints the one it looks for in roofline.traffic_source).
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
R=$GRAFT_REPO_ROOT; key=$1; shift; out=$R/gpurun_out/traffic/$key; mkdir -p $out; cd /tmp; export TMPDIR=/tmp TRON_BENCH_NO_BURN_IN=1   # (no burn-in child under the profiler)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --sustain 0 --cpu-slices 0 --no-irt --no-check --no-fresh --no-one-coil "$@" > $out/$c.log 2> $out/$c.err
done
CMD="bench.py --steps 6 --warmup 2 $*" OUT=$out KEY=$key python3 - <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tron_amd.buildinfo import kernel_source_hash
out = os.environ["OUT"]
line = [l for l in open(out + "/FETCH_SIZE.log") if l.startswith("{")][-1]
roof = json.loads(line)["roofline"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].split('::')[-1]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
res = {"source_hash": kernel_source_hash(), "command": os.environ["CMD"], "units_per_launch": roof["units_per_launch"],
       "units": "KiB summed over dispatches; gfx950: FETCH_SIZE counts half the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM)",
       "kernels": {}}
for k, v in agg.items():
    if any(t in k for t in ('grid', 'fft', 'post', 'pre', 'reduce')) and 'warm' not in k and 'prep' not in k:
        res["kernels"][k] = {c: {"kib": val, "dispatches": len(n[(k, c)])} for c, val in v.items()}
path = os.path.join(os.path.dirname(out), f"traffic_{os.environ['KEY']}.json")
json.dump(res, open(path, "w"), indent=1)
print(path, json.dumps({k: {c: round(x["kib"] / x["dispatches"] / 1024, 1) for c, x in v.items()} for k, v in res["kernels"].items()}))
PY
