#!/bin/bash
# usage (GPU box): bash tools/fetch_size_units.sh  -- what FETCH_SIZE / WRITE_SIZE count per byte moved, by access width: a 1 GiB
# device copy with 4-, 8- and 16-byte accesses per lane (tools/probe/copywidth_main) under rocprofv3 --pmc, one counter per pass.
# MI355X_MICROARCH.md says FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming reads on gfx950; bench.py applies that x2
# to every kernel, so the factor has to be known for the 8-byte loads of the one-coil kernels too (round-4 review).
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/fetch_units; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace -d $out/$c --output-format csv -- $R/tools/probe/copywidth_main > $out/$c.log 2>&1; done
python3 - <<PY | tee $out/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
print("1 GiB read + 1 GiB written per dispatch; counters in KiB as rocprofv3 reports them")
for k, v in sorted(agg.items()):
    if 'copy_k' in k or 'line' in k:
        print(f"{k:40s} " + "  ".join(f"{c} = {sum(x)/len(x)/1048576:.3f} GiB per dispatch ({len(x)} dispatches)" for c, x in sorted(v.items())))
PY
