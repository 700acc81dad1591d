#!/bin/bash
# usage (GPU box): bash tools/ktrace.sh <tag> <script.py> [args...]  -- kernel timeline of one run (rocprofv3 --kernel-trace): start / end of every
# gridding-stage dispatch relative to the first, so that overlap between streams (arc kernel || inner tile) can be read off
export TRON_TUNING=1
R=$GRAFT_REPO_ROOT; tag=$1; shift; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_$tag; rocprofv3 --kernel-trace -d /tmp/kt_$tag --output-format csv -- python3 $R/"$@" > /tmp/kt_$tag.log 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob('/tmp/kt_$tag/*/*kernel_trace.csv'):
    rows+=list(csv.DictReader(open(f)))
rows=[r for r in rows if any(t in r['Kernel_Name'] for t in ('grid_','fft512'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-12:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print(f"{r['Kernel_Name'].split('(')[0][-44:]:44s} start {s:9.1f} us  end {e:9.1f} us  dur {e-s:8.1f} us")
PY
