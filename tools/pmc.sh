#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <outdir> <python script + args...>   -- collects two SQ counter passes
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rocprofv3 --pmc $pass --kernel-trace -d $out/p$i --output-format csv -- python3 $R/"$@" > $out/p$i.log 2>&1
done
python3 - <<PY | tee $out/summary.txt
import csv,glob,collections,sys
sys.path.insert(0,'$R')
from tron_amd.buildinfo import kernel_source_hash
print('kernel sources', kernel_source_hash(), '| command: $*')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); calls=collections.Counter()
for f in glob.glob('$out/p*/*/*_counter_collection.csv'):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if any(t in k for t in ("grid","post","fft","pre")):
        w=v.get('SQ_WAVES',1)
        print(k); print('   '+'  '.join(f"{n}={val:.3g}" for n,val in sorted(v.items())))
        if 'SQ_INSTS_VALU' in v: print(f"   per wave: VALU {v['SQ_INSTS_VALU']/w:.0f} LDS {v['SQ_INSTS_LDS']/w:.0f} SALU {v['SQ_INSTS_SALU']/w:.0f} wave_cycles(quad) {v['SQ_WAVE_CYCLES']/w:.0f}  lane_util {v.get('SQ_THREAD_CYCLES_VALU',0)/max(v.get('SQ_ACTIVE_INST_VALU',1),1)/64:.2f}")
PY
