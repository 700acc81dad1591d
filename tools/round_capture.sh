#!/bin/bash
# usage (GPU box): bash tools/round_capture.sh <tag>  -- the evidence set of a round: per-workload HBM traffic captures
# (tools/traffic.sh), bench.py lines of the default and the secondary workloads, rocprofv3 --kernel-trace --stats of the default
# bench command and of the forward bench, SQ counters and the arc kernel's phase clock, the host-buffer
# and CLI measurements.  Everything lands under gpurun_out/<tag>/ (+ gpurun_out/traffic/); the caller copies it to profiles/.
export TRON_TUNING=1
tag=${1:-round}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
t() { key=$1; shift; bash tools/traffic.sh $key "$@" > $out/traffic_$key.log 2>&1; cp gpurun_out/traffic/traffic_$key.json profiles/ 2>/dev/null; }
b() { name=$1; shift; python bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err; python -c "
import json,sys
d=json.load(open('$out/bench_$name.json')); r=d['roofline']
print('$name', d['value'], d['unit'], 'alg', d.get('algorithmic_frac_of_peak'), 'kernel frac', r['frac'], 'traffic', r.get('traffic'), 'parity', d.get('parity_rel_l2_vs_oracle'))" ; }
# traffic first: the bench lines that follow find a fresh capture of their own workload
t nc8_npe402_nz256
t nc6_npe402_nz256 --coils 6
t nc4_npe402_nz256 --coils 4
t nc2_npe402_nz256 --coils 2
t nc1_npe402_nz256 --coils 1
t nc8_npe402_nz256_half --half
t nc6_npe402_nz256_half --half --coils 6
t nc8_npe804_nz256 --spokes 804
t nc8_npe804_nz32 --spokes 804 --slices 32
t nc8_npe402_nz32 --slices 32
t forward_nc8 --forward
t nc1_npe402_nz256_linear --linear --coils 1
t nc8_npe402_nz256_exact --kb exact
b default
NI="--cpu-slices 0 --no-irt --no-one-coil"
b nc6 $NI --coils 6
b nc4 $NI --coils 4
b nc2 $NI --coils 2
b nc1 $NI --coils 1
b half $NI --half
b half_nc6 $NI --half --coils 6
b half_nc1 $NI --half --coils 1
b 804spokes $NI --spokes 804
b cfg4_share $NI --spokes 804 --slices 32
b 32slices $NI --slices 32
b exact $NI --kb exact
b forward $NI --forward
b linear_nc1 $NI --linear --coils 1
TRON_GRID_KERNEL=binned python bench.py $NI > $out/bench_binned_kernel.json 2> $out/bench_binned_kernel.err
TRON_BENCH_SHARE_GPU=1 python bench.py $NI --gpus 2 --scaling strong --spokes 804 > $out/bench_cfg4_strong_2ranks_shared_gpu.json 2> $out/bench_cfg4_strong_2ranks_shared_gpu.err
# (no burn-in child under rocprofv3: it inherits the profiler and writes a <pid>_kernel_stats.csv of its own -- round 4's forward stats
#  file was the child's; and the CSV kept is the one with the most dispatches, i.e. the bench process's)
( cd /tmp && export TMPDIR=/tmp TRON_BENCH_NO_BURN_IN=1 && rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --cpu-slices 0 --no-irt --no-check > $out/stats.log 2>&1
  rocprofv3 --kernel-trace --stats -d $out/stats_forward --output-format csv -- python3 $R/bench.py --cpu-slices 0 --no-irt --no-check --forward > $out/stats_forward.log 2>&1 )
for k in stats stats_forward; do
  best=$(for f in $(find $out/$k -name "*kernel_stats.csv"); do echo "$(awk -F, 'NR>1{gsub(/"/,"",$2); s+=$2} END{print s+0}' $f) $f"; done | sort -n | tail -1 | cut -d' ' -f2)
  cp $best $out/${k}.csv
done
WARM=20 bash tools/pmc.sh $tag/sq tools/gridbench.py 8 128 fast 3 > /dev/null 2>&1; cp gpurun_out/$tag/sq/summary.txt $out/sq_counters.txt
WARM=20 bash tools/pmc.sh $tag/sq1 tools/gridbench.py 1 128 fast 3 > /dev/null 2>&1; cp gpurun_out/$tag/sq1/summary.txt $out/sq_counters_nc1.txt
WARM=5 bash tools/pmc.sh $tag/sqf tools/fwdbench.py 8 64 fast > /dev/null 2>&1; cp gpurun_out/$tag/sqf/summary.txt $out/forward_sq_counters.txt
if [ -f tron_amd/lib/libtronhip_aprof.so ]; then
  cp tron_amd/lib/libtronhip.so /tmp/orig.so; cp tron_amd/lib/libtronhip_aprof.so tron_amd/lib/libtronhip.so
  WARM=20 python tools/arcprof.py 8 128 > $out/phase_clock.log 2>&1; cp /tmp/orig.so tron_amd/lib/libtronhip.so
fi
if [ -f tron_amd/lib/libtronhip_sprof.so ]; then
  cp tron_amd/lib/libtronhip.so /tmp/orig.so; cp tron_amd/lib/libtronhip_sprof.so tron_amd/lib/libtronhip.so
  python tools/scatprof.py 1 128 > $out/scatter_phase_clock.log 2>&1; cp /tmp/orig.so tron_amd/lib/libtronhip.so
fi
if [ -f tron_amd/lib/libtronhip_cprof.so ]; then
  cp tron_amd/lib/libtronhip.so /tmp/orig.so; cp tron_amd/lib/libtronhip_cprof.so tron_amd/lib/libtronhip.so
  python tools/cenprof.py 8 128 > $out/centre_phase_clock.log 2>&1; python tools/cenprof.py 8 32 >> $out/centre_phase_clock.log 2>&1; cp /tmp/orig.so tron_amd/lib/libtronhip.so
fi
python tools/hostbench.py 8 256 > $out/hostbench.log 2>&1; python tools/hostbench.py 8 256 --half >> $out/hostbench.log 2>&1
python tools/wholebody.py /tmp > $out/wholebody_cli.log 2>&1
python tools/plantime.py 8 256 402 > $out/plan_time.log 2>&1; python tools/plantime.py 1 256 402 >> $out/plan_time.log 2>&1
python tools/fwdbench.py 8 64 fast > $out/fwdbench.log 2>&1
for n in 8 6 4 2 1; do python tools/gridbench.py $n 128 fast 5 2>&1 | tail -1; done > $out/gridbench.log
TRON_GRID_KERNEL=binned python tools/gridbench.py 8 128 fast 5 2>&1 | tail -1 >> $out/gridbench.log
bash tools/ktrace.sh $tag tools/gridbench.py 8 128 fast 3 > $out/ktrace.log 2>&1
WARM=20 python tools/gridbench.py 8 128 fast 20 2>&1 | tail -1 > $out/gridbench_warm.log
python tools/config1.py /tmp/c1 > $out/config1.json 2> $out/config1.err
rm -rf $out/stats $out/stats_forward $out/sq $out/sq1 $out/sqf gpurun_out/$tag/sq gpurun_out/$tag/sq1 gpurun_out/$tag/sqf
