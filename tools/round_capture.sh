#!/bin/bash
# usage (GPU box): bash tools/round_capture.sh <tag>  -- the evidence set of a round: bench.py lines of the default and secondary
# workloads, rocprofv3 --kernel-trace --stats of the default bench command, per-workload HBM traffic captures (tools/traffic.sh).
# Everything lands under gpurun_out/<tag>/ and gpurun_out/traffic/; copy what is to be judged into profiles/.
export TRON_TUNING=1   # the library reads TRON_* switches only under TRON_TUNING=1
tag=${1:-round}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
b() { name=$1; shift; python bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err; tail -c 600 $out/bench_$name.json | head -c 300; echo; }
# traffic first: the bench lines that follow find a fresh capture
bash tools/traffic.sh nc8_npe402_nz256 > $out/traffic_default.log 2>&1
cp gpurun_out/traffic/traffic_*.json profiles/ 2>/dev/null
b default
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $R/bench.py --cpu-slices 0 --no-irt --no-check > $out/stats.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && TRON_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -d $out/stats_one_lane --output-format csv -- python3 $R/bench.py --cpu-slices 0 --no-irt --no-check > $out/stats_one_lane.log 2>&1 )
find $out/stats $out/stats_one_lane -name "*kernel_stats.csv" | head
