export TRON_TUNING=1
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/round5d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp TRON_BENCH_NO_BURN_IN=1
rocprofv3 --kernel-trace --stats -d $out/stats_forward --output-format csv -- python3 $R/bench.py --cpu-slices 0 --no-irt --no-check --forward > $out/stats_forward.log 2>&1
cd $R
best=$(for f in $(find $out/stats_forward -name "*kernel_stats.csv"); do echo "$(awk -F, 'NR>1{gsub(/"/,"",$2); s+=$2} END{print s+0}' $f) $f"; done | sort -n | tail -1 | cut -d' ' -f2)
cp $best $out/stats_forward.csv
WARM=5 bash tools/pmc.sh round5d/sqf tools/fwdbench.py 8 64 fast > /dev/null 2>&1; cp gpurun_out/round5d/sqf/summary.txt $out/forward_sq_counters.txt
rm -rf $out/stats_forward gpurun_out/round5d/sqf
head -5 $out/stats_forward.csv | cut -c1-160; head -30 $out/forward_sq_counters.txt
