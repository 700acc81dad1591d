# usage (GPU box): bash tools/cgnr_profile.sh <tag>  -- rocprofv3 kernel stats of a CGNR run (-i 5) and of a Walsh-combined adjoint, metric shape
export TRON_TUNING=1
tag=${1:-cgnr}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/cgnrbench.py cgnr 8 16 5 3 | tee $out/cgnr_time.log
python3 $R/tools/cgnrbench.py walsh 8 16 1 3 | tee $out/walsh_time.log
rocprofv3 --kernel-trace --stats -d $out/cgnr --output-format csv -- python3 $R/tools/cgnrbench.py cgnr 8 16 5 2 > $out/cgnr.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/walsh --output-format csv -- python3 $R/tools/cgnrbench.py walsh 8 16 1 2 > $out/walsh.log 2>&1
for k in cgnr walsh; do f=$(find $out/$k -name "*kernel_stats.csv" | head -1); echo "== $k"; cut -d, -f1-4 $f | head -16 | cut -c1-140; cp $f $out/${k}_kernel_stats.csv; done
