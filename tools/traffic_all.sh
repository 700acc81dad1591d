# usage (GPU box): bash tools/traffic_all.sh  -- refreshes every profiles/traffic_<key>.json bench.py looks for (after any edit under tron_amd/csrc: the captures are stamped with the source hash); copy gpurun_out/traffic/traffic_*.json to profiles/ afterwards
export TRON_TUNING=1
R=$GRAFT_REPO_ROOT; cd $R
t() { key=$1; shift; bash tools/traffic.sh $key "$@" > /dev/null 2>&1; }
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
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/traffic/traffic_*.json')):
    d=json.load(open(f)); print(f.split('traffic_')[-1], d['source_hash'], d['units_per_launch'])"
