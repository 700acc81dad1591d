# usage (GPU box): bash tools/ab_env.sh VAR=value [rounds] [script args...]  -- interleaved runs of a bench script with and without one tuning variable
export TRON_TUNING=1
V=$1; R=${2:-3}; S=${3:-tools/fwdbench.py}; shift 3; A=${@:-8 64 fast}
for r in $(seq $R); do
  echo -n "default : "; python $S $A 2>&1 | grep -v "^W\|amdgpu" | tail -1
  echo -n "$V : "; env $V python $S $A 2>&1 | grep -v "^W\|amdgpu" | tail -1
done
