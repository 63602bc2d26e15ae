#!/bin/bash
# A/B with the order rotated per repetition (clock / thermal drift inside a repetition otherwise favours fixed positions):
# tools/ab_rot.sh REPS "ENV=a" "ENV=b" ...   prints ms_per_step per run and the per-configuration mean
reps=$1; shift
cfgs=("$@"); n=${#cfgs[@]}
declare -A sum
for ((r = 0; r < reps; ++r)); do
  for ((k = 0; k < n; ++k)); do
    i=$(( (k + r) % n )); cfg=${cfgs[$i]}
    ms=$(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alone --no-input-staging ${AB_ARGS} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $r [$cfg] $ms"
    sum[$i]=$(python -c "print(${sum[$i]:-0} + $ms)")
  done
done
for ((i = 0; i < n; ++i)); do echo "mean [${cfgs[$i]}] $(python -c "print(round(${sum[$i]} / $reps, 3))")"; done
