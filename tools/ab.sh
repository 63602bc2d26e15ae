#!/bin/bash
# A/B of bench.py configurations inside ONE process sequence on ONE box: tools/ab.sh "ENV1=.. ENV2=.." "ENV.." ...
# each configuration is run twice, interleaved; prints ms_per_step
for rep in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep [$cfg] ms_per_step $ms"
  done
done
