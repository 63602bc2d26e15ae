#!/bin/bash
# A/B of an environment switch over the four bench models on ONE box: tools/ab_models.sh "ENV=a" "ENV=b"   (ms_per_step, 10 steps)
for m in "--model beit_base_patch16_224" "--model dist_beit_base_patch16_224" "--model beit_large_patch16_224 --batch 64" "--model dist_beit_large_patch16_224 --batch 64"; do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alone --no-input-staging $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "[$m] [$cfg] ms_per_step img/s: $ms"
  done
done
