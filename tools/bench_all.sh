#!/bin/bash
# the four bench lines of profiles/round4_bench_*.json (on the GPU box, via gpurun): default run of the headline, short runs of the others
out=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-bench_all}
mkdir -p $out
python bench.py > $out/bench_base.json 2> $out/bench_base.err || { tail -3 $out/bench_base.err; exit 1; }
python bench.py --model dist_beit_base_patch16_224 --no-cpu-baseline > $out/bench_dist_beit_base.json 2> $out/b2.err || { tail -3 $out/b2.err; exit 1; }
python bench.py --model beit_large_patch16_224 --batch 64 --no-cpu-baseline > $out/bench_beit_large.json 2> $out/b3.err || { tail -3 $out/b3.err; exit 1; }
python bench.py --model dist_beit_large_patch16_224 --batch 64 --no-cpu-baseline > $out/bench_dist_beit_large.json 2> $out/b4.err || { tail -3 $out/b4.err; exit 1; }
for f in base dist_beit_base beit_large dist_beit_large; do python3 -c "import json,sys; d=json.load(open('$out/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['config'].get('value_all_rows'), d['config']['step_mfma_frac'])"; done
