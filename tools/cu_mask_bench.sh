#!/bin/bash
# usage (on the GPU box): bash tools/cu_mask_bench.sh <tag>
# The step with some CUs taken away (ROC_GLOBAL_CU_MASK), as RCCL's channel workgroups do during a bucket's all-reduce:
# a launch sized to fill all 256 CUs in exactly one round then needs a second, nearly empty one.
tag=${1:-cumask}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd $root
# UVIT_STREAM_MODE=1: the second stream at the caller's priority -- the mode data-parallel runs use (a single-GPU run picks the lower-priority
# second stream, which is faster on 256 CUs and slower with CUs taken away: DESIGN.md section 6)
export UVIT_STREAM_MODE=${UVIT_STREAM_MODE:-1}
for n in 256 240 224; do
  mask=0x$(python3 -c "print('f' * ($n // 4))")
  ROC_GLOBAL_CU_MASK=$mask timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-input-staging > $out/b$n.json 2> $out/b$n.err || { tail -3 $out/b$n.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/b$n.json')); print('$n CUs: %.2f ms/step  %.0f img/s' % (d['ms_per_step'], d['value']))" | tee -a $out/summary.txt
done
