#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/pmc_lists.sh <tag>
# HBM-side bytes per launch of the LayerNorm kernels in the DEFAULT step (drop-path sample lists on): the dense kernels beside the keep-list ones.
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains beside --pmc), as tools/pmc_run.sh.
tag=${1:-pmc_lists}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-input-staging --single-stream > $out/$c.log 2>&1 || { echo "pmc pass $c failed"; tail -5 $out/$c.log; exit 1; }
done
cd $root
for k in ln_fwd_keep ln_bwd_keep "ln_fwd_kernel" "ln_bwd_kernel" attn_fwd attn_bwd_fused "gemm_tn256_group" token_bwd transpose_batch mask_compact droppath_lists; do
  python tools/pmc_summary.py $out/FETCH_SIZE "$k" | sed 's/^.*counter_collection.csv: /fetch: /'
  python tools/pmc_summary.py $out/WRITE_SIZE "$k" | sed 's/^.*counter_collection.csv: /write: /'
done | tee $out/summary.txt
