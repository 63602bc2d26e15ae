#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/pmc_run.sh <tag>
# HBM bytes per launch of the main kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# single-stream bench run, as MI355X_MICROARCH.md prescribes (no trace domains beside --pmc; FETCH_SIZE x2 on gfx950).
tag=${1:-pmc}
model=${2:-beit_base_patch16_224}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd /tmp
# (--all-rows below = full-size launches: no masked-row bound, every branch on every sample -- the shapes bench.py quotes per-launch traffic for)
# optional third / fourth pass (PMC_EXTRA="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum", when `rocprofv3 -L` lists them): the requests that
# reach HBM itself -- FETCH_SIZE / WRITE_SIZE count the L2's fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM)
for c in FETCH_SIZE WRITE_SIZE $PMC_EXTRA; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-input-staging --single-stream --all-rows --model $model > $out/$c.log 2>&1 || { echo "pmc pass $c failed"; tail -5 $out/$c.log; exit 1; }
done
cd $root
# one instantiation, two shapes in a fixed rotation: proj, fc2, proj, fc2, ... (single-stream --all-rows run: with the masked-row bound the
# last block's fc2 is another kernel and the rotation would flip after every forward)
for c in FETCH_SIZE WRITE_SIZE $PMC_EXTRA; do
  python tools/pmc_summary.py $out/$c "gemm_nt256_kernel<3, 5" --alternate 2 proj,fc2 | sed "s/^.*counter_collection.csv: /$c: /"
done | tee $out/summary_resid.txt
for k in "gemm_nt256_kernel<2, 4" "gemm_nt256_kernel<8, 4" "gemm_nt256_kernel<9, 4" "gemm_ntr_kernel<3, 10>" "gemm_nt256_kernel<1, 4" "gemm_nt256_kernel<0, 4" "gemm_nt256_kernel<0, 5" "gemm_nt_kernel<" "gemm_tn256_group" attn_fwd attn_bwd_fused attn_dbias_reduce attn2_fwd attn2_bwd_fused attn2_dbias_reduce ln_bwd ln_fwd adamw; do
  python tools/pmc_summary.py $out/FETCH_SIZE "$k" | sed 's/^.*counter_collection.csv: /fetch: /'
  python tools/pmc_summary.py $out/WRITE_SIZE "$k" | sed 's/^.*counter_collection.csv: /write: /'
  for c in $PMC_EXTRA; do python tools/pmc_summary.py $out/$c "$k" | sed "s/^.*counter_collection.csv: /$c: /"; done
done | tee $out/summary.txt
