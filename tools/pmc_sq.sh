#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sq.sh <tag> [model]    SQ counters of the step's main kernels (single-stream bench run), one --pmc pass per group
tag=${1:-pmcsq}
model=${2:-beit_base_patch16_224}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-input-staging --single-stream --model $model > $out/g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/g$i.log; }
done
cd $root
python3 - <<PY | tee $out/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
tags = ("gemm_tn256_group", "gemm_nt256_kernel<3, 5", "gemm_nt256_kernel<0, 5", "gemm_nt256_kernel<1, 4", "gemm_nt256_kernel<8, 4", "gemm_nt256_kernel<9, 4",
        "gemm_nt256_kernel<2, 4", "attn2_bwd_fused", "attn2_fwd", "attn_bwd_fused", "attn_fwd", "ln_bwd", "ln_fwd")
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in tags:
            if tag in k:
                acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag in tags:
    d = acc.get(tag)
    if not d: continue
    print(tag)
    g = lambda c: sum(d[c]) / len(d[c]) if c in d else float("nan")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  ({len(v)} dispatches)")
    # MFMA utilisation in [0, 1]: SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip (256 CUs x 4), GRBM_GUI_ACTIVE over the
    # 8 XCDs, so the chip's SIMD-cycles of the dispatch are GRBM_GUI_ACTIVE / 8 x 1024 = GRBM_GUI_ACTIVE x 128
    print(f"   -> MFMA utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE x 128): {g('SQ_VALU_MFMA_BUSY_CYCLES') / (128 * g('GRBM_GUI_ACTIVE')):.3f}   LDS active / BUSY: {g('SQ_ACTIVE_INST_LDS') / g('SQ_BUSY_CYCLES'):.3f}"
          f"   bank conflict / LDS idx active: {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f}   wait_any / wave_cycles: {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
PY
