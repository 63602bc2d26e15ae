#!/bin/bash
# usage (on the GPU box): bash tools/pmc_attn.sh <tag>     SQ counters of the attention kernels (tools/bench_attn.py), one --pmc pass per group
tag=${1:-pmcattn}
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
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python $root/tools/bench_attn.py > $out/g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/g$i.log; }
done
cd $root
python3 - <<PY | tee $out/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag in ("attn2_bwd_fused", "attn2_fwd", "attn2_dbias", "attn_bwd_fused_kernel", "attn_fwd_kernel", "attn_dbias"):
            if tag in k:
                acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag, d in acc.items():
    print(tag)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  ({len(v)} dispatches)")
PY
