#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_step.sh <tag> [model]
# rocprofv3 kernel-trace + stats of the bench step, single-stream and two-stream, with the per-step summaries and the idle-gap report
tag=${1:-prof}
model=${2:-beit_base_patch16_224}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ss -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-input-staging --single-stream --model $model > $out/ss.json 2> $out/ss.err || { tail -5 $out/ss.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ds -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-input-staging --no-alone --model $model > $out/ds.json 2> $out/ds.err || { tail -5 $out/ds.err; exit 1; }
cd $root
python tools/prof_summary.py $out/ss | tee $out/kernel_stats_singlestream.txt
python tools/prof_summary.py $out/ds | tee $out/kernel_stats_dualstream.txt
python tools/trace_gaps.py $out/ss | tee $out/gaps_singlestream.txt
python tools/trace_gaps.py $out/ds | tee $out/gaps_dualstream.txt
