#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_check.sh <tag> [pytest-args]
# runs the GPU parity tests, then bench.py, then a rocprofv3 kernel-stats pass; outputs under gpurun_out/<tag>*
tag=${1:-run}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 -x "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED"; grep -E "^E  |Error|assert" gpurun_out/${tag}_tests.log | head -30; exit 1; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-input-staging --no-alone > gpurun_out/${tag}_prof.json 2> gpurun_out/${tag}_prof.err
python tools/prof_summary.py gpurun_out/${tag}_prof | tee gpurun_out/${tag}_prof_summary.txt
