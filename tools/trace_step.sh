#!/bin/bash
# kernel trace of ONE denoise step (bench.py --steps 1 --warmup 1, no VAE / CPU baseline) -> gpurun_out/trace_<tag>/ + summary md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/trace_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$tag -- python3 bench.py --steps 1 --warmup 1 --no-vae --no-cpu-baseline "$@" > gpurun_out/trace_$tag.json 2> gpurun_out/trace_$tag.err || { tail -5 gpurun_out/trace_$tag.err; exit 1; }
f=$(ls gpurun_out/trace_$tag/*/*_kernel_trace.csv | head -1)
python3 tools/summarize_trace.py $f gpurun_out/trace_${tag}_summary.md | head -30
