#!/bin/bash
# kernel trace of the 720p x 129f tiled VAE decode -> gpurun_out/trace_vae/ + summary md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VAE_STREAMS=1      # per-kernel durations: decode the tiles one after the other (the default, two streams, co-runs kernels)
rm -rf gpurun_out/trace_vae
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_vae -- python3 tools/bench_vae.py > gpurun_out/trace_vae.log 2>&1 || { tail -5 gpurun_out/trace_vae.log; exit 1; }
f=$(ls gpurun_out/trace_vae/*/*_kernel_trace.csv | head -1)
python3 tools/summarize_trace.py $f gpurun_out/trace_vae_summary.md | head -40
