#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench command (python bench.py: 2 timed steps + 1 warm-up, VAE decode, cpu_baseline) ->
# gpurun_out/trace_bench_default_summary.md + the JSON line that run printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_bench_default
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_bench_default -- python3 bench.py > gpurun_out/trace_bench_default.json 2> gpurun_out/trace_bench_default.err || { tail -5 gpurun_out/trace_bench_default.err; exit 1; }
f=$(ls gpurun_out/trace_bench_default/*/*_kernel_trace.csv | head -1)
python3 tools/summarize_trace.py $f gpurun_out/trace_bench_default_summary.md | head -12
cut -c1-300 gpurun_out/trace_bench_default.json
