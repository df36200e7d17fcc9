#!/bin/bash
# rocprofv3 kernel statistics of the default bench (and of the fp8-attention bench) on the GPU box:
#   tools/profile_bench.sh   -> gpurun_out/r02/prof/{bf16,fp8,fp8lin}_kernel_stats.csv + the bench lines printed under the profiler
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -o bf16 -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02/prof/bf16_bench_line.json 2> $R/gpurun_out/r02/prof/bf16.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fp8 -o fp8 -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --attn fp8 > $R/gpurun_out/r02/prof/fp8_bench_line.json 2> $R/gpurun_out/r02/prof/fp8.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fp8lin -o fp8lin -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --attn fp8-linear > $R/gpurun_out/r02/prof/fp8lin_bench_line.json 2> $R/gpurun_out/r02/prof/fp8lin.err || exit 1
find /tmp/prof_bf16 /tmp/prof_fp8 /tmp/prof_fp8lin -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02/prof/ \;
ls -la $R/gpurun_out/r02/prof
