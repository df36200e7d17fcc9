#!/bin/bash
# rocprofv3 kernel statistics of the default bench and of the W8A8 bench on the GPU box, then the unprofiled default line:
#   tools/profile_bench_r03.sh -> gpurun_out/r03/prof/{bf16,linfp8}_kernel_stats.csv, the bench lines printed under the profiler, bench_line.json
set -o pipefail
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/r03/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -o bf16 -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $P/bf16_bench_line.json 2> $P/bf16.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_linfp8 -o linfp8 -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --linear fp8 > $P/linfp8_bench_line.json 2> $P/linfp8.err || exit 1
find /tmp/prof_bf16 /tmp/prof_linfp8 -name "*kernel_stats.csv" -exec cp {} $P/ \;
cd $R && timeout -k 10 600 python3 -u bench.py > $P/bench_line.json 2> $P/bench.err || exit 1
ls -la $P
