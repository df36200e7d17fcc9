set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/sweep2
run() { tag=$1; shift; timeout -k 10 400 python -u bench.py "$@" > gpurun_out/r02/sweep2/$tag.json 2> gpurun_out/r02/sweep2/$tag.err || { echo "$tag failed"; tail -5 gpurun_out/r02/sweep2/$tag.err; return 1; }; python3 -c "
import json,sys; d=json.load(open('gpurun_out/r02/sweep2/$tag.json')); print('$tag', round(d['value'],3), 's/step; attn', round(d['roofline']['launch_ms'],2), 'ms', round(d['roofline']['achieved'],0), 'TF', round(d['roofline']['frac'],3))"; }
N=--no-cpu-baseline
run default $N && run attnfp8 $N --attn fp8 && run attnfp8lin $N --attn fp8-linear && run allfp8 $N --attn fp8 --linear fp8 && run allfp8lin $N --attn fp8-linear --linear fp8 && run hyfp8 $N --model hunyuan-1.5 --steps 4 --attn fp8 --linear fp8 && run hyfp8lin $N --model hunyuan-1.5 --steps 4 --attn fp8-linear --linear fp8
