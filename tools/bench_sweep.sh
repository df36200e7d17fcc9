set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/sweep
run() { tag=$1; shift; timeout -k 10 400 python -u bench.py "$@" > gpurun_out/r02/sweep/$tag.json 2> gpurun_out/r02/sweep/$tag.err || { echo "$tag failed"; tail -5 gpurun_out/r02/sweep/$tag.err; return 1; }; python3 -c "
import json,sys; d=json.load(open('gpurun_out/r02/sweep/$tag.json')); print('$tag', round(d['value'],3), 's/step; attn', round(d['roofline']['launch_ms'],2), 'ms', round(d['roofline']['achieved'],0), 'TF', round(d['roofline']['frac'],3))"; }
N=--no-cpu-baseline
run default && run i2v14b $N --model wan2.1-i2v-14b && run linfp8 $N --linear fp8 && run attnfp8 $N --attn fp8 && run allfp8 $N --attn fp8 --linear fp8 && run attnfp8lin $N --attn fp8-linear && run allfp8lin $N --attn fp8-linear --linear fp8 && run a14b $N --model wan2.2-i2v-a14b && run hy $N --model hunyuan-1.5 --steps 4 && run hyfp8 $N --model hunyuan-1.5 --steps 4 --attn fp8 --linear fp8 && run hyfp8lin $N --model hunyuan-1.5 --steps 4 --attn fp8-linear --linear fp8 && run w13b $N --model wan2.1-t2v-1.3b --res 480p --steps 10 --warmup 2 && run sp1 $N --sp1
