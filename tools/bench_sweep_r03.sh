#!/bin/bash
# Every bench.py configuration of the report table, back to back on one box (so that the rows compare:
# boxes differ by a few per cent in the clock they hold).  Output: gpurun_out/r03/sweep/<tag>.json, one line each.
#   usage: tools/bench_sweep_r03.sh [tags...]   (default: all)
set -o pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/r03/sweep
mkdir -p $out
run() {
  tag=$1; shift
  timeout -k 10 500 python -u bench.py "$@" > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; tail -5 $out/$tag.err; return 1; }
  python3 - "$out/$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
lin = d.get("linears_w8a8") or d.get("linears_bf16") or {}
cb = d.get("cpu_baseline", {})
print(f"{sys.argv[2]:>12}: {d['value']:.3f} s/step; attn {d['roofline']['launch_ms']:.2f} ms {d['roofline']['achieved']:.0f} TF ({d['roofline']['frac']:.3f}); linears {lin.get('achieved_tflops', 0):.0f} TF"
      + (f"; cpu {cb.get('value', 0):.0f} s/step extrapolated" if cb else "") + (f"; cpu config-1 block {cb['c1_block_s']:.2f} s" if "c1_block_s" in cb else ""), flush=True)
PY
}
N=--no-cpu-baseline
declare -A CFG=(
  [default]="" [i2v14b]="$N --model wan2.1-i2v-14b" [linfp8]="$N --linear fp8" [linint8]="$N --linear int8" [attnfp8]="$N --attn fp8" [allfp8]="$N --attn fp8 --linear fp8"
  [attnfp8lin]="$N --attn fp8-linear" [allfp8lin]="$N --attn fp8-linear --linear fp8" [a14b]="$N --model wan2.2-i2v-a14b" [resfp32]="$N --residual fp32"
  [hy]="$N --model hunyuan-1.5 --steps 4" [hyfp8]="$N --model hunyuan-1.5 --steps 4 --attn fp8 --linear fp8" [hyfp8lin]="$N --model hunyuan-1.5 --steps 4 --attn fp8-linear --linear fp8"
  [w13b]="--model wan2.1-t2v-1.3b --res 480p --steps 10 --warmup 2" [sp1]="$N --sp1"
)
tags=("$@")
[ ${#tags[@]} -eq 0 ] && tags=(default i2v14b linfp8 linint8 attnfp8 allfp8 attnfp8lin allfp8lin a14b resfp32 hy hyfp8 hyfp8lin w13b sp1)
for t in "${tags[@]}"; do run $t ${CFG[$t]} || exit 1; done
