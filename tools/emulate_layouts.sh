#!/bin/bash
# One rank's compute under each (cfg_p x seq_p) layout of the headline workload, on ONE GPU
# (bench.py --as-rank-of: exchanges replaced by device copies of the rank's own data), next to the plain
# single-GPU step and the sequence-parallel path on an RCCL group of one rank (--sp1).
# A single-GPU emulation, not a scaling measurement.   usage: tools/emulate_layouts.sh [out_dir] [extra bench.py args]
set -u
out=${1:-gpurun_out/r03_layouts}
shift || true
mkdir -p "$out"
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/$name.json" 2> "$out/$name.err" || { echo "$name failed (rc $?)"; tail -5 "$out/$name.err"; return 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
c = r.get("composition_ms_per_step", {})
print(f"{sys.argv[2]:>8}: {r['value']:.3f} s/step  self-attn {r['roofline']['achieved']:.0f} TFLOP/s ({r['roofline']['launch_ms']:.2f} ms x {r['roofline']['launches_timed']})  "
      f"linears {r.get('linears_bf16', {}).get('achieved_tflops', 0):.0f} TFLOP/s  chunks {r.get('ulysses_chunk_heads')}  composition {c}", flush=True)
PY
}
run 1x1 "$@" && run sp1 --sp1 "$@" && run 2x1 --as-rank-of 2x1 "$@" && run 2x2 --as-rank-of 2x2 "$@" && run 2x4 --as-rank-of 2x4 "$@" && run 1x8 --as-rank-of 1x8 "$@"
