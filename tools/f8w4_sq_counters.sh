#!/bin/bash
# GPU box: SQ counters of the fp8 self-attention launch (one-wave-per-SIMD kernel), exp formulation ("exp") or exp-free P ("lin").
#   tools/f8w4_sq_counters.sh exp lin
set -o pipefail
cd "$(dirname "$0")/.."
for m in "$@"; do
  unset LX2V_ATTN_FP8_P; [ "$m" = lin ] && export LX2V_ATTN_FP8_P=linear
  d=gpurun_out/sq_f8_$m
  mkdir -p $d
  LD_LIBRARY_PATH=lightx2v_amd timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $d -o p -- ./build/kernel_harness selfattn8 75600 40 1 > $d.log 2>&1 || { echo "run failed"; tail -5 $d.log; exit 1; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$m" "$f" <<'PY'
import csv, sys, collections
tag, path = sys.argv[1:3]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "attn_fwd_fp8" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: float(f"{sum(v) / len(v):.4g}") for k, v in sorted(acc.items())})
PY
done
