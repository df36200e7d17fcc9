#!/bin/bash
# GPU box: L2 hit rate and cycles of the self-attention launch for "<heads>[:w8]" arguments.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/l2
for a in "$@"; do
  h=${a%%:*}; k=""; [[ "$a" == *:w8 ]] && k=w8
  LX2V_ATTN_KERNEL=$k timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/l2/$a -o p -- ./build/kernel_harness selfattn 75600 $h 2 > gpurun_out/l2/$a.log 2>&1 || { echo "$a failed"; exit 1; }
  python3 - "$a" gpurun_out/l2/$a <<'PY'
import csv, glob, sys
a, d = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "attn_fwd" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"{a:>8}: L2 hit {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}  misses {m['TCC_MISS_sum'] * 128 / 1e9:.1f} GB  cycles {m['GRBM_GUI_ACTIVE'] / 8 / 1e6:.1f} M")
PY
done
