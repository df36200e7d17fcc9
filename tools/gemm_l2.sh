#!/bin/bash
# GPU box: L2 hit rate, miss bytes and cycles of the [S,5120]x[5120,5120] GEMM for library dirs given as arguments.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/gemm_l2
for lib in "$@"; do
  tag=$(basename $lib)
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gemm_l2/$tag -o p -- ./build/kernel_harness gemm1 75600 40 3 > gpurun_out/gemm_l2/$tag.log 2>&1 || { echo "$tag failed"; exit 1; }
  python3 - "$tag" gpurun_out/gemm_l2/$tag <<'PY'
import csv, glob, sys
a, d = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "gemm" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"{a:>10}: L2 hit {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}  requests {(m['TCC_HIT_sum'] + m['TCC_MISS_sum']) * 128 / 1e9:.2f} GB  misses {m['TCC_MISS_sum'] * 128 / 1e9:.2f} GB  cycles {m['GRBM_GUI_ACTIVE'] / 8 / 1e6:.2f} M")
PY
done
