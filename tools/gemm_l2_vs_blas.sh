#!/bin/bash
# GPU box: L2 hit rate / request and miss bytes / cycles per launch of this library's GEMM and of the
# hipBLASLt kernel torch.matmul picks, at the four 14B Linear shapes (tools/gemm_vs_blas.py).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/gemm_l2b
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gemm_l2b/p -o p -- python3 tools/gemm_vs_blas.py 2 > gpurun_out/gemm_l2b/run.log 2>&1 || { echo failed; tail -5 gpurun_out/gemm_l2b/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gemm_l2b/p/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" in n.lower() or "cijk" in n.lower():
        key = ("ours" if "lx2v" in n or "gemm_bf16" in n else "blas", r["Grid_Size"], r["Dispatch_Id"])
        per.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for (who, grid, _), m in per.items():
    agg.setdefault((who, grid), []).append(m)
for (who, grid), ms in agg.items():
    avg = {k: sum(m[k] for m in ms) / len(ms) for k in ms[0]}
    h, mi = avg["TCC_HIT_sum"], avg["TCC_MISS_sum"]
    print(f"{who} grid {grid:>9}: L2 hit {h / (h + mi):.3f}  requests {(h + mi) * 128 / 1e9:6.2f} GB  misses {mi * 128 / 1e9:5.2f} GB  cycles {avg['GRBM_GUI_ACTIVE'] / 8 / 1e6:.2f} M (n={len(ms)})")
PY
