#!/bin/bash
# GPU box: which hipBLASLt/rocBLAS kernel torch.matmul runs at the 14B Linear shapes, its duration and busy cycles.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/blas
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/blas/p -o p -- python3 tools/gemm_vs_blas.py 3 > gpurun_out/blas/run.log 2>&1 || { echo failed; tail -5 gpurun_out/blas/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/blas/p/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" in n.lower() or "cijk" in n.lower():
        agg[n].append((float(r["Counter_Value"]) / 8, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size")))
for n, v in agg.items():
    cyc = sum(x[0] for x in v) / len(v); dur = sum(x[1] for x in v) / len(v)
    print(f"{n[:230]}\n    n={len(v)} cycles {cyc/1e6:.2f} M dur {dur:.3f} ms clock {cyc/dur/1e6:.2f} GHz vgpr {v[0][2]} agpr {v[0][3]} lds {v[0][4]} wg {v[0][5]} grid {v[0][6]}")
PY
