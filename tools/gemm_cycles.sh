#!/bin/bash
# GPU box: cycles and wall time of the [S,5120]x[5120,5120] GEMM for library dirs given as arguments.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/gemm
for lib in "$@"; do
  tag=$(basename $lib)
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gemm/$tag -o p -- ./build/kernel_harness gemm1 75600 ${GEMM_H:-40} 3 > gpurun_out/gemm/$tag.log 2>&1 || { echo "$tag failed"; exit 1; }
  python3 - "$tag" gpurun_out/gemm/$tag <<'PY'
import csv, glob, sys
tag, d = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
cyc = [float(r["Counter_Value"]) / 8 for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(f"{tag:>12}: cycles {sum(cyc)/len(cyc)/1e6:.2f} M  dur {sum(dur)/len(dur):.3f} ms  clock {sum(cyc)/sum(dur)/1e6:.2f} GHz (n={len(rows)})")
PY
done
