#!/bin/bash
# GPU box: cycles (GRBM_GUI_ACTIVE) and wall time of the self-attention launch for each ablation
# library built by build_w4_ablations.sh ("0" = the product library).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abl
export LX2V_ATTN_KERNEL=w4
for m in "$@"; do
  lib=build/abl_$m; [ "$m" = 0 ] && lib=lightx2v_amd
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/abl/m$m -o p -- ./build/kernel_harness selfattn 75600 40 2 > gpurun_out/abl/m$m.log 2>&1 || { echo "mask $m: run failed"; exit 1; }
  f=$(find gpurun_out/abl/m$m -name "*counter_collection.csv" | head -1)
  python3 - "$m" "$f" gpurun_out/abl/m$m.log <<'PY'
import csv, sys
m, f, log = sys.argv[1:4]
cyc = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn_fwd" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
ms = [l.split(":")[1].split("ms")[0].strip() for l in open(log) if l.startswith("attn ")]
print(f"mask {m:>2}: GRBM cycles/launch {sum(cyc)/len(cyc)/8:.0f} (n={len(cyc)})  harness {ms}")
PY
done
