#!/bin/bash
# GPU box: cycles (GRBM_GUI_ACTIVE) and wall time of the self-attention launch for each ablation
# library built by build_x16_ablations.sh ("0" = the product library with the 16x16x32 kernel, "w4" = with the 32x32x16 one).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abl16
for m in "$@"; do
  lib=build/x16_abl$m; [ "$m" = 0 ] && lib=lightx2v_amd; export LX2V_ATTN_KERNEL=x16; [ "$m" = w4 ] && { lib=lightx2v_amd; export LX2V_ATTN_KERNEL=w4; }
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/abl16/m$m -o p -- ./build/kernel_harness selfattn 75600 40 2 > gpurun_out/abl16/m$m.log 2>&1 || { echo "mask $m: run failed"; exit 1; }
  f=$(find gpurun_out/abl16/m$m -name "*counter_collection.csv" | head -1)
  python3 - "$m" "$f" gpurun_out/abl16/m$m.log <<'PY'
import csv, sys, collections
m, f, log = sys.argv[1:4]
by = collections.defaultdict(list)   # a self-attention call may be two launches (main + tail): one mean per symbol, summed
for r in csv.DictReader(open(f)):
    if "attn_fwd" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
cyc = sum(sum(v) / len(v) for v in by.values()) / 8
ms = [l.split(":")[1].split("ms")[0].strip() for l in open(log) if l.startswith("attn ")]
print(f"mask {m:>2}: GRBM cycles per call {cyc:.0f} ({len(by)} kernel(s))  harness {ms}")
PY
done
