#!/bin/bash
# Timing ablations of the persistent bf16 GEMM kernel (W4_ABL masks, see gemm_bf16_w4.hip).
#   tools/gemm_w4_ablations.sh build 0 1 2 ...   (CPU box: build/w4abl_<mask>/liblx2v_hip.so)
#   tools/gemm_w4_ablations.sh run 0 1 2 ...     (GPU box: cycles / duration / clock of [75600,5120]x[5120,5120])
set -o pipefail
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  cd lightx2v_amd/csrc
  for m in "$@"; do
    d=../../build/w4abl_$m; mkdir -p $d
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DW4_ABL=${m%%_*} ${W4_EXTRA} -x hip -c gemm_bf16_w4.hip -o $d/w4.o || exit 1
    objs=$(ls ../../build/obj/*.o | grep -v gemm_bf16_w4)
    hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so $objs $d/w4.o && rm $d/w4.o
  done
  exit 0
fi
mkdir -p gpurun_out/w4abl
for m in "$@"; do
  lib=build/w4abl_$m
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/w4abl/$m -o p -- ./build/kernel_harness gemm1 75600 ${GEMM_H:-40} 5 > gpurun_out/w4abl/$m.log 2>&1 || { echo "mask $m failed"; tail -3 gpurun_out/w4abl/$m.log; exit 1; }
  python3 - "$m" gpurun_out/w4abl/$m <<'PY'
import csv, glob, sys
tag, d = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
cyc = [float(r["Counter_Value"]) / 8 for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(f"mask {tag:>4}: cycles {sum(cyc)/len(cyc)/1e6:.2f} M  dur {sum(dur)/len(dur):.3f} ms  clock {sum(cyc)/sum(dur)/1e6:.2f} GHz (n={len(rows)})", flush=True)
PY
done
