#!/bin/bash
# Timing ablations of the persistent W8A8 GEMM kernel (W8_ABL masks, see gemm_w8a8_w4.hip).
#   tools/gemm_w8_ablations.sh build 0 8 15 ...   (CPU box: build/w8abl_<mask>/liblx2v_hip.so)
#   tools/gemm_w8_ablations.sh run 0 8 15 ...     (GPU box: cycles / duration / clock of the fp8 [75600,5120]x[5120,5120])
set -o pipefail
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  cd lightx2v_amd/csrc
  for m in "$@"; do
    d=../../build/w8abl_$m; mkdir -p $d
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DW8_ABL=${m%%_*} ${W8_EXTRA} -x hip -c gemm_w8a8_w4.hip -o $d/w8.o || exit 1
    objs=$(ls ../../build/obj/*.o | grep -v gemm_w8a8_w4)
    hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so $objs $d/w8.o && rm $d/w8.o
  done
  exit 0
fi
mkdir -p gpurun_out/w8abl
for m in "$@"; do
  lib=build/w8abl_$m
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/w8abl/$m -o p -- ./build/kernel_harness gemm8_1 75600 ${GEMM_H:-40} 5 > gpurun_out/w8abl/$m.log 2>&1 || { echo "mask $m failed"; tail -3 gpurun_out/w8abl/$m.log; exit 1; }
  python3 - "$m" gpurun_out/w8abl/$m <<'PY'
import csv, glob, sys
tag, d = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_w8a8" in r["Kernel_Name"]]
cyc = [float(r["Counter_Value"]) / 8 for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(f"mask {tag:>4}: cycles {sum(cyc)/len(cyc)/1e6:.2f} M  dur {sum(dur)/len(dur):.3f} ms  clock {sum(cyc)/sum(dur)/1e6:.2f} GHz (n={len(rows)})", flush=True)
PY
done
