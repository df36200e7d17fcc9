#!/bin/bash
# fp8 one-wave-per-SIMD attention kernel: timing ablations (results are wrong with any bit set).
#   tools/f8w4_ablations.sh build 1 2 4 ...   (here: builds build/f8w4_abl<N>/liblx2v_hip.so)
#   (a library built by hand into build/f8w4_abllv<N> is run with the exp-free P: tools/f8w4_ablations.sh run lv<N>)
#   tools/f8w4_ablations.sh run 0 1 2 4 ...   (GPU box: kernel time of the attention launch per library; 0 = the product library, w8 = its 8-wave kernel, lin = its exp-free P)
set -o pipefail
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  cd lightx2v_amd/csrc
  objs=$(ls ../../build/obj/*.o | grep -v attn_fp8_w4)
  for n in "$@"; do
    d=../../build/f8w4_abl$n
    mkdir -p $d
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -DF8W4_ABL=$n ${F8W4_EXTRA} -x hip -c attn_fp8_w4.hip -o $d/k.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so $objs $d/k.o || exit 1
    rm $d/k.o
  done
  exit 0
fi
mkdir -p gpurun_out/f8abl
for m in "$@"; do
  lib=build/f8w4_abl$m; unset LX2V_ATTN_FP8_KERNEL
  [ "$m" = 0 ] && lib=lightx2v_amd
  [ "$m" = w8 ] && { lib=lightx2v_amd; export LX2V_ATTN_FP8_KERNEL=w8; }
  [ "$m" = nosync ] && { lib=lightx2v_amd; export LX2V_ATTN_ROUND_SYNC=0; }
  [ "$m" = lin ] && { lib=lightx2v_amd; export LX2V_ATTN_FP8_P=linear; }
  case "$m" in lv*) export LX2V_ATTN_FP8_P=linear;; esac
  LD_LIBRARY_PATH=$lib timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/f8abl/m$m -o p -- ./build/kernel_harness selfattn8 75600 40 3 > gpurun_out/f8abl/m$m.log 2>&1 || { echo "mask $m: run failed"; tail -3 gpurun_out/f8abl/m$m.log; exit 1; }
  unset LX2V_ATTN_ROUND_SYNC LX2V_ATTN_FP8_P
  f=$(find gpurun_out/f8abl/m$m -name "*counter_collection.csv" | head -1)
  t=$(find gpurun_out/f8abl/m$m -name "*kernel_trace.csv" | head -1)
  python3 - "$m" "$f" "$t" <<'PY'
import csv, sys
m, f, t = sys.argv[1:4]
cyc = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn_fwd_fp8" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
ns = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(t)) if "attn_fwd_fp8" in r["Kernel_Name"]]
c, ms = sum(cyc) / len(cyc) / 8, sum(ns) / len(ns) / 1e6
print(f"mask {m:>6}: {c / 1e6:7.2f} M cycles  {ms:.3f} ms  ({c / ms / 1e6:.2f} GHz)")
PY
done
