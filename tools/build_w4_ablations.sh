#!/bin/bash
# Builds timing-only variants of the library (one per ablation mask of the 32x32x16 one-wave-per-SIMD attention kernel; time it with LX2V_ATTN_KERNEL=w4)
# into build/abl_<mask>/liblx2v_hip.so; run the harness against one with LD_LIBRARY_PATH.
set -e
cd "$(dirname "$0")/../lightx2v_amd/csrc"
for m in "$@"; do
  d=../../build/abl_$m
  mkdir -p $d
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -DW4_ABL=$m -x hip -c attn_fwd_w4.hip -o $d/attn_fwd_w4.o
  objs=$(ls ../../build/obj/*.o | grep -v attn_fwd_w4)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so $objs $d/attn_fwd_w4.o
  rm $d/attn_fwd_w4.o
done
