#!/bin/bash
# Builds timing-only variants of the library (one per ablation mask of the 4-wave attention kernel)
# into build/abl_<mask>/liblx2v_hip.so; run the harness against one with LD_LIBRARY_PATH.
set -e
cd "$(dirname "$0")/../lightx2v_amd/csrc"
for m in "$@"; do
  d=../../build/abl_$m
  mkdir -p $d
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -DW4_ABL=$m -x hip -c attn_fwd_w4.hip -o $d/attn_fwd_w4.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so ../../build/obj/norm_ew.hip.o ../../build/obj/gemm_bf16.hip.o ../../build/obj/attn_fwd.hip.o $d/attn_fwd_w4.o ../../build/obj/capi.cpp.o
  rm $d/attn_fwd_w4.o
done
