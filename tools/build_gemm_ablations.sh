#!/bin/bash
# Builds timing-only variants of the library, one per GEMM_ABL mask (see gemm_bf16.hip), into
# build/gabl_<mask>/liblx2v_hip.so; tools/gemm_cycles.sh runs the harness against them.
set -e
cd "$(dirname "$0")/../lightx2v_amd/csrc"
for m in "$@"; do
  d=../../build/gabl_$m
  mkdir -p $d
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DGEMM_ABL=$m -x hip -c gemm_bf16.hip -o $d/gemm_bf16.o
  objs=$(ls ../../build/obj/*.o | grep -v gemm_bf16)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so $objs $d/gemm_bf16.o
  rm $d/gemm_bf16.o
done
