#!/bin/bash
# Builds build/x16_abl<N>/liblx2v_hip.so for each timing ablation N of the 16x16x32 attention kernel
# (results are wrong with any bit set; for timing with `LD_LIBRARY_PATH=build/x16_abl<N> build/kernel_harness selfattn ...`).
set -e
cd "$(dirname "$0")/../lightx2v_amd/csrc"
objs=$(ls ../../build/obj/*.o | grep -v attn_fwd_x16)
for n in "$@"; do
  d=../../build/x16_abl$n
  mkdir -p $d
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -DX16_ABL=$n -x hip -c attn_fwd_x16.hip -o $d/x16.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblx2v_hip.so $objs $d/x16.o
  rm $d/x16.o
done
