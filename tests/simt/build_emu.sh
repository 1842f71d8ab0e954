#!/bin/sh
# Build the SIMT-emulated library from the UNMODIFIED kernel sources (test infrastructure).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../deepqmc_amd/csrc
OUT=$HERE/_build
mkdir -p "$OUT"
FLAGS="-x c++ -std=c++17 -O1 -fPIC -w -I$HERE -I$SRC -I$HERE/../../include"
pids=""
for f in engine kernel_linear kernel_fused2 kernel_attention kernel_attention_mfma kernels_graph kernels_head kernels_mcmc kernels_ecp; do
  if [ ! -f "$OUT/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/$f.o" ] || [ "$SRC/common.h" -nt "$OUT/$f.o" ] || \
     [ "$SRC/kernels.h" -nt "$OUT/$f.o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$OUT/$f.o" ] || \
     [ "$HERE/../../include/dqmc.h" -nt "$OUT/$f.o" ]; then
    g++ $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
g++ -std=c++17 -O1 -fPIC -w -I"$HERE" -c "$HERE/simt_runtime.cpp" -o "$OUT/simt_runtime.o"
g++ -shared -o "$OUT/libdqmc_emu.so" "$OUT"/engine.o "$OUT"/kernel_linear.o "$OUT"/kernel_fused2.o "$OUT"/kernel_attention.o "$OUT"/kernel_attention_mfma.o "$OUT"/kernels_graph.o "$OUT"/kernels_head.o "$OUT"/kernels_mcmc.o "$OUT"/kernels_ecp.o "$OUT"/simt_runtime.o
echo "$OUT/libdqmc_emu.so"
