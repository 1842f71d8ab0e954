#!/bin/sh
# Build the SIMT-emulated library from the UNMODIFIED kernel sources (test infrastructure).
# Concurrent callers (two ranks of a gloo test) are serialised by a lock; the library is re-linked only when an object
# changed and replaced atomically, so a process that has it mapped never sees a half-written file.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../deepqmc_amd/csrc
OUT=$HERE/_build
mkdir -p "$OUT"
exec 9> "$OUT/.lock"
flock 9
FLAGS="-x c++ -std=c++17 -O1 -fPIC -w -I$HERE -I$SRC -I$HERE/../../include"
pids=""
GEN=$(cd "$SRC" && ls gen/*.hip 2>/dev/null | sed 's/\.hip$//')
mkdir -p "$OUT/gen"
for f in engine kernel_linear kernel_fused2 kernel_attention kernel_attention_mfma kernels_graph kernels_head kernels_mcmc kernels_ecp spec_registry $GEN; do
  stale_inl=0
  if [ "$f" = engine ]; then for i in "$SRC"/engine_*.inl; do [ "$i" -nt "$OUT/$f.o" ] && stale_inl=1; done; fi
  if [ ! -f "$OUT/$f.o" ] || [ $stale_inl = 1 ] || [ "$SRC/$f.hip" -nt "$OUT/$f.o" ] || [ "$SRC/common.h" -nt "$OUT/$f.o" ] || \
     [ "$SRC/kernels.h" -nt "$OUT/$f.o" ] || [ "$SRC/spec_device.h" -nt "$OUT/$f.o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$OUT/$f.o" ] || \
     [ "$HERE/../../include/dqmc.h" -nt "$OUT/$f.o" ]; then
    g++ $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
if [ ! -f "$OUT/simt_runtime.o" ] || [ "$HERE/simt_runtime.cpp" -nt "$OUT/simt_runtime.o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$OUT/simt_runtime.o" ]; then
  g++ -std=c++17 -O1 -fPIC -w -I"$HERE" -c "$HERE/simt_runtime.cpp" -o "$OUT/simt_runtime.o"
fi
relink=0
[ -f "$OUT/libdqmc_emu.so" ] || relink=1
for o in "$OUT"/*.o "$OUT"/gen/*.o; do [ "$o" -nt "$OUT/libdqmc_emu.so" ] && relink=1; done
if [ $relink = 1 ]; then
  g++ -shared -o "$OUT/libdqmc_emu.so.tmp$$" "$OUT"/*.o "$OUT"/gen/*.o
  mv -f "$OUT/libdqmc_emu.so.tmp$$" "$OUT/libdqmc_emu.so"
fi
echo "$OUT/libdqmc_emu.so"
