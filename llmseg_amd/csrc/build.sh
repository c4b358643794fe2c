#!/bin/bash
# Build libllmseg_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot; it is git-ignored).
# One object per source (compiled in parallel, rebuilt only when the source or a header changed), then one link.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${LLMSEG_OUT:-$ROOT/llmseg_amd/libllmseg_hip.so}"      # LLMSEG_OUT: side builds for experiments (tools/)
OBJ="${LLMSEG_OBJ:-$ROOT/build/obj}"                          # build/ is git-ignored and gpurun-ignored
SRCS="gemm.hip attention.hip attention_bwd.hip pointwise.hip head.hip head_f32.hip backward.hip targets.hip image.hip capi.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE $*"
mkdir -p "$OBJ"
SIG="$(echo "$FLAGS" | md5sum | cut -c1-8)"
pids=()
objs=()
for s in $SRCS; do
  [ -f "$HERE/$s" ] || continue
  o="$OBJ/${s%.*}.$SIG.o"
  objs+=("$o")
  newest="$(ls -t "$HERE/$s" "$HERE"/*.h "$ROOT/include/llmseg_hip.h" | head -1)"
  if [ ! -f "$o" ] || [ "$newest" -nt "$o" ]; then
    ( hipcc $FLAGS -c "$HERE/$s" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $OUT"
# a kernel whose host stub the compiler dropped shows up only as an undefined symbol at load time: fail the build here
python3 -c "import ctypes, sys; ctypes.CDLL(sys.argv[1])" "$OUT"
