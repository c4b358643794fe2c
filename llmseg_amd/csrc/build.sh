#!/bin/bash
# Build libllmseg_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot; it is git-ignored).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${LLMSEG_OUT:-$ROOT/llmseg_amd/libllmseg_hip.so}"      # LLMSEG_OUT: side builds for experiments (tools/)
SRCS="$HERE/gemm.hip $HERE/attention.hip $HERE/attention_bwd.hip $HERE/pointwise.hip $HERE/head.hip $HERE/backward.hip $HERE/targets.hip $HERE/capi.cpp"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$ROOT/include" -I"$HERE" $SRCS -o "$OUT" "$@"
echo "built $OUT"
# a kernel whose host stub the compiler dropped shows up only as an undefined symbol at load time: fail the build here
python3 -c "import ctypes, sys; ctypes.CDLL(sys.argv[1])" "$OUT"
