#!/bin/bash
# Side build of the library with extra -D flags on gemm.hip only (experiments): tools/side_gemm.sh <name> <flags...> -> build_exp/lib_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
SIG="$(ls "$ROOT"/build/obj/capi.*.o | head -1 | sed 's/.*capi\.\(.*\)\.o/\1/')"
mkdir -p "$ROOT/build_exp"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$ROOT/llmseg_amd/csrc" "$@" -c "$ROOT/llmseg_amd/csrc/gemm.hip" -o "$ROOT/build_exp/gemm_$name.o"
objs=$(ls "$ROOT"/build/obj/*."$SIG".o | grep -v "/gemm\.")
hipcc --offload-arch=gfx950 -shared -fPIC $objs "$ROOT/build_exp/gemm_$name.o" -o "$ROOT/build_exp/lib_$name.so"
rm -f "$ROOT/build_exp/gemm_$name.o"
echo "built build_exp/lib_$name.so"
