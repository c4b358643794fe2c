#!/bin/bash
# Register / LDS / spill table of every kernel in one source file (device-only compile for gfx950, metadata notes of the code object).
#   tools/kernel_regs.sh llmseg_amd/csrc/gemm.hip [regex]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$1"; PAT="${2:-.}"
OUT="$ROOT/build/dev_$(basename "${SRC%.*}")"
mkdir -p "$ROOT/build"
[ -n "$SKIP_COMPILE" ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -I"$ROOT/include" -I"$ROOT/llmseg_amd/csrc" -c "$SRC" -o "$OUT.bundle" 2>/dev/null
if /opt/rocm/lib/llvm/bin/llvm-readelf -h "$OUT.bundle" >/dev/null 2>&1; then cp "$OUT.bundle" "$OUT.o"; else
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input="$OUT.bundle" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$OUT.o" --unbundle; fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT.o" | python3 "$ROOT/tools/kernel_regs.py" "$PAT"
