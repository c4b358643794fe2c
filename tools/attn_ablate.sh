#!/bin/bash
# Side builds of attention.hip with one section removed each (results wrong by design): what bounds the tile loop of attn_fwd_kernel.
#   tools/attn_ablate.sh build      (here, CPU)      -> build_exp/libattn_ab<k>.so
#   tools/attn_ablate.sh run [B]    (GPU box)        -> timings of tools/attn_one.py glob for every side build
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "build" ]; then
  mkdir -p "$ROOT/build_exp"
  for k in 1 2 3 4 5; do
    ( LLMSEG_OBJ="$ROOT/build/exp_obj$k" LLMSEG_OUT="$ROOT/build_exp/libattn_ab$k.so" bash "$ROOT/llmseg_amd/csrc/build.sh" -DATTN_ABLATE=$k -DWIN_ABLATE=$k > "$ROOT/build_exp/ab$k.log" 2>&1; tail -1 "$ROOT/build_exp/ab$k.log" ) &
  done
  wait
else
  B=${2:-2}
  for m in "wint $B" ; do
    echo -n "shipped   "; python "$ROOT/tools/attn_one.py" $m 2>&1 | grep -v amdgpu | grep "variant 1"
    for k in 1 2 3 4 5; do echo -n "ablate $k  "; LLMSEG_LIB="$ROOT/build_exp/libattn_ab$k.so" python "$ROOT/tools/attn_one.py" $m 2>&1 | grep -v amdgpu | grep "variant 1"; done
  done
  [ -n "$WIN_ONLY" ] && exit 0
  for m in glob dino; do
    echo -n "shipped   "; python "$ROOT/tools/attn_one.py" $m $B 2>&1 | grep -v amdgpu
    for k in 1 2 3 4 5; do echo -n "ablate $k  "; LLMSEG_LIB="$ROOT/build_exp/libattn_ab$k.so" python "$ROOT/tools/attn_one.py" $m $B 2>&1 | grep -v amdgpu; done
  done
fi
