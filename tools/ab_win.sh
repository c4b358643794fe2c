cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "attention or sam" 2>&1 | tail -2
C="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --batch 2 --extra-batch 0 --steps 20 --warmup 5"
for i in 1 2 3; do
  for m in balanced 256wgs; do
    if [ $m = balanced ]; then E=""; else E="LLMSEG_WIN_NO_BALANCE=1"; fi
    env $E python bench.py $C 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],3), 'ms  fwd-only', round(d['fwd_only']['ms_per_step'],3))"
  done
done
