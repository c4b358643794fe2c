# alternating same-box A/B of one environment switch on the 2-image bench: tools/ab_env.sh "VAR=1" [pairs]
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --batch 2 --extra-batch 0 --steps 20 --warmup 5"
for i in $(seq 1 ${2:-3}); do
  for m in default switched; do
    if [ $m = default ]; then E=""; else E="$1"; fi
    env $E python bench.py $C 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],3), 'ms  fwd-only', round(d['fwd_only']['ms_per_step'],3))"
  done
done
