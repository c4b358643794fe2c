#!/bin/bash
# PMC counters for the GEMM kernel (own run: --pmc with --kernel-trace only).  usage: pmc_gemm.sh <tag> <variant> M N K
TAG=$1; V=$2; M=$3; N=$4; K=$5
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $grp | awk '{print $1}')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$n -o p -- python $R/tools/gemm_one.py $V $M $N $K 3 > $OUT/$n.log 2>&1
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"  {k:28s} per-dispatch {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
