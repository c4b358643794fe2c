#!/bin/bash
# Issue-side counters of one kernel run standalone (two rocprofv3 --pmc passes, --kernel-trace only):
#   tools/pmc_kernel.sh <tag> <kernel-regex> <command...>      -> gpurun_out/<tag>_pmc.md
TAG=$1; RE=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck_$TAG
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$RE" --output-format csv -d /tmp/pmck_$TAG/p$i -o p -- "$@" > $OUT/pmck_${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
python - "$TAG" "$OUT" <<'PY'
import csv, glob, sys, collections, re
tag, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmck_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"^void ", "", name).split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_pmc.md", "w") as f:
    for k, c in agg.items():
        f.write(f"## `{k}`\n\n| counter | mean per dispatch | / SQ_WAVE_CYCLES |\n|---|---:|---:|\n")
        wc = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else 1.0
        for n in sorted(c):
            m = sum(c[n]) / len(c[n]); f.write(f"| {n} | {m:.4g} | {m / wc:.3f} |\n")
        f.write("\n")
print(open(f"{out}/{tag}_pmc.md").read())
PY
