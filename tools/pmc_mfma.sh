#!/bin/bash
# MFMA-pipe occupancy of the GEMM / attention kernels inside the benchmark step (SURVEY.md 8d evidence row): ONE rocprofv3 --pmc pass
# (--kernel-trace only; no other trace domain) over `bench.py --no-overlap` (one stream: a dispatch's counters are its own), then per kernel
#   MFMA occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)      (MI355X_MICROARCH.md: BUSY counts cycles)
# usage: tools/pmc_mfma.sh <tag> [bench args...]   -> gpurun_out/<tag>_mfma.md
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma_$TAG
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-include-regex "gemm_bf16|attn_" --output-format csv -d /tmp/pmc_mfma_$TAG -o p -- python $R/bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-fwd-only --no-overlap --no-graph "$@" > $OUT/pmc_mfma_$TAG.log 2>&1
echo "pmc rc=$?"
python - "$TAG" "$OUT" "$*" <<'PY'
import csv, glob, sys, collections, re
tag, out, args = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmc_mfma_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    m = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else 0.0
    gui = m("GRBM_GUI_ACTIVE")
    occ = m("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) if gui else 0.0
    wc = m("SQ_WAVE_CYCLES") or 1.0
    rows.append((sum(c["GRBM_GUI_ACTIVE"]) if c.get("GRBM_GUI_ACTIVE") else 0, k, len(c["GRBM_GUI_ACTIVE"]), occ, m("SQ_INSTS_MFMA"), m("SQ_VALU_MFMA_BUSY_CYCLES"), gui,
                 m("SQ_WAIT_INST_ANY") / wc, m("SQ_WAIT_ANY") / wc, m("SQ_ACTIVE_INST_ANY") / wc))
with open(f"{out}/{tag}_mfma.md", "w") as f:
    f.write(f"# MFMA-pipe occupancy per kernel inside the benchmark step ({tag}): rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE -- python bench.py --no-overlap --no-graph {args}\n\n")
    f.write("One counter pass (no other trace domain), one stream, per-dispatch means.  occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs);\n"
            "x effective clock / 2.4 GHz = fraction of the 2.5 PF/s peak (random data clocks ~1.8-2.0 GHz).  wave-cycle shares: issue-stalled / parked at waitcnt or barrier / issuing.\n\n")
    f.write("| kernel | dispatches | MFMA occupancy | SQ_INSTS_MFMA | MFMA_BUSY cycles | GUI_ACTIVE (sum of 8 XCDs) | WAIT_INST | WAIT_ANY | ACTIVE_INST |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    for tot, k, n, occ, im, busy, gui, wi, wa, ai in sorted(rows, reverse=True)[:16]:
        f.write(f"| `{k[:70]}` | {n} | {100 * occ:.1f} % | {im:.3g} | {busy:.3g} | {gui:.3g} | {100 * wi:.0f} % | {100 * wa:.0f} % | {100 * ai:.0f} % |\n")
print(open(f"{out}/{tag}_mfma.md").read())
PY
