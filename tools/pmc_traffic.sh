#!/bin/bash
# HBM traffic of kernels from the fabric counters (MI355X_MICROARCH.md §HBM: separate --pmc passes, FETCH_SIZE x 2 on gfx950).
# usage: pmc_traffic.sh <tag> <command...>   -> gpurun_out/<tag>_pmc.json (per kernel name: launches, mean FETCH_SIZE / WRITE_SIZE in KiB)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${TAG}_$c
  # PMC_REGEX (optional): collect counters only for matching kernels (the unfiltered FETCH_SIZE pass segfaults inside rocprofv3 on the 24-image step)
  rocprofv3 --kernel-trace --pmc $c ${PMC_REGEX:+--kernel-include-regex "$PMC_REGEX"} --output-format csv -d /tmp/pmc_${TAG}_$c -o p -- "$@" > $OUT/pmc_${TAG}_$c.log 2>&1
  echo "pmc $c rc=$?"
done
python - "$TAG" "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
tag, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: {"launches": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        name = re.sub(r"\(anonymous namespace\)::", "", k)
        name = re.sub(r"^void ", "", name).split("(")[0]
        res[name][c + "_KiB_mean"] = sum(v) / len(v)
        res[name]["launches"] = max(res[name]["launches"], len(v))
json.dump(res, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KiB_mean", 0) * kv[1]["launches"])[:25]:
    print(f"{k[:70]:70s} n={v['launches']:6d} fetch {v.get('FETCH_SIZE_KiB_mean', 0):12.1f} KiB  write {v.get('WRITE_SIZE_KiB_mean', 0):12.1f} KiB")
PY
