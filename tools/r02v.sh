#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_f; rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "gemm_bf16|attn_|splitk" --output-format csv -d /tmp/pmc_f -o p -- python -X faulthandler $GRAFT_REPO_ROOT/bench.py --batch 24 --extra-batch 0 --no-cpu-baseline --steps 3 --warmup 1 --no-graph --no-fwd-only > $GRAFT_REPO_ROOT/gpurun_out/r02v_fetch.log 2>&1
echo rc=$?
grep -v "^    @" $GRAFT_REPO_ROOT/gpurun_out/r02v_fetch.log | tail -40 | cut -c1-250
python - <<'PY'
import csv, glob, collections, re, json, os
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in agg.items():
    name = re.sub(r"\(anonymous namespace\)::", "", k); name = re.sub(r"^void ", "", name).split("(")[0]
    res[name] = {"FETCH_SIZE_KiB_mean": sum(v) / len(v), "launches": len(v)}
json.dump(res, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02v_fetch.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["FETCH_SIZE_KiB_mean"] * kv[1]["launches"])[:8]:
    print(k[:70], v)
PY
