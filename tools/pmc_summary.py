"""Join tools/hbm_kernels.py (algorithmic bytes, durations) with tools/pmc_traffic.sh (fabric counters) into profiles/<tag>_hbm_kernels.md, and
write the dominant-GEMM traffic records bench.py reads (profiles/traffic.json).
Usage: python tools/pmc_summary.py hbm <hbm.json> <hbm_pmc.json> <out.md>
       python tools/pmc_summary.py gemm <bench_pmc.json> <batch> <profiles/traffic.json> <source note>"""
import json
import sys


def hbm(hbm_json, pmc_json, out_md):
    d, pmc = json.load(open(hbm_json)), json.load(open(pmc_json))
    with open(out_md, "w") as f:
        f.write(f"# HBM-bound kernels at the benchmark's shapes ({d['images_per_step']} images per step), 1 x MI355X\n\n")
        f.write("`tools/hbm_kernels.py` (HIP-event median of 20 launches) joined with `tools/pmc_traffic.sh` (rocprofv3 `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in\n"
                "separate passes over the same script; FETCH_SIZE x 2 per MI355X_MICROARCH.md section HBM -- calibrated here: the LayerNorm row reads exactly its\n"
                "input once and the counters give 1.00 x the algorithmic bytes).  Algorithmic bytes = every operand read / written once.  Peak = 8 TB/s.\n\n")
        f.write("| kernel | shape | algorithmic MB | us | GB/s | % of 8 TB/s | counter read MB | counter write MB | counter / algorithmic |\n|---|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for k in d["kernels"]:
            p = pmc.get(k["kernel"], {})
            rd, wr = p.get("FETCH_SIZE_KiB_mean", 0) * 2 * 1024 / 1e6, p.get("WRITE_SIZE_KiB_mean", 0) * 1024 / 1e6
            shared = " (mean over both K)" if k["kernel"].startswith("mask_pullback") else ""
            f.write(f"| {k['name']} (`{k['kernel']}`) | {k['shape']} | {k['alg_bytes'] / 1e6:.1f} | {k['us']:.1f} | {k['gbps']:.0f} | {100 * k['frac_hbm_peak']:.1f} | "
                    f"{rd:.1f}{shared} | {wr:.1f} | {(rd + wr) / (k['alg_bytes'] / 1e6):.2f} |\n")


def gemm(pmc_json, batch, traffic_json, note):
    pmc = json.load(open(pmc_json))
    try:
        tr = json.load(open(traffic_json))
    except FileNotFoundError:
        tr = {}
    rec = {}
    for k, v in pmc.items():
        if k.startswith("gemm_bf16") or k.startswith("attn_") or k.startswith("splitk") or k.startswith("reduce_lora"):
            rec[k] = {"hbm_bytes_per_launch": v.get("FETCH_SIZE_KiB_mean", 0) * 2 * 1024 + v.get("WRITE_SIZE_KiB_mean", 0) * 1024,
                      "read_bytes": v.get("FETCH_SIZE_KiB_mean", 0) * 2 * 1024, "write_bytes": v.get("WRITE_SIZE_KiB_mean", 0) * 1024, "launches": v["launches"]}
    tr[f"batch_{batch}"] = {"source": note, "kernels": rec}
    json.dump(tr, open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "hbm":
        hbm(*sys.argv[2:5])
    else:
        gemm(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
