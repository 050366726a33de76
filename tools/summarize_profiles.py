#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the small files kept in profiles/.

usage: tools/summarize_profiles.py <tag> <kernel-trace-dir> <pmc_fetch_dir> <pmc_write_dir> [<pmc_l2_dir>]
Writes profiles/<tag>_kernel_stats.csv (copy of the --stats table) and
profiles/<tag>_pmc_summary.json (per-kernel mean counter values, HBM traffic per launch
computed as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate passes; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes, so reads are
2 x FETCH_SIZE)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def means(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in out.items()}


def main():
    tag, kt, pf, pw = sys.argv[1:5]
    pl2 = sys.argv[5] if len(sys.argv) > 5 else None
    os.makedirs("profiles", exist_ok=True)
    for f in glob.glob(os.path.join(kt, "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
    fetch, write = means(pf), means(pw)
    l2 = means(pl2) if pl2 else {}
    summary = {}
    for k in sorted(set(fetch) | set(write)):
        e = {}
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            e["FETCH_SIZE_KiB_mean"], e["launches"] = fetch[k]["FETCH_SIZE"]
        if k in write and "WRITE_SIZE" in write[k]:
            e["WRITE_SIZE_KiB_mean"] = write[k]["WRITE_SIZE"][0]
        if "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
            e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024)
            e["note"] = "reads = 2 x FETCH_SIZE (gfx950 128-B requests tallied as 64 B); access widths here are 4/8 B per lane (uncalibrated)"
        if k in l2:
            for c, (m, n) in l2[k].items():
                e[c + "_mean"] = m
            if "TCC_HIT_sum" in l2[k] and "TCC_MISS_sum" in l2[k]:
                h, m = l2[k]["TCC_HIT_sum"][0], l2[k]["TCC_MISS_sum"][0]
                e["l2_hit_rate"] = h / (h + m) if h + m else None
        summary[k] = e
    json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
    # the bench line of the FETCH pass names the kernel signature and configuration the counters belong to: with it the
    # summary becomes profiles/pmc_current.json, which bench.py refuses to quote for any other kernel / layout
    log = pf.rstrip("/") + ".log"
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
        if lines:
            b = json.loads(lines[-1])
            cur = {"config": {"N": b["config"]["N"], "nnz_row": int(round(b["config"]["nnz_per_row"])) - 1, "n_gpus": b["n_gpus"]},
                   "signature": b["roofline"].get("kernel_signature"), "layout": b["config"].get("layout"),
                   "kernels": summary, "source": f"tools/profile_round.sh {tag} (separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE, "
                                                 "TCC_HIT/MISS; reads = 2 x FETCH_SIZE on gfx950)"}
            json.dump(cur, open("profiles/pmc_current.json", "w"), indent=1, sort_keys=True)
            print("profiles/pmc_current.json <-", cur["config"], cur["signature"])
    print(json.dumps({k: v for k, v in summary.items() if "spmv" in k or "minres" in k}, indent=1))


if __name__ == "__main__":
    main()
