#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into profiles/<name>.json.

    python tools/pmc_summary.py gpurun_out/pmc3 profiles/r01_pmc_tower.json [--last N]

--last N: only the last N launches of every kernel (the timed window of a bench.py run whose last launches are the timed steps).

Traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE reads one half of the bytes of wide coalesced streams, so the read
side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import json
import sys


def main(src, dst, last=0):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    files = sorted(glob.glob(src + "/*/*/*_counter_collection.csv")) + sorted(glob.glob(src + "/*/*_counter_collection.csv"))
    for f in files:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        if last > 0:  # the last `last` launches of every (kernel, counter) of this pass
            seen = collections.Counter()
            keep = []
            for r in reversed(rows):
                key = (r["Kernel_Name"], r["Counter_Name"])
                seen[key] += 1
                if seen[key] <= last:
                    keep.append(r)
            rows = keep
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {"kernels": {}}
    if last > 0:
        out["launches_considered"] = "the last %d launches of every kernel" % last
    for k, v in per.items():
        if k.startswith("__amd"):
            continue
        out["kernels"][k] = {c: {"launches": len(x), "mean": sum(x) / len(x)} for c, x in v.items()}
        out["kernels"][k]["mean_duration_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
    tower = sorted((k for k in out["kernels"] if k.startswith("k_tower")),
                   key=lambda k: -out["kernels"][k]["mean_duration_ns_under_pmc"])  # main launch, not the tail variants
    if tower:
        t = out["kernels"][tower[0]]
        fetch = t.get("FETCH_SIZE", {}).get("mean")
        write = t.get("WRITE_SIZE", {}).get("mean")
        if fetch is not None and write is not None:
            out["traffic_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
            out["note"] = ("(2*FETCH_SIZE + WRITE_SIZE)*1024 B per k_tower launch (gfx950 FETCH_SIZE correction); the read side "
                           "is the 5.9 MB of packed weights re-streamed Infinity-Cache -> L2 by each XCD on every pass "
                           "(weights exceed the 4 MiB L2), not HBM; algorithmic bytes per launch = features in + head "
                           "activations out + weights once")
        busy = t.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean")
        gui = t.get("GRBM_GUI_ACTIVE", {}).get("mean")
        if busy and gui:
            out["mfma_busy_frac"] = busy / (gui / 8.0 * 1024.0)  # 8 XCDs summed; 256 CUs x 4 SIMDs
    # per-kernel traffic and MFMA busy fraction for every kernel that has the counters (training-step runs: no k_tower)
    for k, t in out["kernels"].items():
        fetch, write = t.get("FETCH_SIZE", {}).get("mean"), t.get("WRITE_SIZE", {}).get("mean")
        if fetch is not None and write is not None:
            t["traffic_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
        busy, gui = t.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean"), t.get("GRBM_GUI_ACTIVE", {}).get("mean")
        if busy and gui:
            t["mfma_busy_frac"] = busy / (gui / 8.0 * 1024.0)
    # the build the counters were taken on and the evaluations per launch: from the bench line of the first pass
    import os
    line = os.path.join(src, "pass1.json")
    if os.path.exists(line):
        try:
            b = json.loads(open(line).read().strip().splitlines()[-1])
            out["build"] = b.get("build")
            out["steps"], out["warmup"] = b.get("steps"), b.get("warmup")
            if b.get("roofline", {}).get("flops_per_launch"):
                out["evals_per_launch"] = b["nn_evals_per_sec"] * b["ms_per_step"] * 1e-3
        except (ValueError, KeyError):
            pass
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: out.get(k) for k in ("traffic_bytes_per_launch", "mfma_busy_frac", "evals_per_launch", "build")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--last" else 0)
