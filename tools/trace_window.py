#!/usr/bin/env python3
"""Average kernel durations over the TIMED steps of a bench.py run from a rocprofv3 kernel trace.

    python tools/trace_window.py gpurun_out/prof_<tag>/p_kernel_trace.csv gpurun_out/prof_<tag>/bench.json K out.json

bench.py's default run spends thousands of untimed launches on preparing the population; the --stats
table averages over all of them.  This picks, for every kernel, its launches inside the window of the
last K launches of the main tower kernel (= the K timed steps when the run was made with
--no-f32-side-run --games-leg 0) and compares the tower's sum with the in-bench HIP-event figure.
"""
import csv
import json
import sys


def main(trace, bench_json, K, out):
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    main_k = max((r["Kernel_Name"] for r in rows if "k_tower" in r["Kernel_Name"]),
                 key=lambda k: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == k))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"] == main_k]
    first = idx[-K]
    # the step's k_select precedes the main tower launch: start the window there
    while first > 0 and "k_select" not in rows[first]["Kernel_Name"]:
        first -= 1
    win = rows[first:]
    per = {}
    for r in win:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    summary = {k: {"launches": len(v), "mean_us": sum(v) / len(v), "us_per_step": sum(v) / K} for k, v in sorted(per.items())}
    tower_us = sum(v["us_per_step"] for k, v in summary.items() if k.startswith("k_tower"))
    b = json.load(open(bench_json))
    res = {"main_kernel": main_k.split("(")[0], "timed_steps": K, "kernels": summary,
           "tower_kernels_us_per_step_from_trace": tower_us,
           "tower_us_per_step_in_bench_hip_events": b["roofline"]["avg_launch_us"],
           "bench_ms_per_step": b["ms_per_step"], "bench_value": b["value"],
           "note": "HIP events bracket the main launch, the remainder launch and the idle f32-fallback launch of a step; "
                   "the trace figure is the sum of those kernels' own durations (gaps between them excluded)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("main_kernel", "tower_kernels_us_per_step_from_trace", "tower_us_per_step_in_bench_hip_events")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
