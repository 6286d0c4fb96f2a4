#!/usr/bin/env python3
"""Average kernel durations of the bench's HEADLINE window out of a rocprofv3 --kernel-trace CSV of `python3 bench.py --gpus 1 --steps K --warmup W`.

    python tools/headline_from_trace.py <kernel_trace.csv> [--steps 20] [--warmup 5] [--bench-json line.json]

bench.py runs W warm-up steps, the K timed steps, then a replay of the same K steps with event-timed launches (and, unless
--headline-only, the steady-state and all-armed windows), so rocprofv3's --stats average mixes regimes.  This picks launches W..W+K
of the sub-step kernel and of the engage kernel (the timed window itself) and W+K..W+2K (its replay), and prints the roofline fraction
`algorithmic bytes / average duration / 8 TB/s` next to the one bench.py printed, so that the line's `roofline.frac` can be re-derived
from the committed trace."""
import argparse
import csv
import json

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--bench-json", default=None)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
dur = lambda key: [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
k1, k2 = dur("substeps_kernel"), dur("engage_")
W, K = a.warmup, a.steps
out = {"trace": a.trace, "launches_in_trace": len(k1),
       "window": {"substeps_kernel_us": float(np.mean(k1[W:W + K])), "engage_kernel_us": float(np.mean(k2[W:W + K])), "launches": K},
       "replay": {"substeps_kernel_us": float(np.mean(k1[W + K:W + 2 * K])), "engage_kernel_us": float(np.mean(k2[W + K:W + 2 * K])), "launches": K},
       "whole_run_stats_average_us": {"substeps_kernel": float(np.mean(k1)), "engage_kernel": float(np.mean(k2))}}
if a.bench_json:
    d = json.loads([l for l in open(a.bench_json) if l.startswith("{")][0])
    r = d["roofline"]
    by = r["algorithmic_bytes_per_launch"]
    out["bench_line"] = {"roofline_frac": r["frac"], "avg_launch_us": 1e3 * r["avg_launch_ms"], "kernel": r["kernel"], "algorithmic_bytes_per_launch": by}
    for k in ("window", "replay"):
        out[k]["roofline_frac_from_trace"] = by / (out[k]["substeps_kernel_us"] * 1e-6) / 8e12
    out["printed_over_trace_replay"] = r["frac"] / out["replay"]["roofline_frac_from_trace"]
print(json.dumps(out, indent=1))
