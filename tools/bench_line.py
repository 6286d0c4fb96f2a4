#!/usr/bin/env python3
"""One line per bench.py JSON on stdin: the three regimes with their kernel times (used by the sweep scripts)."""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, s, a = d["roofline_env_step"], d.get("steady_state"), d.get("all_armed")
out = "headline %.1f M (%.1f us; K1 %.1f K2 %.1f)" % (d["value"] / 1e6, d["ms_per_step"] * 1e3, r["substeps_kernel_ms"] * 1e3, r["engage_observe_kernel_ms"] * 1e3)
if s: out += "  steady %.1f M (%.1f us; K1 %.1f K2 %.1f)" % (s["value"] / 1e6, s["ms_per_step"] * 1e3, s["kernels"]["substeps_kernel_ms"] * 1e3, s["kernels"]["engage_observe_kernel_ms"] * 1e3)
if a: out += "  all-armed %.1f M" % (a["value"] / 1e6)
print(out)
