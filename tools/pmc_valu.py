#!/usr/bin/env python3
"""Turn the counter passes of tools/pmc_valu.sh into profiles/pmc_valu_latest.json (read by bench.py for roofline.valu_*): VALU instructions
and VALU-busy cycles of one sub-step launch, per armed-drone sub-step.

    python3 tools/pmc_valu.py gpurun_out/pmc_valu <armed drones per env in the profiled window> <envs>

SQ_ACTIVE_INST_VALU counts quad-cycles summed over every SIMD (x 4 = cycles, /opt/skills/guides/MI355X_MICROARCH.md); GRBM_GUI_ACTIVE
is summed over the 8 XCDs.  VALU-busy fraction of the launch = (SQ_ACTIVE_INST_VALU x 4 / 1 024 SIMDs) / (GRBM_GUI_ACTIVE / 8)."""
import collections, csv, glob, json, os, sys

out, armed, n_envs = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
SIMDS, SUBSTEPS = 1024, 16
vals = {}
for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES"):
    f = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = "substeps_kernel" if "substeps_kernel" in r["Kernel_Name"] else ("engage" if "engage" in r["Kernel_Name"] else None)
        if k and r["Counter_Name"] == c:
            agg[k].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in agg:
        vals.setdefault(k, {})[c] = sum(agg[k][-10:]) / len(agg[k][-10:])
        vals[k].setdefault("_dur_ns", []).append(sum(dur[k][-10:]) / len(dur[k][-10:]))
res = {"source": "tools/pmc_valu.sh: one rocprofv3 --pmc pass per counter of `bench.py --steps 20 --warmup 10 --headline-only --no-profile-events`, "
                 "averages of the last 10 launches", "armed_drones_per_env": armed, "n_envs": n_envs}
for k, v in vals.items():
    d = {c: x for c, x in v.items() if c != "_dur_ns"}
    d["launch_ns_under_the_profiler"] = sum(v["_dur_ns"]) / len(v["_dur_ns"])
    if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        d["valu_busy_cycles_per_simd"] = d["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS
        d["launch_cycles"] = d["GRBM_GUI_ACTIVE"] / 8
        d["valu_busy_frac"] = d["valu_busy_cycles_per_simd"] / d["launch_cycles"]
    if k == "substeps_kernel":
        flights = armed * n_envs / 64.0 * SUBSTEPS          # wave-level drone sub-steps per launch
        if "SQ_INSTS_VALU" in d:
            d["valu_instructions_per_drone_substep"] = d["SQ_INSTS_VALU"] / flights
        if "SQ_ACTIVE_INST_VALU" in d:
            d["valu_busy_cycles_per_drone_substep"] = d["SQ_ACTIVE_INST_VALU"] * 4 / flights
    res[k] = d
os.makedirs("profiles", exist_ok=True)
json.dump(res, open(os.path.join(out, "pmc_valu_latest.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
