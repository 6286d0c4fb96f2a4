#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into per-launch HBM bytes, corrected with the
calibration kernels of tools/pmc_calib.hip (known 1 GiB streams at 4 B/lane and 16 B/lane).  Writes
profiles/pmc_latest.json (read by bench.py for roofline.traffic)."""
import collections, csv, json, os, sys

out = sys.argv[1]
def load(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return agg
GIB = float(1 << 30)
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a = load(os.path.join(out, f"calib_{c}", "c_counter_collection.csv"))
    for k, v in a.items():
        cal[(c, k)] = sum(v) / len(v)
# counters are in KB (rocprofv3 derived metric): bytes = value * 1024; factor = true / reported
f4 = GIB / (cal[("FETCH_SIZE", "read4")] * 1024); f16 = GIB / (cal[("FETCH_SIZE", "read16")] * 1024)
w4 = GIB / (cal[("WRITE_SIZE", "write4")] * 1024); w16 = GIB / (cal[("WRITE_SIZE", "write16")] * 1024)
print(f"calibration: FETCH_SIZE x{f4:.3f} (4 B/lane) x{f16:.3f} (16 B/lane); WRITE_SIZE x{w4:.3f} (4 B/lane) x{w16:.3f} (16 B/lane)")
res = {"calibration": {"fetch_4B": f4, "fetch_16B": f16, "write_4B": w4, "write_16B": w16,
                       "note": "factor = true bytes / (counter * 1024) on 1 GiB streams (tools/pmc_calib.hip)"}}
fetch = load(os.path.join(out, "bench_FETCH_SIZE", "b_counter_collection.csv"))
write = load(os.path.join(out, "bench_WRITE_SIZE", "b_counter_collection.csv"))
for k in fetch:
    if "substeps_kernel" in k or "engage" in k:
        name = "substeps_kernel" if "substeps" in k else "engage_observe_kernel"   # engage_slots_kernel / engage_kernel<PM, IM>: the forms of the engage/observe step
        fr = sum(fetch[k][-10:]) / 10 * 1024; wr = sum(write[k][-10:]) / 10 * 1024
        # both kernels read with 4 B/lane loads; K1 writes state with 4 B/lane stores + the LIDAR background with 16 B/lane
        # stores, K2 writes with 4 B/lane stores: apply the 4 B factors (the 16 B write factor is reported alongside)
        res[name] = {"kernel": k, "fetch_raw_bytes": fr, "write_raw_bytes": wr, "hbm_bytes_per_launch": fr * f4 + wr * w4,
                     "fetch_bytes": fr * f4, "write_bytes": wr * w4}
        print(name, {kk: round(vv / 1e6, 1) for kk, vv in res[name].items() if kk != "kernel"}, "MB")
os.makedirs("profiles", exist_ok=True)
res["source"] = "tools/pmc_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 20 --warmup 10 --headline-only`, averages of the last 10 launches"
json.dump(res, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
print("wrote", os.path.join(out, "pmc_latest.json"))
