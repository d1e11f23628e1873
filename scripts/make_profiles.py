"""Turn the outputs of scripts/refresh_profiles.sh (under gpurun_out/) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
csv.field_size_limit(10**9)
OUT = os.environ.get("RQ_PROFILE_OUT", os.path.join(ROOT, "profiles"))
os.makedirs(OUT, exist_ok=True)


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    if not fs:
        raise SystemExit(f"missing gpurun_out/{pattern}")
    return fs[-1]


# ---- kernel stats ---------------------------------------------------------------------------------
stats = newest("kstats/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
shutil.copy(stats, os.path.join(OUT, f"{ROUND}_kernel_stats_full.csv"))
ours = [r for r in rows if not any(t in r["Name"] for t in ("at::native", "Cijk_", "__amd_rocclr", "at::cuda", "rocprim", "hipcub"))]
with open(os.path.join(OUT, f"{ROUND}_kernel_stats.csv"), "w") as f:
    bcfg = json.load(open(newest("bench_final.json")))["config"]
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-two-in-flight --no-secondary --no-batch-sweep --small-batch 0\n")
    f.write(f"# ({bcfg['workload']}: 2 warm-up + 3 timed + 1 breakdown query batches -- every scan_mfma_kernel launch in this table is a\n")
    f.write("#  full-batch launch, every launch alone on the device --, 68 single queries, one streamed 100M build); engine kernels only,\n")
    f.write("#  torch data-generation / ground-truth kernels are in the _full file\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in ours:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])

# ---- PMC pass: HBM bytes per scan launch --------------------------------------------------------------
cc = newest("pmc_fetch/*/*counter_collection.csv")
kt = newest("pmc_fetch/*/*kernel_trace.csv")
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))}
disp = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "vgpr": r.get("VGPR_Count", ""),
                                           "sgpr": r.get("SGPR_Count", ""), "fetch_kb": 0.0})
    d["fetch_kb"] += float(r["Counter_Value"])
engine = [(i, d) for i, d in disp.items() if not any(t in d["name"] for t in ("at::native", "Cijk_", "__amd_rocclr", "at::cuda"))]
with open(os.path.join(OUT, f"{ROUND}_pmc_fetch_size.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size", "VGPR_Count", "SGPR_Count", "FETCH_SIZE_KB_raw", "duration_ms"])
    for i, d in engine:
        w.writerow([d["name"].split("(")[0], d["grid"], d["vgpr"], d["sgpr"], f"{d['fetch_kb']:.6f}", f"{dur.get(i, 0):.4f}"])
import re
is_scan = re.compile(r"^(void )?scan_(kernel|mfma_kernel|generic_kernel)")
scans = [(i, d) for i, d in engine if is_scan.match(d["name"])]
# the launches of one full batch end with its matrix-core launch; bench.py runs warm-up, timed, breakdown and counting
# batches: take the TIMED one (the second), never the counting step
mf = [n for n, (i, d) in enumerate(scans) if "scan_mfma_kernel" in d["name"]]
pick = 1 if len(mf) > 1 else 0
last = mf[pick]
first = mf[pick - 1] + 1 if pick > 0 else 0
batch = scans[first:last + 1]
bench = json.load(open(newest("bench_final.json")))
ra = bench.get("roofline_scan_all_launches", bench["roofline"])
alg = ra["algorithmic_bytes_per_launch"] * ra["launches"] / bench["steps"]
dom_i, dom = batch[-1]
launches = [{"kernel": d["name"].split("(")[0].replace("void ", ""), "ms_under_pmc": round(dur.get(i, 0), 4),
             "hbm_read_bytes": int(d["fetch_kb"] * 1024 * 2)} for i, d in batch]
traffic = {
    "config": {"vectors": bench["config"]["n_per_gpu"], "dim": bench["config"]["dim"], "lists": bench["config"]["lists_total"] // bench["n_gpus"],
               "nprobe": bench["config"]["nprobe"], "batch": bench["config"]["batch"],
               "distribution": "hard" if str(bench["config"].get("distribution", "easy")).startswith("hard") else "easy"},
    "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-two-in-flight --gt-queries 10 "
              f"--small-batch 0 ({bench['config']['workload']})",
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2",
    "dominant_launch": {"kernel": launches[-1]["kernel"] + " (final stage: stream positions past the first list)",
                        "ms_under_pmc": launches[-1]["ms_under_pmc"], "fetch_bytes_raw": int(dom["fetch_kb"] * 1024),
                        "hbm_read_bytes": launches[-1]["hbm_read_bytes"]},
    "algorithmic_bytes_per_batch": int(alg),
    "scan_launches_per_batch": len(batch),
    "hbm_bytes_per_launch": int(sum(l["hbm_read_bytes"] for l in launches) / len(launches)),
    "launches_of_one_batch": launches,
}
json.dump(traffic, open(os.path.join(OUT, "scan_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(OUT, f"{ROUND}_scan_traffic.json"), "w"), indent=1)
# the bench line read the previous scan_traffic.json (or none, if the workload changed): attach this pass's figures
if bench.get("roofline", {}).get("bound") == "mfma":
    bench["roofline"]["traffic"] = traffic["dominant_launch"]["hbm_read_bytes"]
if "roofline_scan_all_launches" in bench:
    bench["roofline_scan_all_launches"]["traffic"] = traffic["hbm_bytes_per_launch"]
json.dump(bench, open(os.path.join(OUT, f"{ROUND}_bench_100M.json"), "w"))
tif = glob.glob(os.path.join(ROOT, "gpurun_out", "bench_two_in_flight.json"))
if tif:
    shutil.copy(tif[0], os.path.join(OUT, f"{ROUND}_bench_100M_two_in_flight.json"))
print(json.dumps(traffic, indent=1)[:1500])
