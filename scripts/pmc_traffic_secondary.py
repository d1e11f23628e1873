"""gpurun_out/pmc_fetch_<tag>/ (rocprofv3 --pmc FETCH_SIZE --kernel-trace of bench.py on a secondary workload) + the bench line of
the same run -> profiles/scan_traffic_<tag>.json: HBM bytes (FETCH_SIZE x 2, gfx950) of the workload's matrix-core scan launches,
what bench.py quotes as `roofline.traffic` for that workload.  usage: pmc_traffic_secondary.py <tag> <round>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
csv.field_size_limit(10**9)
cc = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_fetch_{tag}", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
kt = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_fetch_{tag}", "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
bench = json.loads(open(os.path.join(ROOT, "gpurun_out", f"pmc_fetch_{tag}.json")).read().strip().splitlines()[-1])
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))}
disp = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "FETCH_SIZE" and "scan_mfma_kernel" in r["Kernel_Name"]:
        d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "kb": 0.0})
        d["kb"] += float(r["Counter_Value"])
ids = list(disp)
per_step = max(1, round(len(ids) / max(1, bench["steps"] + bench["warmup"] + 1)))   # matrix-core launches of one batch
last = ids[-per_step:]                                                               # the last batch's launches
launches = [{"kernel": disp[i]["name"], "ms_under_pmc": round(dur.get(i, 0), 4), "hbm_read_bytes": int(disp[i]["kb"] * 1024 * 2)} for i in last]
dom = max(launches, key=lambda l: l["ms_under_pmc"])
c = bench["config"]
out = {"config": {"vectors": c["n_per_gpu"], "dim": c["dim"], "lists": c["lists_total"] // bench["n_gpus"], "nprobe": c["nprobe"], "batch": c["batch"],
                  "distribution": "hard" if str(c.get("distribution", "easy")).startswith("hard") else "easy"},
       "source": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py ... ({c['workload']}); the matrix-core launches of the last batch of the run",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2",
       "dominant_launch": {"kernel": dom["kernel"], "ms_under_pmc": dom["ms_under_pmc"], "hbm_read_bytes": dom["hbm_read_bytes"]},
       "matrix_launches_of_one_batch": launches,
       "hbm_bytes_per_launch": int(sum(l["hbm_read_bytes"] for l in launches) / len(launches))}
dst = os.environ.get("RQ_PROFILE_OUT", os.path.join(ROOT, "profiles"))
json.dump(out, open(os.path.join(dst, f"scan_traffic_{tag}.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{rnd}_scan_traffic_{tag}.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:900])
