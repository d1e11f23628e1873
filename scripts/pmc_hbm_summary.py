"""Summarise scripts/pmc_hbm_regime.sh: physical HBM bytes (PMC FETCH_SIZE, x2 gfx950 correction) and kernel
durations (separate --kernel-trace pass) of the scan launches in the small-batch regimes -> profiles/<round>_hbm_regime.json.

The three passes run the same deterministic command, so scan dispatches are matched by order: per regime
(3 warm-up + reps) calls x launches per call; the timed calls' dispatches are the last reps x L of the slice."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
csv.field_size_limit(10**9)
OUTDIR = os.environ.get("RQ_PROFILE_OUT", os.path.join(ROOT, "profiles"))
os.makedirs(OUTDIR, exist_ok=True)
is_scan = re.compile(r"^(void )?scan_(kernel|mfma_kernel|generic_kernel)")
is_early = re.compile(r"^(void )?sb_query_kernel")   # small-batch path: the per-query block that scans the head of the stream itself


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


def scan_dispatches_pmc(tag, pat=is_scan):
    cc = newest(f"hbm_{tag}_pmc/*/*counter_collection.csv")
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] != "FETCH_SIZE" or not pat.match(r["Kernel_Name"]):
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "kb": 0.0})
        d["kb"] += float(r["Counter_Value"])
    return [disp[i] for i in sorted(disp)]


def scan_dispatches_kt(tag, pat=is_scan):
    kt = newest(f"hbm_{tag}_kt/*/*kernel_trace.csv")
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for r in csv.DictReader(open(kt)) if pat.match(r["Kernel_Name"])]
    rows.sort()
    return [{"name": n.split("(")[0].replace("void ", ""), "ns": ns} for _, n, ns in rows]


out = {"correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section): bytes = FETCH_SIZE x 2.  "
                     "The x2 is applied here because the launches counted are the scan's: their loads are 16 bytes per lane, coalesced "
                     "(codes and factors of consecutive list positions); it would not be valid for gather kernels",
       "what_is_counted": "FABRIC bytes: FETCH_SIZE counts what the L2 requests from the fabric, i.e. it INCLUDES Infinity-Cache (MALL) hits -- "
                          "a rate above the ~6.3 TB/s that HBM itself delivers is fabric + cache bandwidth, not HBM bandwidth.  Keys named "
                          "physical_* (kept for continuity with rounds 2-3) hold these fabric bytes",
       "peak_GBps": 8000.0, "achievable_hbm_GBps": 6300.0, "workloads": []}
for tag in ("d128", "d768"):
    pj = os.path.join(ROOT, "gpurun_out", f"hbm_{tag}_plain.json")
    if not os.path.exists(pj) or newest(f"hbm_{tag}_pmc/*/*counter_collection.csv") is None:
        continue
    plain = json.load(open(pj))
    pmc, kt = scan_dispatches_pmc(tag), scan_dispatches_kt(tag)
    e_pmc, e_kt = scan_dispatches_pmc(tag, is_early), scan_dispatches_kt(tag, is_early)
    e_at = 0
    wl = {"config": plain["config"],
          "commands": [f"python3 scripts/hbm_regime.py ... (HIP events)",
                       "rocprofv3 --kernel-trace --stats -- (same)", "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- (same)"],
          "scan_dispatches": {"pmc_pass": len(pmc), "kernel_trace_pass": len(kt)}, "regimes": []}
    at = 0
    for rg in plain["regimes"]:
        L = int(round(rg["scan_launches_per_call"]))
        total = (rg["warmup_calls"] + rg["calls"]) * L
        sl_p, sl_k = pmc[at:at + total][-rg["calls"] * L:], kt[at:at + total][-rg["calls"] * L:]
        at += total
        phys = sum(d["kb"] for d in sl_p) * 1024 * 2 / rg["calls"]
        ns = sum(d["ns"] for d in sl_k) / rg["calls"]
        # the launch that streams most of the bytes (final stage)
        per_launch = collections.defaultdict(lambda: [0.0, 0.0])
        for j, (p, kk) in enumerate(zip(sl_p, sl_k)):
            per_launch[j % L][0] += p["kb"] * 2048 / rg["calls"]
            per_launch[j % L][1] += kk["ns"] / rg["calls"]
        dom = max(per_launch, key=lambda j: per_launch[j][0])
        early = None
        if rg.get("small_batch_passes_per_call", 0) >= 0.5:   # one sb_query_kernel per call (warm-up calls included)
            n_e = rg["warmup_calls"] + rg["calls"]
            ep, ek = e_pmc[e_at:e_at + n_e][-rg["calls"]:], e_kt[e_at:e_at + n_e][-rg["calls"]:]
            e_at += n_e
            if ep and ek:
                early = {"kernel": "sb_query_kernel (probe selection, query quantisation, the first stages in LDS)",
                         "ms_per_call_kernel_trace": sum(d["ns"] for d in ek) / len(ek) / 1e6,
                         "physical_hbm_bytes_per_call": sum(d["kb"] for d in ep) * 2048 / len(ep)}
        wl["regimes"].append({
            "whole_call_ms_hip_events": rg["total_ms_per_call"],
            "whole_call_algorithmic_GBps": rg["algorithmic_bytes_per_call"] / (rg["total_ms_per_call"] * 1e-3) / 1e9,
            "whole_call_algorithmic_frac_of_8TBps": rg["algorithmic_bytes_per_call"] / (rg["total_ms_per_call"] * 1e-3) / 1e9 / 8000.0,
            "early_part": early,
            "batch": rg["batch"], "scan_launches_per_call": L, "kernels": sorted({d["name"] for d in sl_p}),
            "algorithmic_bytes_per_call": rg["algorithmic_bytes_per_call"],
            "physical_hbm_bytes_per_call": phys,
            "scan_ms_per_call_hip_events": rg["scan_ms_per_call"],
            "scan_ms_per_call_kernel_trace": ns / 1e6,
            "algorithmic_GBps": rg["algorithmic_bytes_per_call"] / (ns * 1e-9) / 1e9 if ns else None,
            "physical_GBps": phys / (ns * 1e-9) / 1e9 if ns else None,
            "physical_frac_of_8TBps": phys / (ns * 1e-9) / 1e9 / 8000.0 if ns else None,
            "fabric_GBps_incl_infinity_cache": phys / (ns * 1e-9) / 1e9 if ns else None,
            "fabric_bytes_over_algorithmic_bytes": phys / rg["algorithmic_bytes_per_call"] if rg["algorithmic_bytes_per_call"] else None,
            "dominant_launch": {"index_in_call": dom, "physical_bytes": per_launch[dom][0], "ms": per_launch[dom][1] / 1e6,
                                "physical_GBps": per_launch[dom][0] / (per_launch[dom][1] * 1e-9) / 1e9 if per_launch[dom][1] else None},
        })
    out["workloads"].append(wl)
    st = newest(f"hbm_{tag}_kt/*/*kernel_stats.csv")
    if st:
        rows = [r for r in csv.DictReader(open(st)) if not any(t in r["Name"] for t in ("at::native", "Cijk_", "__amd_rocclr", "at::cuda", "rocprim", "hipcub"))]
        with open(os.path.join(OUTDIR, f"{ROUND}_hbm_regime_{tag}_kernel_stats.csv"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 scripts/hbm_regime.py ({json.dumps(plain['config'])}); engine kernels only\n")
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
json.dump(out, open(os.path.join(OUTDIR, f"{ROUND}_hbm_regime.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
