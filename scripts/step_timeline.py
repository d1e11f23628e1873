"""Print the kernel timeline of the last full-batch query step found in a rocprofv3 kernel trace
(gpurun_out/kstats/*/*kernel_trace.csv)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "kstats", "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scan_mfma_kernel" in r["Kernel_Name"]]
pick = int(sys.argv[1]) if len(sys.argv) > 1 else -1
i0 = idx[pick]
j = i0
while j > 0 and "rotate_mfma" not in rows[j]["Kernel_Name"] and "rotate_valu" not in rows[j]["Kernel_Name"]:
    j -= 1
t0 = int(rows[j]["Start_Timestamp"])
prev_end = t0
for r in rows[j:i0 + 7]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f}us  dur {(e - s) / 1e3:8.1f}us gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:48]}")
    prev_end = e
