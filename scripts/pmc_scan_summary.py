"""Summarise gpurun_out/pmc_mfma*/ (scripts/pmc_scan.sh): per-launch SQ counters of scan_mfma_kernel (or of the kernel
whose name contains argv[1]); the launch with the largest grid is reported."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "scan_mfma_kernel"  # e.g. "scan_kernel<" for the VALU scan stages
csv.field_size_limit(10**9)
tot = {}
for d in ("pmc_mfma", "pmc_mfma2"):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
    if not fs:
        print(d, "missing")
        continue
    fs.sort(key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(fs[-1])) if KERNEL in r["Kernel_Name"]]
    byd = collections.defaultdict(dict)
    grid = {}
    for r in rows:
        byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = byd[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        grid[int(r["Dispatch_Id"])] = int(r["Grid_Size"])
    # the dominant launch = the final-stage launch (largest grid); of its repetitions (warm-up, timed, breakdown and
    # counting steps of bench.py) the second one is the timed step
    big = sorted(d for d in byd if grid[d] == max(grid.values()))
    pick = big[1] if len(big) > 1 else big[0]
    print(f"{d}: {len(big)} launches of the dominant grid ({max(grid.values())} threads), dispatch {pick} taken")
    tot.update(byd[pick])
for k, v in sorted(tot.items()):
    print(f"{k:28s} {v:16.0f}")
if "SQ_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
    gui = tot["GRBM_GUI_ACTIVE"] / 8  # the counter is summed over the 8 XCDs
    simd_cycles = gui * 256 * 4  # one count per SIMD-cycle if every SIMD were busy the whole launch
    print("launch cycles (GRBM_GUI_ACTIVE):", gui)
    # SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in units of 4 cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles
    print(f"  matrix pipe busy          = {tot.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles:.3f}")
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if k in tot:
            print(f"  {k:26s}= {4 * tot[k] / simd_cycles:.3f} of SIMD-cycles")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        if k in tot:
            print(f"  {k:26s}= {4 * tot[k] / simd_cycles:.3f} of the 1-per-4-cycles issue rate")
    if "SQ_WAVE_CYCLES" in tot:
        wc = 4 * tot["SQ_WAVE_CYCLES"]
        print(f"  mean waves per SIMD       = {wc / simd_cycles:.2f}")
        print(f"  wave lifetime             = {wc / tot['SQ_WAVES']:.0f} cycles")
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            print(f"  {k:26s}= {4 * tot[k] / wc:.3f} of wave-cycles")
