"""Print the last N kernel dispatches of a rocprofv3 --kernel-trace CSV (name, microseconds, VGPRs, LDS bytes).
    python tools/kernel_times.py <dir> [N] [substring ...]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
subs = sys.argv[3:]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if subs:
    rows = [r for r in rows if any(s in r["Kernel_Name"] for s in subs)]
for r in rows[-n:]:
    name = r["Kernel_Name"].replace("syg::(anonymous namespace)::", "")
    print("%-60s %8.1f us  VGPR %3s  LDS %6s  grid %s" % (name[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                      r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")))
