"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV directory: python tools/trace_summary.py DIR [substring]"""
import csv, glob, sys
from collections import OrderedDict
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
runs = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void grbda_hip::", "")
    if sub in n:
        runs.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
prev, cnt, tot = None, 0, 0.0
for n, ms in runs + [(None, 0)]:
    if n != prev and prev is not None:
        print(f"{prev:60s} x{cnt:3d}  avg {tot / cnt:8.4f} ms")
        cnt, tot = 0, 0.0
    prev = n
    cnt += 1
    tot += ms
