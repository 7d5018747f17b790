"""Mean per-launch value of every counter found under <dir>/p*/**/*counter_collection.csv for the grbda kernels
(argv[2]: comma-separated substrings a kernel name must contain one of; default aba,rnea)."""
import csv, glob, os, sys
want = (sys.argv[2] if len(sys.argv) > 2 else "aba,rnea").split(",")
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "grbda" not in r["Kernel_Name"] or not any(w in r["Kernel_Name"] for w in want):
            continue  # (input generation runs project_kernel on TelloWithArms: not the kernel under study)
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, name), v in per.items():
        acc[name].append(v)
for name in sorted(acc):
    v = acc[name]
    print(f"{name:32s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
