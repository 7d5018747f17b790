"""Mean per-launch value of every counter found under <dir>/p*/**/*counter_collection.csv for the grbda kernels."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "grbda" not in r["Kernel_Name"] or not ("aba" in r["Kernel_Name"] or "rnea" in r["Kernel_Name"]):
            continue  # (input generation runs project_kernel on TelloWithArms: not the kernel under study)
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, name), v in per.items():
        acc[name].append(v)
for name in sorted(acc):
    v = acc[name]
    print(f"{name:32s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
