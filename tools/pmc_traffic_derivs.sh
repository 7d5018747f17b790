#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) and time of the derivative recursion for build variants.
# usage (GPU box): tools/pmc_traffic_derivs.sh OUT "name1 name2"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_td}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in $2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GRBDA_LIB=$ROOT/build/exp/libgrbda_$n.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${n}_$c -- python3 $ROOT/tools/pmc_target_derivs.py > $OUT/${n}_$c.log 2>&1
  done
  python3 - $OUT $n <<PY
import csv, glob, os, sys
out, n = sys.argv[1:3]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = {}
    for f in glob.glob(os.path.join(out, f"{n}_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void grbda_hip::", "")
            if "deriv" in k or "spd" in k:
                acc.setdefault(k, {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(n, c, k[:40], "launches", len(v), "mean %.4e (counter units) per state %.1f" % (sum(v.values()) / len(v), sum(v.values()) / len(v) / 65536 * 1024))
PY
  cd $ROOT; GRBDA_LIB=build/exp/libgrbda_$n.so python3 tools/time_derivs.py jvrc1_humanoid 131072 2>&1 | grep float32; cd /tmp
done
