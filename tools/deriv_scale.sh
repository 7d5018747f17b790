#!/bin/bash
# Per-kernel times of the derivative pipeline at several batch sizes (is the per-state cost flat in B?).  GPU box, repo root.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-deriv_scale}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for B in 131072 262144 524288 1048576; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$B -- python3 $ROOT/tools/time_derivs.py jvrc1_humanoid $B > $OUT/t$B.txt 2> $OUT/s$B.log
  f=$(find $OUT/s$B -name "*kernel_stats.csv" | head -1)
  echo "== B=$B"; cat $OUT/t$B.txt
  python3 - "$f" $B <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void grbda_hip::", "")
    if any(k in n for k in ("deriv", "spd", "crba", "unpack")):
        print("  %-50s calls %3s avg %9.3f ms  %7.2f ns/state" % (n, r["Calls"], float(r["AverageNs"]) * 1e-6, float(r["AverageNs"]) / int(sys.argv[2])))
PY
done
