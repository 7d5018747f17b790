#!/bin/bash
# Round profile artifacts (run on the GPU box from the repo root): rocprofv3 kernel stats of the default
# bench command, HBM-side traffic counters, and the SQ instruction-mix passes.  Output under gpurun_out/$1.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for algo in aba rnea; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$algo -- python3 $ROOT/tools/pmc_target.py $algo 32 > $OUT/fetch_$algo.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$algo -- python3 $ROOT/tools/pmc_target.py $algo 32 > $OUT/write_$algo.log 2>&1
done
python3 - <<PY
import csv, glob, os
out = "$OUT"
for algo in ("aba", "rnea"):
    for name in ("fetch", "write"):
        vals = {}
        for f in glob.glob(os.path.join(out, f"{name}_{algo}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "grbda" in r["Kernel_Name"]:
                    vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        if vals:
            print(f"{algo} {name.upper()}_SIZE n={len(vals)} mean={sum(vals.values())/len(vals):.4e}")
PY
