#!/bin/bash
# Round profile artifacts (run on the GPU box from the repo root): rocprofv3 kernel stats of the default bench command and
# of the derivative pipeline, HBM-side traffic counters (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the BASELINE
# workloads.  Output under gpurun_out/$1.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_derivs -- python3 $ROOT/tools/time_derivs.py jvrc1_humanoid 131072 > $OUT/derivs_under_rocprof.txt 2> $OUT/stats_derivs.log
: > $OUT/traffic.txt
for spec in "aba 32 mit_humanoid 262144" "rnea 32 mit_humanoid 262144" "aba 32 tello 1048576" "rnea 32 tello 1048576" \
            "aba 64 mini_cheetah 65536" "aba 32 jvrc1_humanoid 1048576" "aba 64 mit_humanoid 262144" "aba 32 four_bar 1048576" "aba 32 six_bar 1048576"; do
  set -- $spec
  tag=$1_$2_$3
  for c in FETCH_SIZE WRITE_SIZE; do
    PMC_BATCH=$4 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${c}_$tag -- python3 $ROOT/tools/pmc_target.py $1 $2 $3 > $OUT/${c}_$tag.log 2>&1
  done
  python3 - "$OUT" "$tag" "$4" >> $OUT/traffic.txt <<PY
import csv, glob, os, sys
out, tag, batch = sys.argv[1:4]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals, names = {}, set()
    for f in glob.glob(os.path.join(out, f"{c}_{tag}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "grbda" in r["Kernel_Name"] and ("aba" in r["Kernel_Name"] or "rnea" in r["Kernel_Name"]):
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
                names.add(r["Kernel_Name"].split("(")[0])
    res[c] = (sum(vals.values()) / max(1, len(vals)), len(vals), sorted(names))
print(tag, "batch", batch, "FETCH_SIZE_KiB %.4e" % res["FETCH_SIZE"][0], "WRITE_SIZE_KiB %.4e" % res["WRITE_SIZE"][0],
      "launches", res["FETCH_SIZE"][1], "kernel", ";".join(res["FETCH_SIZE"][2]))
PY
done
cat $OUT/traffic.txt
find $OUT/stats $OUT/stats_derivs -name "*kernel_stats.csv" | head
