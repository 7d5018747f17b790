#!/bin/bash
# Everything the round's profiles/ directory is made of, in one GPU call (run on the GPU box from the repo root):
#   bench lines of every BASELINE workload, rocprofv3 kernel stats, HBM-side traffic, instruction counters (flops),
#   FETCH_SIZE / WRITE_SIZE calibration, derivative pipeline timings and counters.   usage: tools/round_profiles.sh r3
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${1:-r3}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
# the sha of the kernel sources the counters below are MEASURED on (tools/collect_profiles.py stamps the JSON files with this value, not
# with the sources at collection time)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from bench import kernel_sources_sha; print(kernel_sources_sha())" > $OUT/kernel_sources_sha.txt
bash tools/bench_all.sh $R/bench > $OUT/bench_all.log 2>&1
bash tools/profile_round.sh $R/prof > $OUT/profile_round.log 2>&1
# instruction counters: the sets with the floating-point instruction classes (1: f32, 5-6: f64) and the cycle counters (3-4)
PMC_TO=4 bash tools/pmc_run.sh $R/pmc_mit_aba32 aba 32 > /dev/null 2>&1
PMC_TO=1 bash tools/pmc_run.sh $R/pmc_mit_rnea32 rnea 32 > /dev/null 2>&1
PMC_TO=1 PMC_MODEL=tello PMC_BATCH=1048576 bash tools/pmc_run.sh $R/pmc_tello_aba32 aba 32 > /dev/null 2>&1
PMC_TO=1 PMC_MODEL=jvrc1_humanoid PMC_BATCH=1048576 bash tools/pmc_run.sh $R/pmc_jvrc1_aba32 aba 32 > /dev/null 2>&1
PMC_FROM=5 PMC_TO=6 PMC_MODEL=mini_cheetah PMC_BATCH=65536 bash tools/pmc_run.sh $R/pmc_minicheetah_aba64 aba 64 > /dev/null 2>&1
PMC_TO=4 PMC_MODEL=four_bar PMC_BATCH=1048576 bash tools/pmc_run.sh $R/pmc_four_bar_aba32 aba 32 > /dev/null 2>&1
PMC_TO=4 PMC_MODEL=six_bar PMC_BATCH=1048576 bash tools/pmc_run.sh $R/pmc_six_bar_aba32 aba 32 > /dev/null 2>&1
python3 tools/pmc_flops.py $OUT/pmc_flops.json mit_humanoid:aba:f32:262144:$OUT/pmc_mit_aba32/summary.txt \
    mit_humanoid:rnea:f32:262144:$OUT/pmc_mit_rnea32/summary.txt tello:aba:f32:1048576:$OUT/pmc_tello_aba32/summary.txt \
    jvrc1_humanoid:aba:f32:1048576:$OUT/pmc_jvrc1_aba32/summary.txt mini_cheetah:aba:f64:65536:$OUT/pmc_minicheetah_aba64/summary.txt \
    four_bar:aba:f32:1048576:$OUT/pmc_four_bar_aba32/summary.txt six_bar:aba:f32:1048576:$OUT/pmc_six_bar_aba32/summary.txt > /dev/null
# counter calibration on known byte counts
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/calib_$c -- $ROOT/build/tools/traffic_calib > $OUT/calib_$c.log 2>&1
done
python3 - "$OUT" > $OUT/traffic_calibration.txt <<PY
import csv, glob, os, sys
out = sys.argv[1]
print("rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB as reported) against 1 GiB (1048576 KiB) moved per kernel, mean of 3 launches")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = {}
    for f in glob.glob(os.path.join(out, f"calib_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc.setdefault(k, {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()):
        m = sum(v.values()) / len(v)
        print(f"  {c:11s} {k:16s} {m:12.0f} KiB   counted / moved = {m / 1048576:.3f}")
PY
bash $ROOT/tools/pmc_derivs.sh $R/pmc_derivs > /dev/null 2>&1
cd $ROOT
python3 tools/strong_scaling_proxy.py > $OUT/strong_scaling_proxy.txt 2>/dev/null
python3 tools/lm_waves_ab.py mit_humanoid mini_cheetah > $OUT/latency_mode_waves.txt 2>/dev/null
python3 tools/gate_f32_oracle.py 1048576 > $OUT/gate_f32_oracle.txt 2>/dev/null
python3 tools/stress_random_models.py 250 > $OUT/stress_random_models.txt 2>/dev/null
python3 tools/tello_acc2.py generalized_rbda_amd/libgrbda_hip.so 2>/dev/null | tr "\n" " " > $OUT/tello_acc.txt
cat $OUT/traffic_calibration.txt
