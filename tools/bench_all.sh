#!/bin/bash
# Every BASELINE workload through bench.py (one JSON line each), then the derivative timings of config 5.
# usage (GPU box, repo root): tools/bench_all.sh OUTDIR
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-bench}
mkdir -p $OUT
cd $ROOT
FAILED=0
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { echo "bench.py (headline) failed: rc $?" >&2; FAILED=1; }
: > $OUT/bench_other_workloads.jsonl
for spec in "mit_humanoid rnea" "mit_humanoid aba --dtype f64" "mit_humanoid rnea --dtype f64" "mini_cheetah aba" "mini_cheetah rnea" \
            "jvrc1_humanoid aba" "jvrc1_humanoid rnea" "tello aba" "tello rnea" "revolute_rotor_chain aba" \
            "four_bar aba" "four_bar rnea" "six_bar aba" "six_bar rnea" "four_bar aba --dtype f64" "six_bar aba --dtype f64"; do
  set -- $spec
  w=$1; a=$2; shift 2
  python3 bench.py --workload $w --algo $a "$@" --steps 30 --warmup 3 --no-cpu-baseline >> $OUT/bench_other_workloads.jsonl 2>> $OUT/bench_other.err \
    || { echo "bench.py $w $a $* failed: rc $?" >&2; FAILED=1; }
done
python3 tools/time_derivs.py jvrc1_humanoid 1048576 > $OUT/derivatives_timing.txt 2>/dev/null
python3 tools/time_derivs.py mit_humanoid 262144 >> $OUT/derivatives_timing.txt 2>/dev/null
python3 tools/time_derivs.py mini_cheetah 65536 >> $OUT/derivatives_timing.txt 2>/dev/null
# models with implicit clusters: the analytic route through the spanning tree against the difference batches (GRBDA_NO_MANIFOLD=1)
python3 tools/manifold_check.py 262144 > $OUT/manifold_derivatives.txt 2>/dev/null
exit $FAILED
