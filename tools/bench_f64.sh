#!/bin/bash
# fp64 workloads through bench.py (ms per step), for before / after comparisons of fp64-only changes; GPU box, repo root.
mkdir -p gpurun_out/f64_workloads
for spec in "mit_humanoid --algo aba --dtype f64" "mit_humanoid --algo rnea --dtype f64" "mini_cheetah --algo aba" "jvrc1_humanoid --algo aba --dtype f64" "six_bar --algo aba --dtype f64"; do
  python3 bench.py --workload $spec --steps 30 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config'], d['dtype'], d['ms_per_step'], d.get('verified'), d.get('verify_max_rel_err'))"
done | tee gpurun_out/f64_workloads/${1:-after}.txt
