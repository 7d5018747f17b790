#!/bin/bash
# fp64 forward dynamics against wavefronts per CU x LDS per wavefront (the slab shrinks as LDS grows): bench.py ms per step.  GPU box, repo root.
mkdir -p gpurun_out/f64_sweep
for wl in mit_humanoid jvrc1_humanoid mini_cheetah; do
for cfg in "8 20480" "7 23040" "6 26880" "5 32000" "4 40960" "4 32768" "6 20480"; do
  set -- $cfg
  GRBDA_NO_LATENCY_MODE=1 GRBDA_WAVES_PER_CU_ABA64=$1 GRBDA_LDS_BYTES_PER_WAVE_ABA64=$2 python3 bench.py --workload $wl --algo aba --dtype f64 --steps 30 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$wl', 'waves per CU $1 LDS per wavefront $2:', round(d['ms_per_step'],4), 'ms', d.get('verified'))"
done
done | tee gpurun_out/f64_sweep/sweep.txt
