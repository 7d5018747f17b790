#!/bin/bash
# A/B: the fp32 chain kernel at 2 (default), 3 (variant w3: make variant VARIANT=w3 VFLAGS=-DGRBDA_EXP_WIDE_WPS=3) and 4 wavefronts per SIMD
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { python3 bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$2', '$1', 'value %.4g' % d['value'], 'ms %.4f' % d['ms_per_step'], d['roofline'].get('kernel','')[:50])"; }
for rep in 1 2; do
for w in mit_humanoid jvrc1_humanoid revolute_rotor_chain; do
  run $w "2 waves"
  GRBDA_HIP_LIB=build/variants/libgrbda_hip_w3.so GRBDA_CHAIN_WIDE=1 run $w "3 waves"
  GRBDA_CHAIN_WIDE=1 run $w "4 waves"
done
done
