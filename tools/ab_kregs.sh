#!/bin/bash
# A/B of the [K | y0] register file of the 2-wavefront fp32 chain kernel (chain_kernels.hip, KFile): variants built with
# make variant VARIANT=k<N> VFLAGS=-DGRBDA_CHAIN_KREGS=<N>; the plan is told the same N through GRBDA_CHAIN_KREGS.
# usage (GPU box, repo root): tools/ab_kregs.sh "0 6 8 12"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for k in $1; do
  for w in "mit_humanoid" "jvrc1_humanoid" "mini_cheetah"; do
    GRBDA_HIP_LIB=build/variants/libgrbda_hip_k$k.so GRBDA_CHAIN_KREGS=$k python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('kregs $k', '$w', 'value %.4g' % d['value'], 'ms %.4f' % d['ms_per_step'], d['roofline'].get('kernel','')[:50])"
  done
done
