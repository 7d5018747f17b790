#!/bin/bash
# A/B of chain-kernel build variants (make variant ...): tools/ab_variants.sh "<bench args>" lib...
set -u
cd ${GRAFT_REPO_ROOT:-.}
args=$1; shift
for lib in "$@"; do
  export GRBDA_HIP_LIB=$PWD/build/variants/libgrbda_hip_$lib.so
  for w in mit_humanoid jvrc1_humanoid mini_cheetah tello; do
    python bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload $w $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$w', '$args', '%.4g evals/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], d.get('verified'))"
  done
done
