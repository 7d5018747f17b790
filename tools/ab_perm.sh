#!/bin/bash
# A/B of the permutation-structured link transforms (GRBDA_NO_PERM_LINKS=1: every link on the general path)
set -u
cd ${GRAFT_REPO_ROOT:-.}
for a in "--workload mit_humanoid" "--workload jvrc1_humanoid" "--workload mini_cheetah" "--workload tello" "--workload mit_humanoid --algo rnea" "--workload jvrc1_humanoid --algo rnea" "--workload mit_humanoid --dtype f64"; do
for e in 0 1 0 1; do
  if [ $e = 1 ]; then export GRBDA_NO_PERM_LINKS=1; else unset GRBDA_NO_PERM_LINKS; fi
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', 'general-only=$e', '%.4g evals/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], d.get('verified'))"
done; done
