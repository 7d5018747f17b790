#!/bin/bash
# A/B of derivative-kernel build variants (build/exp/libgrbda_<name>.so) at several grids of the recursion kernel.
# usage (GPU box, repo root): tools/ab_deriv.sh "name1 name2 ..." "waves1 waves2 ..." [B]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for n in $1; do
  for w in $2; do
    echo "== $n GRBDA_DERIV_WAVES_PER_CU=$w"
    GRBDA_LIB=build/exp/libgrbda_$n.so GRBDA_DERIV_WAVES_PER_CU=$w python3 tools/time_derivs.py jvrc1_humanoid ${3:-262144} 2>&1 | grep float32
  done
done
