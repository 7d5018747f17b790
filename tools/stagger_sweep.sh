#!/bin/bash
# experiment: odd-slot wavefronts start late (GRBDA_CHAIN_DEBUG bits 8.. of an ablation build: units of 127 x 64 clocks ~ 3.9 us at 2.1 GHz)
cd ${GRAFT_REPO_ROOT:-.}
export GRBDA_HIP_LIB=$PWD/build/variants/libgrbda_hip_abl.so
for rep in 1 2; do
for d in 0 1 2 3 4 6 8; do
  export GRBDA_CHAIN_DEBUG=$((d * 256))
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('delay units $d', '%.4g evals/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], d.get('verified'))"
done; done
