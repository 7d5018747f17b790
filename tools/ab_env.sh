# tools/ab_env.sh "<bench args>" "<lib> <ENV=..>" ... : bench lines under library variants and environment settings
cd ${GRAFT_REPO_ROOT:-.}
a=$1; shift
for rep in 1 2; do for cfg in "$@"; do
  lib=${cfg%% *}; envs=${cfg#* }
  env GRBDA_HIP_LIB=$PWD/build/variants/libgrbda_hip_$lib.so $envs python bench.py --steps 30 --warmup 5 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$a', '%.4g evals/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], d.get('verified'))"
done; done
