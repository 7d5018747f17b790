# same-run A/B of library variants: tools/ab3.sh "lib lib ..." "bench args" "bench args" ...
cd ${GRAFT_REPO_ROOT:-.}
libs=$1; shift
for a in "$@"; do
for rep in 1 2; do
for lib in $libs; do
  export GRBDA_HIP_LIB=$PWD/build/variants/libgrbda_hip_$lib.so
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s' % '$lib', '$a', '%.4g evals/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], d.get('verified'))"
done; done; done
