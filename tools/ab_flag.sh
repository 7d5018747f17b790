#!/bin/bash
# A/B of one plan-time environment switch: tools/ab_flag.sh GRBDA_NO_X "<bench args>" ...
cd ${GRAFT_REPO_ROOT:-.}
flag=$1; shift
for a in "$@"; do
for cfg in "A=1" "$flag=1" "A=1" "$flag=1"; do
  env $cfg python bench.py --steps 50 --warmup 5 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', '[$cfg]', '%.4g evals/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], d.get('verified'))"
done; done
