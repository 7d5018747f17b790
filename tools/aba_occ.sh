#!/bin/bash
# forward dynamics (fp32 headline kernel) against resident wavefronts per CU; bench.py kernel ms.  GPU box, repo root.
for wl in mit_humanoid jvrc1_humanoid; do
for w in 12 10 9 8 7 6 5 4; do
  GRBDA_WAVES_PER_CU_ABA32=$w python3 bench.py --workload $wl --algo aba --steps 50 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$wl aba f32 waves per CU $w:', round(d['roofline']['kernel_ms'],4), 'ms')"
done
done
