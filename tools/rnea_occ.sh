#!/bin/bash
# inverse dynamics (fp32 chain kernels) against resident wavefronts per CU: bench.py ms per step (profiles/r6_rnea_occupancy.txt).  GPU box, repo root.
for w in 16 12 10 8 6; do
  GRBDA_WAVES_PER_CU_RNEA32=$w python3 bench.py --workload mit_humanoid --algo rnea --steps 50 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('mit rnea f32 waves per CU $w:', round(d['ms_per_step'],4), 'ms', d['roofline']['kernel'])"
done
for w in 16 12 8; do
  GRBDA_WAVES_PER_CU_RNEA32=$w python3 bench.py --workload tello --algo rnea --steps 30 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('tello rnea f32 waves per CU $w:', round(d['ms_per_step'],4), 'ms', d['roofline']['kernel'])"
done
