#!/bin/bash
# the inverse dynamics' staging + epilogue alone (variant library built with -DGRBDA_EXP_RNEA_NO_SEGS: results are wrong) against the whole kernel
for lib in "" "$PWD/build/variants/libgrbda_hip_rnea_nosegs.so"; do
  for wl in mit_humanoid jvrc1_humanoid; do
    GRBDA_HIP_LIB=$lib python3 bench.py --workload $wl --algo rnea --steps 50 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$wl rnea f32', 'staging + epilogue only' if '$lib' else 'whole kernel', round(d['roofline']['kernel_ms'],4), 'ms')"
  done
done
