#!/bin/bash
# per-kernel times of the derivative pipeline (rocprofv3 kernel trace) for one or more builds of the library
# usage: tools/deriv_kernel_times.sh OUTDIR [lib ...]    (lib: path of a libgrbda .so; default: the in-tree build)
out=gpurun_out/$1; shift
mkdir -p $out
libs="$@"; [ -z "$libs" ] && libs=generalized_rbda_amd/libgrbda_hip.so
export TMPDIR=/tmp
for lib in $libs; do
  name=$(basename $lib .so)
  GRBDA_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o t -- python3 tools/time_derivs.py ${MODEL:-jvrc1_humanoid} ${BATCH:-131072} > $out/$name.log 2>&1
  f=$(find $out/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; grep -v amdgpu.ids $out/$name.log | tail -2
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
done
