#!/bin/bash
# kernel statistics of tools_dev/k4t.py (128 pairs of the default batch): gpurun -- 'bash tools_dev/prof_k4t.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/k4t_prof; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o k -- python3 tools_dev/k4t.py > $P/k4t.log 2>&1
f=$(find $P -name "k_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-64s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
