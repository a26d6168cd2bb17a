#!/bin/bash
# round 5: rocprofv3 kernel stats of any python script; top kernels to stdout.  usage: tools_dev/prof_stats.sh script.py [args ...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/stats_prof; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python3 "$@" > gpurun_out/stats_prof.log 2>&1
find $P -mindepth 2 -type f -exec mv {} $P/ \;
grep -E "^flags|call " gpurun_out/stats_prof.log | tail -4
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/stats_prof/r_kernel_stats.csv')))
for r in rows[:28]:
    print('%-66s calls %6s avg %10.1f us total %8.2f ms %5s%%' % (r['Name'].replace('void s3d::','').replace('s3d::','')[:66], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
rm -f $P/*kernel_trace.csv $P/*agent_info.csv
