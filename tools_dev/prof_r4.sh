#!/bin/bash
# round 4: rocprofv3 kernel stats of tools_dev/r4.py (arguments = the debug_flags variants), summary to stdout;
# TRACE=substring also prints the durations of the kernels whose name contains it, in launch order (last 60)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/r4_prof; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python3 tools_dev/r4.py "$@" > gpurun_out/r4_prof.log 2>&1
find $P -mindepth 2 -type f -exec mv {} $P/ \;
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open('gpurun_out/r4_prof/r_kernel_stats.csv')))
for r in rows[:40]:
    print('%-70s calls %6s avg %10.1f us total %8.2f ms %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
key = os.environ.get('TRACE')
if key:
    tr = list(csv.DictReader(open('gpurun_out/r4_prof/r_kernel_trace.csv')))
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    sel = [r for r in tr if any(k in r['Kernel_Name'] for k in key.split(','))][-60:]
    for r in sel:
        print('%-40s %8.1f us  grid %s' % (r['Kernel_Name'][:40].replace('void s3d::', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', ''))))
PY
rm -f $P/*kernel_trace.csv $P/*agent_info.csv
