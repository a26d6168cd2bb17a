#!/bin/bash
# round 5: rocprofv3 kernel trace of a python script; prints the kernels of the LAST call in launch order with their
# durations and the gaps between them.  usage: tools_dev/prof_trace.sh tools_dev/real_single.py [marker-kernel]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/trace; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P -o r -- python3 "$1" > gpurun_out/trace.log 2>&1
find $P -mindepth 2 -type f -exec mv {} $P/ \;
tail -3 gpurun_out/trace.log
MARK="${2:-k_bbox}" python3 - <<'PY'
import csv, os
tr = list(csv.DictReader(open('gpurun_out/trace/r_kernel_trace.csv')))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
mark = os.environ['MARK']
starts = [i for i, r in enumerate(tr) if mark in r['Kernel_Name']]
# the last complete call: from the second-to-last marker to the last marker
if len(starts) < 2:
    raise SystemExit('marker kernel %r occurs %d time(s) in the trace: need two calls (the last complete one lies between them)' % (mark, len(starts)))
a, b = starts[-2], starts[-1]
prev_end = None
tot = 0
for r in tr[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%-46s %8.1f us  gap %6.1f us  grid %s' % (r['Kernel_Name'].replace('void s3d::', '').replace('s3d::', '')[:46], (e - s) / 1e3, gap, r.get('Grid_Size_X', r.get('Grid_Size', ''))))
    prev_end = e; tot += (e - s) / 1e3
print('kernels %d  busy %.1f us  span %.1f us' % (b - a, tot, (int(tr[b-1]['End_Timestamp']) - int(tr[a]['Start_Timestamp'])) / 1e3))
PY
rm -rf $P
