#!/bin/bash
# dev tool (round 4): where is the GPU idle inside / between the steps of the default bench?  rocprofv3 kernel trace of
# a short bench run; prints the idle gaps > 20 us between consecutive kernels (sorted by start) with their neighbours.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/gaps; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P -o g -- python3 bench.py --steps 4 --warmup 1 --no-single --no-cpu --no-real > gpurun_out/gaps.log 2>&1
find $P -mindepth 2 -type f -exec mv {} $P/ \;
python3 - <<'PY'
import csv
tr = list(csv.DictReader(open('gpurun_out/gaps/g_kernel_trace.csv')))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
prev_end = None; prev = None
busy = 0
out = []
for r in tr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev_end is not None and s - prev_end > 20000:
        out.append((s - prev_end, prev['Kernel_Name'][:50], r['Kernel_Name'][:50], s))
    prev_end = max(prev_end or 0, e); prev = r
t0 = int(tr[0]['Start_Timestamp'])
for g, a, b, s in out[-60:]:
    print('%9.1f us idle at %10.3f ms  after %-50s before %s' % (g / 1e3, (s - t0) / 1e6, a, b))
PY
rm -f $P/*kernel_trace.csv $P/*agent_info.csv
