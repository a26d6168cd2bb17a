#!/bin/bash
# kernel statistics of the default bench and of a single pair (rocprofv3 --kernel-trace --stats); results under gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/${1:-prof}; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o b -- python3 bench.py --no-cpu --no-single --steps 8 --warmup 2 2>/dev/null | grep '^{"metric' | tail -1 > $P/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $P/single -o s -- python3 tools_dev/single.py > $P/single.log 2>&1
for d in stats single; do find $P/$d -mindepth 2 -type f -exec mv {} $P/$d/ \; ; done
rm -f $P/*/*_agent_info.csv
# keep the traces small: the single-pair trace of the LAST registration only
python3 - $P <<'PY'
import csv, sys, os
P = sys.argv[1]
f = os.path.join(P, "single", "s_kernel_trace.csv")
if os.path.exists(f):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = [i for i, r in enumerate(rows) if "k_bbox<0>" in r["Kernel_Name"]]
    start = last[-1] if last else 0               # one bbox kernel per registration (its first kernel)
    sel = rows[start:]
    t0 = int(sel[0]["Start_Timestamp"])
    with open(os.path.join(P, "single_last_registration_trace.csv"), "w") as o:
        o.write("kernel,start_us,duration_us,gap_before_us\n")
        prev_end = t0
        for r in sel:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o.write('"%s",%.2f,%.2f,%.2f\n' % (r["Kernel_Name"][:50], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
            prev_end = e
    os.remove(f)
f = os.path.join(P, "stats", "b_kernel_trace.csv")
if os.path.exists(f): os.remove(f)
PY
ls -la $P $P/*
