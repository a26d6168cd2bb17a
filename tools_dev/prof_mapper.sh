#!/bin/bash
# kernel trace of the LAST batch call of tools_dev/mapper_one.py (one new scan against 8 cached ones); result: gpurun_out/$1/mapper_last_call_trace.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/${1:-prof_mapper}; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P/m -o m -- python3 tools_dev/mapper_one.py > $P/mapper.log 2>&1
find $P/m -mindepth 2 -type f -exec mv {} $P/m/ \;
python3 - $P <<'PY'
import csv, sys, os
P = sys.argv[1]
f = os.path.join(P, "m", "m_kernel_trace.csv")
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "k_bbox<0>" in r["Kernel_Name"]]
sel = rows[last[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
with open(os.path.join(P, "mapper_last_call_trace.csv"), "w") as o:
    o.write("kernel,start_us,duration_us,gap_before_us\n")
    prev_end = t0
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write('"%s",%.2f,%.2f,%.2f\n' % (r["Kernel_Name"][:50], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
os.remove(f)
PY
rm -rf $P/m
cat $P/mapper.log | tail -6
