cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/pmc3; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $P/s1 -o p -- python3 bench.py --no-cpu --steps 1 --warmup 1 > $P/s1.log 2>&1
f=$(find $P/s1 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.OrderedDict()
for r in rows:
    if "nn_search_kernel<0>" in r["Kernel_Name"]:
        per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(per)[-20:]
print("pass  valu_insts(M)  lanes_per_inst  valu_issue_frac")
for n, i in enumerate(ids):
    d = per[i]
    inst = d["SQ_INSTS_VALU"]; thr = d["SQ_THREAD_CYCLES_VALU"]; act = d["SQ_ACTIVE_INST_VALU"]; gui = d["GRBM_GUI_ACTIVE"] / 8
    print("%2d %10.1f %10.1f %10.2f" % (n + 1, inst / 1e6, thr / act * 1.0, inst * 4 / (1024 * gui)))
PY
tail -2 $P/s1.log | cut -c1-200
rm -rf $P/s1
