#!/bin/bash
# round 6: the counters behind roofline.valu (bench.py) - VALU issue slots and active lanes of the NN search passes, K4 and
# K6 on the default batch.  Separate --pmc passes (kernel trace only beside them), summary -> gpurun_out/pmc_valu/;
# tools_dev/pmc_valu.py files it as profiles/<round>/pmc_valu_summary.csv + profiles/nn_valu.json.
#   gpurun --timeout 1100 -- 'bash tools_dev/pmc_valu.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/pmc_valu; rm -rf $P; mkdir -p $P
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/s$i -o p -- python3 bench.py --no-cpu --no-single --no-real --no-search-frac --steps 3 --warmup 1 > $P/s$i.log 2>&1
  f=$(find $P/s$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$P/set$i.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in rows:
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = r.get("Dispatch_Id")
    if d not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
        seen.add(d); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k, d in agg.items():
        if any(t in k for t in ("nn_", "knn", "gicp_accumulate")):
            for c, v in d.items():
                f.write('"%s",%s,%.1f,%d\n' % (k.replace("void ", "").replace("s3d::", "")[:60], c, sum(v) / len(v), len(v)))
            if dur[k]:
                f.write('"%s",DURATION_NS_UNDER_PMC,%.1f,%d\n' % (k.replace("void ", "").replace("s3d::", "")[:60], sum(dur[k]) / len(dur[k]), len(dur[k])))
PY
  else tail -5 $P/s$i.log > $P/set$i.err; fi
  grep '^{"metric' $P/s$i.log | tail -1 > $P/bench_set$i.json
  rm -rf $P/s$i
done
ls -la $P; head -5 $P/set*.csv; cat $P/*.err 2>/dev/null
