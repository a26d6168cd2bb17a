#!/bin/bash
# extra PMC passes for the NN kernel (evidence for the latency / L1-bound reading); results under gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/pmc; rm -rf $P; mkdir -p $P
i=0
QUICK=${1:-}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  if [ "$QUICK" = quick ] && [ $i -gt 2 ]; then break; fi
  if [ "$QUICK" = mem ] && [ $i -le 2 ]; then continue; fi
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/s$i -o p -- python3 bench.py --no-cpu --no-single --no-real --steps 2 --warmup 1 > $P/s$i.log 2>&1
  f=$(find $P/s$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$P/set$i.csv" <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k, d in agg.items():
        if "nn_search" in k or "nn_first" in k or "knn" in k or "gicp_accumulate" in k or "onesweep" in k or "scan27" in k:
            for c, v in d.items():
                f.write('"%s",%s,%.1f,%d\n' % (k, c, sum(v) / len(v), len(v)))
PY
  else tail -3 $P/s$i.log > $P/set$i.err; fi
  rm -rf $P/s$i
done
cat $P/set*.csv $P/*.err 2>/dev/null | head -60
