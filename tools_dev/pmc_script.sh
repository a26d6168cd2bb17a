#!/bin/bash
# round 6: PMC passes over any python script; per-kernel means for kernels matching FILTER (regex).  usage:
#   FILTER="knn|rings" tools_dev/pmc_script.sh tools_dev/real_batch.py 0
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/pmc; rm -rf $P; mkdir -p $P
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/s$i -o p -- python3 "$@" > $P/s$i.log 2>&1
  f=$(find $P/s$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then FILTER="${FILTER:-knn}" python3 - "$f" "$P/set$i.csv" <<'PY'
import csv, sys, collections, os, re
rows = csv.DictReader(open(sys.argv[1]))
flt = re.compile(os.environ["FILTER"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].replace("void s3d::", "")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for k, d in agg.items():
        if flt.search(k):
            for c, v in d.items():
                f.write('%-50s %-32s %16.1f x%d\n' % (k, c, sum(v) / len(v), len(v)))
PY
  else tail -3 $P/s$i.log > $P/set$i.err; fi
  rm -rf $P/s$i
done
cat $P/set*.csv $P/*.err 2>/dev/null | head -80
