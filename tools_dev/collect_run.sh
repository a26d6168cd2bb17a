#!/bin/bash
# Produces gpurun_out/final{,_prof} on the GPU box; tools_dev/collect_profiles.py <round> then files them under profiles/<round>.
#   gpurun --timeout 2400 -- 'bash tools_dev/collect_run.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; P=gpurun_out/final_prof
rm -rf $O $P; mkdir -p $O $P
python3 bench.py --extras --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json
python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu 2>/dev/null | grep "^{\"metric" | tail -1 > $O/bench_2ranks_one_gpu.json
./tools_dev/ubench/valu_rates > $O/valu_rates.txt 2>&1
python3 bench.py --algorithm icp --no-cpu 2>/dev/null | tail -1 > $O/bench_p2p.json
python3 bench.py --pairs 32 --points 1000000 --iters 50 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $O/bench_1M_50it.json
python3 bench_map.py 2>/dev/null | tail -1 > $O/bench_map.json
python3 bench_map.py --scans 640 --no-cpu 2>/dev/null | tail -1 > $O/bench_map_640.json
python3 tools_dev/fixture.py > $O/fixture_batch.txt 2>&1
PROFILE=2 python3 tools_dev/fixture.py > $O/fixture_batch_profile2.txt 2>&1
python3 tools_dev/ndt_try.py > $O/ndt_try.txt 2>&1
python3 tools_dev/plane_time.py > $O/plane_time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o b -- python3 bench.py --no-cpu --no-single --no-real 2>/dev/null | grep '^{"metric' | tail -1 > $P/bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/FETCH_SIZE -o p -- python3 bench.py --no-cpu --no-single --no-real --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/WRITE_SIZE -o p -- python3 bench.py --no-cpu --no-single --no-real --steps 2 --warmup 1 > /dev/null 2>&1
# round 4: BASELINE configs[4]'s per-GPU share under the kernel trace (why is its first pass 2.2x worse per query?)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats1M -o b -- python3 bench.py --pairs 32 --points 1000000 --iters 50 --steps 2 --warmup 1 --no-cpu --no-single --no-real 2>/dev/null | grep '^{"metric' | tail -1 > $P/bench_1M_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $P/map -o m -- python3 bench_map.py --no-cpu > /dev/null 2>&1
# rocprofv3 nests its output under a host-name directory: flatten
for d in stats FETCH_SIZE WRITE_SIZE map stats1M; do find $P/$d -mindepth 2 -type f -exec mv {} $P/$d/ \; ; done
# the counter CSVs are large: keep the columns collect_profiles.py reads
for c in FETCH_SIZE WRITE_SIZE; do
  python3 - "$P/$c/p_counter_collection.csv" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
with open(p, "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
    for r in rows:
        w.writerow([r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"]])
PY
done
rm -f $P/*/*_agent_info.csv $P/*/*kernel_trace.csv
ls -la $O $P/*
