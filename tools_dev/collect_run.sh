#!/bin/bash
# Produces gpurun_out/final{,_prof} on the GPU box; tools_dev/collect_profiles.py <round> then files them under profiles/<round>.
# Three calls (a gpurun call lasts at most 20 minutes):
#   gpurun --timeout 1150 -- 'bash tools_dev/collect_run.sh A'     bench lines and dev-tool outputs
#   gpurun --timeout 1150 -- 'bash tools_dev/collect_run.sh B'     rocprofv3 stats + FETCH / WRITE counter passes of the driver's command
#   gpurun --timeout 1150 -- 'bash tools_dev/collect_run.sh C'     1 M x 50 stats, map stats, real-scan traces, the VALU counter passes
PART=${1:-A}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; P=gpurun_out/final_prof
mkdir -p $O $P
if [ "$PART" = A ]; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default.json     # the driver's command
python3 bench.py --extras --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > $O/bench_default_extras.json
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
python3 tools_dev/real_single.py > $O/real_single_pair.txt 2>&1
python3 tools_dev/real_split.py 1 2 3 1 2 > $O/real_batch_split.txt 2>&1
python3 tools_dev/real_grid_budget.py 2 3 > $O/real_grid_budget.txt 2>&1
NPAIRS=256 SINGLE=1 python3 tools_dev/r4.py 0 0x10000000 0x80000000 0x200 > $O/r4_variants.txt 2>&1
fi
if [ "$PART" = B ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o b -- python3 bench.py --no-cpu --no-single --no-real --no-search-frac 2>/dev/null | grep '^{"metric' | tail -1 > $P/bench_under_rocprof.json
# round 5: the counter passes run the DRIVER's command (every leg of the default line); collect_profiles.py keeps the
# launches of the batch workload by their grid size
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/FETCH_SIZE -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/WRITE_SIZE -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1
for d in stats FETCH_SIZE WRITE_SIZE; do find $P/$d -mindepth 2 -type f -exec mv {} $P/$d/ \; ; done
# the counter CSVs are large: keep the columns collect_profiles.py reads
for c in FETCH_SIZE WRITE_SIZE; do
  python3 - "$P/$c/p_counter_collection.csv" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
order = next((k for k in ("Dispatch_Id", "Start_Timestamp") if rows and k in rows[0]), None)
with open(p, "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size"] + ([order] if order else []))
    for r in rows:
        w.writerow([r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"], r.get("Grid_Size", r.get("Grid_Size_X", ""))] + ([r[order]] if order else []))
PY
done
rm -f $P/*/*_agent_info.csv $P/*/*kernel_trace.csv
fi
if [ "$PART" = C ]; then
# round 4: BASELINE configs[4]'s per-GPU share under the kernel trace (why is its first pass 2.2x worse per query?)
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats1M -o b -- python3 bench.py --pairs 32 --points 1000000 --iters 50 --steps 2 --warmup 1 --no-cpu --no-single --no-real --no-search-frac 2>/dev/null | grep '^{"metric' | tail -1 > $P/bench_1M_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $P/map -o m -- python3 bench_map.py --no-cpu > /dev/null 2>&1
for d in map stats1M; do find $P/$d -mindepth 2 -type f -exec mv {} $P/$d/ \; ; done
rm -f $P/*/*_agent_info.csv $P/*/*kernel_trace.csv
# round 5: the reference's actual call (one registration of two real scans, defaults) as a kernel trace; the 96-pair batch on
# 192 distinct real clouds as kernel stats
bash tools_dev/prof_trace.sh tools_dev/real_single.py > $O/real_single_pair_trace.txt 2>&1
bash tools_dev/prof_stats.sh tools_dev/real_batch.py 0 > $O/real_batch_distinct_kernel_stats.txt 2>&1
# round 6: the counters behind roofline.valu (tools_dev/pmc_valu.py files them)
bash tools_dev/pmc_valu.sh > $O/pmc_valu.log 2>&1
fi
ls -la $O $P/* 2>/dev/null | tail -60
