"""dev tool: turn gpurun_out/final{,_prof} (see profiles/README.md for the commands) into the files under profiles/."""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r2"
R, F, D = (os.path.join(ROOT, p) for p in ("gpurun_out/final_prof", "gpurun_out/final", "profiles/" + ROUND))
os.makedirs(D, exist_ok=True)
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"{R}/{c}/p_counter_collection.csv")))
    if not rows or not {"Counter_Name", "Counter_Value", "Kernel_Name", "Grid_Size"} <= set(rows[0]):
        sys.exit(f"{c}: p_counter_collection.csv is empty or lacks a column this walk reads: {sorted(rows[0]) if rows else []}")
    order_key = next((k for k in ("Dispatch_Id", "Start_Timestamp") if k in rows[0]), None)
    if order_key is None:
        sys.exit(f"{c}: neither Dispatch_Id nor Start_Timestamp in the csv: dispatch order unknown")
    rows.sort(key=lambda r: int(r[order_key]))      # (the per-step walk below needs dispatch order)
    # (round 5: the counter passes run the driver's whole command; the launches of the BATCH workload are those with the
    # largest grid of their kernel - the single-pair / real-scan legs launch the same kernels on smaller grids)
    biggest = collections.defaultdict(int)
    for r in rows:
        if r["Counter_Name"] == c and r.get("Grid_Size"):
            biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == c and (not r.get("Grid_Size") or int(r["Grid_Size"]) == biggest[r["Kernel_Name"]]):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    summ = sorted(((k, sum(v) / len(v), len(v)) for k, v in agg.items()), key=lambda x: -x[1] * x[2])
    with open(f"{D}/rocprofv3_pmc_{c}_summary.csv", "w") as f:
        f.write("Kernel_Name,mean_%s_KB_per_launch,launches\n" % c)
        for k, m, n in summ:
            f.write('"%s",%.1f,%d\n' % (k, m, n))
    # the NN family of the ICP loop, per PASS, over the steps of the BATCH workload only (round 5: the rows are in
    # dispatch order; a batch step runs from its s3d_nn_first_kernel - the launch with the largest grid of that kernel -
    # to the fitness pass s3d_nn_search_kernel<1>, and every family kernel in between belongs to one of its passes)
    FAM = ("nn_search_kernel<0>", "nn_first_kernel", "nn_scan27_kernel", "nn_worklist_kernel", "nn_record_test_kernel",
           "nn_record_touch_kernel", "nn_record_search_kernel")
    seq = [(r["Kernel_Name"], float(r["Counter_Value"]), int(r["Grid_Size"] or 0)) for r in rows if r["Counter_Name"] == c]
    firsts = [g for k, _, g in seq if "nn_first_kernel" in k]
    if not firsts:
        sys.exit(f"{c}: no s3d_nn_first_kernel launch in the counter rows (kernel renamed?)")
    G = max(firsts)
    steps, cur, npass = [], None, 0
    STARTS = ("nn_first_kernel", "nn_scan27_kernel", "nn_record_test_kernel", "nn_record_touch_kernel<false>", "nn_search_kernel<0>")
    for k, v, g in seq:
        if "nn_first_kernel" in k and g == G:
            cur, npass = 0.0, 0
        if cur is None:
            continue
        if "nn_search_kernel<1>" in k:
            steps.append(cur / max(npass, 1)); cur = None
            continue
        if any(t in k for t in FAM):
            cur += v
            npass += 1 if any(t in k for t in STARTS) else 0
    # the first 25 batch steps of the run are the driver command's 5 warm-up + 20 timed steps; later legs of the line launch the
    # batch again (profile runs; two_in_flight, whose two contexts interleave their launches): not walked
    bj = json.load(open(f"{R}/bench_under_rocprof.json")) if os.path.exists(f"{R}/bench_under_rocprof.json") else {}
    n_driver = int(bj.get("steps", 20)) + int(bj.get("warmup", 5))      # the driver command's warm-up + timed steps
    if len(steps) < n_driver:
        sys.exit(f"{c}: {len(steps)} batch steps found (first kernel at the largest grid ... fitness pass), {n_driver} expected")
    steps = steps[:n_driver]
    out[c] = sum(steps) / len(steps)
    print(c, "batch steps", len(steps), "passes per step", npass, "KB per pass", round(out[c], 1))
hbm = int(2 * out["FETCH_SIZE"] * 1024 + out["WRITE_SIZE"] * 1024)
json.dump({"kernel": "s3d_nn_first_kernel + s3d_nn_scan27_kernel + s3d_nn_worklist_kernel + s3d_nn_search_kernel<0> + s3d_nn_record_{test,touch,search}_kernel (per pass of the ICP loop)", "fetch_size_kb_per_launch": round(out["FETCH_SIZE"], 1),
           "write_size_kb_per_launch": round(out["WRITE_SIZE"], 1), "hbm_bytes_per_launch": hbm,
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command; batch launches selected by grid size)",
           "workload": "256 pairs x 100k points, 20 iterations (bench default)",
           "kernel_src_sha256": kernel_source_hash(), "round": ROUND},
          open(os.path.join(ROOT, "profiles/nn_traffic.json"), "w"), indent=1)
for f in os.listdir(F):
    shutil.copy(f"{F}/{f}", f"{D}/{f}")
shutil.copy(f"{R}/stats/b_kernel_stats.csv", f"{D}/rocprofv3_kernel_stats_bench_default.csv")
shutil.copy(f"{R}/bench_under_rocprof.json", f"{D}/bench_under_rocprof.json")
shutil.copy(f"{R}/map/m_kernel_stats.csv", f"{D}/rocprofv3_kernel_stats_bench_map.csv")
if os.path.exists(f"{R}/stats1M/b_kernel_stats.csv"):
    shutil.copy(f"{R}/stats1M/b_kernel_stats.csv", f"{D}/rocprofv3_kernel_stats_bench_1M_50it.csv")
    shutil.copy(f"{R}/bench_1M_under_rocprof.json", f"{D}/bench_1M_under_rocprof.json")
d = json.load(open(f"{D}/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_search_passes"), d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"],
      d["cpu_baseline"]["value"], d["cpu_baseline_parallel"]["value"], d["single_pair"], d["stage_ms"], d["nn_launch_ms"][:6])
d = json.load(open(f"{D}/bench_under_rocprof.json")); print("rocprof", d["value"], d["roofline"]["avg_launch_ms"])
rows = list(csv.DictReader(open(f"{D}/rocprofv3_kernel_stats_bench_default.csv")))
for r in rows[:8]:
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), r["Percentage"])
print("p2p", json.load(open(f"{D}/bench_p2p.json"))["value"])
d = json.load(open(f"{D}/bench_1M_50it.json")); print("1M", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["nn_launch_ms"][:6])
d = json.load(open(f"{D}/bench_map.json")); print("map", d["value"], d["ms_per_step"], d["cpu_baseline"]["value"])
d = json.load(open(f"{D}/bench_map_640.json")); print("map640", d["value"], d["ms_per_step"])
print("traffic", hbm)
