"""dev tool (round 6): gpurun_out/pmc_valu/set{1,2}.csv (tools_dev/pmc_valu.sh) -> profiles/<round>/pmc_valu_summary.csv and
profiles/nn_valu.json, which bench.py attaches as roofline.valu while the library's source hash is the one it was measured
on.  usage: python tools_dev/pmc_valu.py r6"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r6"
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
P = os.path.join(ROOT, "gpurun_out", "pmc_valu")
D = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(D, exist_ok=True)
rows = []
for f in ("set1.csv", "set2.csv"):
    path = os.path.join(P, f)
    if not os.path.exists(path):
        sys.exit("missing %s (run tools_dev/pmc_valu.sh on the GPU box first)" % path)
    rows += list(csv.DictReader(open(path)))
if not rows:
    sys.exit("no counter rows")
with open(os.path.join(D, "pmc_valu_summary.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for r in rows:
        f.write('"%s",%s,%s,%s\n' % (r["kernel"], r["counter"], r["mean_per_launch"], r["launches"]))
K = {}
for r in rows:
    K.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_launch"])
# the names bench.py looks the kernels up by
TAGS = {"k4": "s3d_knn3_moments_kernel<20>", "nn_pass1": "s3d_nn_first_kernel", "nn_pass2": "s3d_nn_scan27_kernel<false, false>",
        "nn_pass3": "s3d_nn_scan27_kernel<true, true>", "nn_pass4": "s3d_nn_record_touch_kernel<false>", "k6": "s3d_gicp_accumulate_kernel"}
out = {}
for tag, name in TAGS.items():
    m = [k for k in K if k.startswith(name)]
    if not m:
        sys.exit("kernel %s not in the counter rows (renamed?)" % name)
    c = K[m[0]]
    need = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE", "DURATION_NS_UNDER_PMC")
    miss = [x for x in need if x not in c]
    if miss:
        sys.exit("%s: counters missing %s" % (name, miss))
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                     # (the counter is summed over the 8 XCDs: MI355X_MICROARCH.md)
    out[tag] = {"kernel": name, "insts_valu": c["SQ_INSTS_VALU"], "waves": c["SQ_WAVES"],
                "insts_valu_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1),
                "gpu_cycles": cycles, "duration_ms_under_pmc": round(c["DURATION_NS_UNDER_PMC"] / 1e6, 4),
                "clock_ghz": round(cycles / c["DURATION_NS_UNDER_PMC"], 3),
                "valu_issue_frac": round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles), 4),
                "active_lanes": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 2),
                "active_lane_frac": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0, 4)}
    print(tag, out[tag])
json.dump({"definition": "valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the VALU issue "
                         "slots of the launch a wave64 instruction of 4 cycles each would fill (above 1: instructions that issue in "
                         "fewer cycles, e.g. with EXEC = 0); active_lane_frac = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64",
           "command": "tools_dev/pmc_valu.sh: rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --no-cpu --no-single --no-real "
                      "--no-search-frac --steps 3 --warmup 1, two separate passes (SQ set, GRBM / SQ cycle set)",
           "workload": "256 pairs x 100k points, 20 iterations (bench default)", "kernels": out,
           "kernel_src_sha256": kernel_source_hash(), "round": ROUND},
          open(os.path.join(ROOT, "profiles", "nn_valu.json"), "w"), indent=1)
