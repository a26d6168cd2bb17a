#!/usr/bin/env python3
"""bench.py — scan-pair registrations/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the registration hot path (align(): voxel filter of both clouds ->
search grid -> k-NN normals -> exactly `--iters` outer ICP iterations -> fitness pass -> gates)
over the rank's batch of `--pairs` independent synthetic scan pairs that are already resident in
HBM.  value = pairs processed by all ranks / max-over-ranks wall time.

Workload at N=1: BASELINE.json configs[2] ("batch of 256 independent 100k-pt scan pairs,
1xMI355X"), the largest single-GPU configuration and the per-GPU share of the 8-GPU sweep
(configs[3], 512 pairs per GPU) — the metric is a throughput.  configs[1] (one pair) is reported
beside it as `single_pair` latency.  Multi-GPU: pairs are sharded over ranks (no data-path
collective); each step ends with one RCCL all-gather of the 128-byte edge records.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 under torch.distributed.run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _gen_pair(args):
    n, idx = args
    import slam3d_amd.synthetic as syn
    return syn.make_pair(n, idx)


def kernel_source_hash():
    """sha256 over every source of libslam3d_hip.so (kernels AND the host file that holds the pass thresholds) - the
    hash the binary carries (s3d_source_hash): roofline.traffic / roofline.valu (PMC passes, profiles/nn_traffic.json)
    are attached only while the code they were measured on is the code that runs."""
    from slam3d_amd import api
    return api.source_hash()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _self_launch(n):
    """Run this script on n ranks (one per GPU) under torch.distributed.run, as the driver does for N > 1."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def _usable_cpus():
    """(CPUs this process can keep busy, the cgroup quota in CPUs or None): affinity mask, capped by cpu.max / cfs quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def _host_threads(usable_cpus, world):
    """host threads one rank may use (input generation, bulk hand-over): its share of the CPUs the JOB may use, 1 ... 16"""
    return max(1, min(16, int(usable_cpus) // max(int(world), 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=256, help="scan pairs per GPU and step")
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--algorithm", default="gicp", choices=["gicp", "icp"])
    ap.add_argument("--density", type=float, default=0.02)
    ap.add_argument("--cpu-pairs", type=int, default=10,
                    help="pairs timed one after another on the CPU oracle (rank 0, N=1): ~12 s at the default size")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="also time this many pairs on as many oracle threads at once (cpu_baseline_parallel); 0 = every "
                         "CPU this process may use (affinity mask and cgroup quota; SURVEY 8d 'all cores, pair-parallel'), 1 = off")
    ap.add_argument("--no-real", action="store_true", help="skip the real_scans block (the reference's own scans and defaults)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--nn-reps", type=int, default=20)
    ap.add_argument("--cells-per-point", type=int, default=0, help="search-grid budget (0 = library default)")
    ap.add_argument("--no-single", action="store_true",
                    help="skip the single-pair latency (BASELINE.json configs[1]) that is otherwise measured after the "
                         "timed region: with it off every s3d_nn_search_kernel<0> launch of the run belongs to the "
                         "batch workload (rocprofv3 average == roofline.avg_launch_ms, the profiles/ cross-check)")
    ap.add_argument("--no-search-frac", action="store_true",
                    help="skip the profile=2 run behind roofline.frac_search_passes (its counters slow the first passes: under "
                         "rocprofv3 --stats it would pollute the per-kernel averages the profiles/ cross-check reads)")
    ap.add_argument("--extras", action="store_true",
                    help="also time the first-iteration NN launch and the other algorithm of the path")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  Fresh child processes (torch.distributed.run),
        # started BEFORE this process has touched the GPU; this process only relays their output and exit code.
        sys.exit(_self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import slam3d_amd as s3d

    dist = None
    shared_devices = False
    # S3D_BENCH_FORCE_DIST=1: take the RCCL path with a single rank too (smoke test of the collective on a 1-GPU box)
    if world > 1 or os.environ.get("S3D_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        # The driver's runs use nccl (= RCCL over xGMI), one rank per GPU.  On a box with FEWER GPUs than ranks RCCL
        # cannot form the communicator (one rank per device): the ranks then share devices modulo the device count and
        # the all-gather runs over gloo on host tensors - a self-test of the multi-rank logic, flagged in the output
        # line ("collective", "ranks_share_devices"), never a scaling measurement.  S3D_BENCH_BACKEND forces either.
        ndev = max(torch.cuda.device_count(), 1)          # (counting devices does not initialise the GPU)
        backend = os.environ.get("S3D_BENCH_BACKEND") or ("nccl" if ndev >= world else "gloo")
        if backend == "gloo":
            shared_devices = ndev < world
            local_rank = local_rank % ndev
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    coll_backend = dist.get_backend() if dist is not None else None
    coll_dev = torch.device("cpu") if coll_backend == "gloo" else dev

    # ---- synthetic input (SURVEY.md §8d generator), distinct pairs per rank
    t0 = time.time()
    # threads, not processes: forking after a profiler / the HIP runtime is loaded hangs on this pool
    from multiprocessing.pool import ThreadPool
    jobs = [(args.points, rank * args.pairs + i) for i in range(args.pairs)]
    # host threads of this rank (input generation, the bulk hand-over): its share of the CPUs the job may use - affinity
    # mask and cgroup quota, e.g. 16 for 8 ranks -, so that N ranks do not oversubscribe the host into the timed region
    usable_cpus, cpu_quota = _usable_cpus()
    host_threads = _host_threads(usable_cpus, world)
    with ThreadPool(host_threads) as pool:
        pairs = pool.map(_gen_pair, jobs, chunksize=2)
    gen_s = time.time() - t0

    ctx = s3d.Context(local_rank)
    ctx.set_upload_threads(min(host_threads, 8))
    alg = s3d.ALG_GICP if args.algorithm == "gicp" else s3d.ALG_ICP
    params = s3d.default_params(registration_algorithm=alg, point_cloud_density=args.density,
                                maximum_iterations=args.iters, max_correspondence_distance=2.5,
                                correspondence_randomness=20)
    # cache_prepass=0: every step repeats the whole path (voxel filter, grid, k-NN pre-pass) like the reference
    opts = s3d.ExecOptions(force_iterations=1, check_interval=0, grid_cells_per_point=args.cells_per_point, profile=0,
                           cache_prepass=0)
    # host -> HBM hand-over of the batch's clouds (packed xyz from pageable memory).  NOT part of `value`: a mapper uploads
    # a scan once in its lifetime and registers it against many others from HBM (the C++ mirror caches the device copy
    # per measurement); reported beside it as upload_ms / value_incl_upload for the reader who re-uploads everything for
    # every batch.  Timed twice: one s3d_cloud_upload per cloud (hipMalloc + staged copy + float4 expansion + wait each:
    # upload_single_ms) and the bulk hand-over s3d_cloud_upload_many (one allocation, host threads filling pinned slots
    # while the expansion kernels read them over PCIe: upload_ms, the one value_incl_upload uses).
    warm = ctx.upload_many([p[0] for p in pairs[:16]])      # (first touches: allocator, pinned slots, threads; not timed)
    for c in warm + [ctx.upload(pairs[0][0])]:
        c.release()
    torch.cuda.synchronize()
    tu = time.perf_counter()
    one_by_one = [ctx.upload(p[0]) for p in pairs] + [ctx.upload(p[1]) for p in pairs]
    torch.cuda.synchronize()
    upload_single_ms = (time.perf_counter() - tu) * 1e3
    for c in one_by_one:
        c.release()
    tu = time.perf_counter()
    both = ctx.upload_many([p[0] for p in pairs] + [p[1] for p in pairs])
    torch.cuda.synchronize()
    upload_ms = (time.perf_counter() - tu) * 1e3
    src, tgt = both[:len(pairs)], both[len(pairs):]
    guesses = np.tile(np.eye(4), (args.pairs, 1, 1))

    def step():
        rec = ctx.align_batch(src, tgt, guesses, params, opts)
        if dist is not None:
            local = torch.from_numpy(rec).to(coll_dev)
            out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=local.dtype, device=coll_dev)
            dist.all_gather_into_tensor(out, local)  # RCCL over xGMI: 128 B per edge
            return out
        return rec

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        last = step()          # align_batch returns after its D2H of the records: the step is complete
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms_per_rank = [step_ms]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank's own step times (until round 6 only the maximum survived): a straggling rank shows
        mine = torch.tensor(step_ms, dtype=torch.float64, device=coll_dev)
        allr = torch.empty((world * len(step_ms),), dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(allr, mine)
        step_ms_per_rank = [[round(float(x), 3) for x in allr[r * len(step_ms):(r + 1) * len(step_ms)].tolist()] for r in range(world)]

    total_pairs = args.pairs * world * args.steps
    value = total_pairs / elapsed
    rec_local = last.cpu().numpy() if hasattr(last, "cpu") else last
    n_ok = int((rec_local[:, 15] == 0).sum())
    # (N > 1: `last` holds the gathered records of every rank - each rank's pairs come from their own generator seeds, so
    # the transforms are all different unless two ranks registered the same pairs)
    distinct = int(len({rec_local[i, :12].tobytes() for i in range(rec_local.shape[0])}))

    line = None
    if rank == 0:
        # ---- accuracy of this rank's batch vs the generator's ground truth (information only)
        errs = []
        for i in range(min(args.pairs, rec_local.shape[0])):
            T = s3d.api.record_transform(rec_local[i])
            d = np.linalg.inv(pairs[i][2]) @ T
            errs.append(np.linalg.norm(d[:3, 3]))
        # ---- per-stage profile of one step + NN kernel timing (HIP events on the context stream)
        popts = s3d.ExecOptions(force_iterations=1, check_interval=0, grid_cells_per_point=args.cells_per_point, profile=1)
        ctx.align_batch(src, tgt, guesses, params, popts)
        prof = ctx.last_profile()
        # roofline of the dominant kernel family (K5, the correspondence pass): algorithmic bytes per launch
        # (20*M + 12*N per NN pass, summed over the batch) / average launch duration over
        # the ICP loop of one step, HIP events on the stream the kernel runs on.
        n_launch = max(prof["nn_launches"], 1)
        nq, nt = prof["nn_queries"] / n_launch, prof["nn_targets"] / n_launch
        alg_bytes = 20.0 * nq + 12.0 * nt
        avg_ms = prof["nn_ms"] / n_launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC counters (separate rocprofv3 --pmc passes of this same command, summary
        # committed as profiles/nn_traffic.json; FETCH_SIZE doubled as the gfx950 guide prescribes).  Attached only for
        # the workload and the kernel sources it was measured on; otherwise null.
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "nn_traffic.json")
        if os.path.exists(tfile) and args.pairs == 256 and args.points == 100000 and args.iters == 20:
            tj = json.load(open(tfile))
            if tj.get("kernel_src_sha256") == kernel_source_hash():
                traffic = tj.get("hbm_bytes_per_launch")
        # "bound": the roofline the contract prices this path against (no dense contraction -> HBM).  What actually
        # limits the kernel is in "limiter" (PMC evidence in profiles/README.md): `achieved` is ALGORITHMIC bytes per
        # second, i.e. how far an exact grid search is from streaming its compulsory traffic.
        # The correspondence pass is a kernel FAMILY: pass 1 of a registration runs s3d_nn_first_kernel, passes 2 and 3
        # s3d_nn_scan27_kernel (+ s3d_nn_worklist_kernel for the queries it declines), and from pass 4 on (round 4, batches
        # of >= 65 536 records; smaller ones keep s3d_nn_search_kernel<0>) the record-wise kernels: s3d_nn_record_touch_kernel<false>
        # once, then s3d_nn_record_test_kernel + s3d_nn_record_touch_kernel<true> per pass, each followed by
        # s3d_nn_record_search_kernel.  avg_launch_ms is (the time of all of them in one step) / I, HIP events around every
        # pass (cross-check against rocprofv3: the TotalDurationNs of those kernel names / (I x steps)).  The ALGORITHMIC
        # bytes stay 20 M + 12 N per pass whatever is actually moved: a record-wise pass reads 32 bytes per 64 queries for
        # the records it proves unchanged, which is why its own `steady_frac` can exceed 1.
        roofline = {"kernel": "s3d_nn_first_kernel (pass 1) + s3d_nn_scan27_kernel<*> + s3d_nn_worklist_kernel (passes 2-3) "
                              "+ s3d_nn_record_{test,touch,search}_kernel (passes 4-%d)"
                              % n_launch,
                    "bound": "hbm", "achieved": round(achieved, 2),
                    "limiter": "VALU issue of divergent per-lane candidate walks in the first passes (75 / 78 / 52 % of the "
                               "issue slots with 44 / 64 / 57 % of the lanes active in passes 1 / 2 / 3: roofline.valu); in "
                               "the settled passes the latency of three dependent launches (record test, touch of the "
                               "~10 % failing records, search of ~15 queries per pair)",
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": n_launch,
                    "algorithmic_bytes_per_launch": int(alg_bytes),
                    "queries_per_launch": int(nq), "targets_per_launch": int(nt),
                    "gqueries_per_s": round(nq / (avg_ms * 1e-3) / 1e9, 3)}
        # the average above mixes two regimes (profiles/README.md): the badly aligned first passes and the
        # re-validated ones; reported separately for the reader, `frac` stays the all-launch figure
        # Two fractions that cannot be moved by adding forced iterations: the passes that SEARCH
        # (>= 1 % of their queries need a grid search: counted by a profile=2 run of the same batch, whose counters slow
        # the kernels - the times are the profile=1 run's) priced at the same algorithmic bytes, and the counter-measured
        # HBM traffic of the family over its time.
        try:
            if args.no_search_frac:
                raise RuntimeError("skipped (--no-search-frac)")
            popts2 = s3d.ExecOptions(force_iterations=1, check_interval=0, grid_cells_per_point=args.cells_per_point, profile=2)
            ctx.align_batch(src, tgt, guesses, params, popts2)
            prof2 = ctx.last_profile()
            searching = [i for i in range(min(n_launch, 64)) if prof2["nn_searched"][i] >= 0.01 * nq]
            t_search = sum(prof["nn_launch_ms"][i] for i in searching)
            if searching and t_search > 0:
                roofline["search_passes"] = [i + 1 for i in searching]
                roofline["search_passes_ms"] = round(t_search, 4)
                roofline["frac_search_passes"] = round(alg_bytes * len(searching) / (t_search * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        except Exception as e:
            roofline["frac_search_passes_error"] = str(e)[:120]
        roofline["hbm_frac"] = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None
        # The bound that actually binds: VALU issue slots and active lanes of the search passes, K4 and
        # K6 from the committed counter summary (profiles/nn_valu.json <- tools_dev/pmc_valu.sh + pmc_valu.py: separate
        # rocprofv3 --pmc passes of this workload), attached only while the library's source hash is the one measured.
        # issue_frac_live prices the counted instructions against THIS run's launch time (HIP events) at the clock the
        # counters saw: SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x clock x time).
        valu = None
        vfile = os.path.join(ROOT, "profiles", "nn_valu.json")
        if os.path.exists(vfile) and args.pairs == 256 and args.points == 100000 and args.iters == 20:
            vj = json.load(open(vfile))
            if vj.get("kernel_src_sha256") == kernel_source_hash():
                live_ms = {"nn_pass1": prof["nn_launch_ms"][0] if n_launch > 0 else None,
                           "nn_pass2": None, "nn_pass3": None, "nn_pass4": None,     # (these passes are two or three launches: counter time only)
                           "k4": None, "k6": None}
                valu = {"source": "profiles/nn_valu.json (round %s)" % vj.get("round"), "definition": vj.get("definition"), "kernels": {}}
                for tag, kv in vj.get("kernels", {}).items():
                    e = {k: kv[k] for k in ("kernel", "insts_valu_per_wave", "clock_ghz", "duration_ms_under_pmc", "valu_issue_frac",
                                            "active_lane_frac") if k in kv}
                    t = live_ms.get(tag)
                    if t:
                        e["issue_frac_live"] = round(kv["insts_valu"] * 4.0 / (1024.0 * kv["clock_ghz"] * 1e9 * t * 1e-3), 4)
                    valu["kernels"][tag] = e
        roofline["valu"] = valu
        lms = [x for x in prof["nn_launch_ms"] if x > 0]
        if len(lms) >= 12:
            steady = float(np.mean(lms[8:]))
            roofline["regimes"] = {"first_pass_ms": round(lms[0], 4), "passes_2_to_8_ms": [round(x, 4) for x in lms[1:8]],
                                   "steady_avg_launch_ms": round(steady, 4),
                                   "steady_achieved_gbs": round(alg_bytes / (steady * 1e-3) / 1e9, 1),
                                   "steady_frac": round(alg_bytes / (steady * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # ---- BASELINE.json configs[1]: one pair through the same entry point (latency of the reference's actual
        # call pattern, ScanSensor.cpp:113 - one createConstraint per new scan)
        single = None
        if not args.no_single:
            one_s, one_t = [src[0]], [tgt[0]]
            ctx.align_batch(one_s, one_t, guesses[:1], params, opts)
            reps = 10
            t1 = time.perf_counter()
            for _ in range(reps):
                ctx.align_batch(one_s, one_t, guesses[:1], params, opts)
            single_ms = (time.perf_counter() - t1) / reps * 1e3
            single = {"workload": "single %dk-pt synthetic scan pair, %d outer iterations, 1 GPU" %
                                  (args.points // 1000, args.iters),
                      "latency_ms": round(single_ms, 3), "registrations_per_s": round(1e3 / single_ms, 2)}
        # ---- secondary: the mapper's call pattern (ScanSensor.cpp:113, :179-201) - every NEW scan is linked to 8
        # scans that were registered before.  With the cross-call pre-pass cache (s3d_exec_options.cache_prepass, not
        # in the reference, never used by the timed region above) the 8 old scans pay the voxel filter / grid / k-NN
        # pre-pass once, when they were new.
        mapper = None
        if not args.no_single and args.pairs >= 16:
            nb = 8
            res = {}
            for label, cache in (("cache_off", 0), ("cache_on", 1)):
                ctx.cache_control(clear=True)
                o = s3d.ExecOptions(force_iterations=1, check_interval=0, grid_cells_per_point=args.cells_per_point,
                                    profile=0, cache_prepass=cache)
                # the synthetic sources all sample the same scene at the identity pose: any of them is a valid neighbour
                ctx.align_batch(src[:nb], [tgt[0]] * nb, guesses[:nb], params, o)          # warm-up (fills the cache)
                n_new = min(args.pairs - 1, 20)
                t1 = time.perf_counter()
                for j in range(1, n_new + 1):
                    ctx.align_batch(src[:nb], [tgt[j]] * nb, guesses[:nb], params, o)
                res[label] = (time.perf_counter() - t1) / n_new * 1e3
            ctx.cache_control(clear=True)
            mapper = {"workload": "each new %dk-pt scan registered against %d earlier scans (one batch call per new "
                                  "scan), %d outer iterations" % (args.points // 1000, nb, args.iters),
                      "ms_per_new_scan_cache_off": round(res["cache_off"], 3),
                      "ms_per_new_scan_cache_on": round(res["cache_on"], 3),
                      "links_per_s_cache_on": round(nb / res["cache_on"] * 1e3, 1)}
        # ---- secondary: two batches in flight.  A sweep (BASELINE configs[3]: 512 pairs per GPU) is a sequence of batches,
        # and the tail of one - the settled ICP passes, one-wave-per-pair controller launches - leaves most of the chip
        # idle while the next one's pre-pass could run.  The library serialises calls per context and runs calls on two
        # contexts side by side, so a caller pipelines with two contexts and two host threads (s3d_align_batch_multi with
        # the device listed twice does the same inside the C ABI); one 512-pair call gets the same rate.  The SAME
        # batches as the timed region, every step still paying the whole pre-pass; never part of `value`.
        inflight = None
        if not args.no_single and world == 1 and args.pairs >= 16:
            import threading
            ctx2 = s3d.Context(local_rank)
            both2 = ctx2.upload_many([p[0] for p in pairs] + [p[1] for p in pairs])
            lanes = [(ctx, src, tgt), (ctx2, both2[:len(pairs)], both2[len(pairs):])]
            rec2 = None
            for _ in range(2):
                rec2 = ctx2.align_batch(lanes[1][1], lanes[1][2], guesses, params, opts)
            per_lane = max(args.steps // 2, 3)

            lane_errors = []

            def lane_loop(k):
                c, a, b = lanes[k]
                try:
                    for _ in range(per_lane):
                        c.align_batch(a, b, guesses, params, opts)
                except Exception as e:      # (a thread's exception would otherwise vanish and leave a wrong figure)
                    lane_errors.append(str(e)[:200])

            torch.cuda.synchronize()
            t1 = time.perf_counter()
            th = [threading.Thread(target=lane_loop, args=(k,)) for k in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            fl_ms = (time.perf_counter() - t1) / (2 * per_lane) * 1e3
            inflight = {"workload": "the timed region's batches, two in flight (two contexts, two host threads, %d batches "
                                    "each)" % per_lane,
                        "ms_per_batch": round(fl_ms, 3), "registrations_per_s": round(args.pairs / fl_ms * 1e3, 2),
                        "records_equal_serial": bool(np.array_equal(rec2, rec_local))}
            if lane_errors:
                inflight = {"error": lane_errors[0]}
            for c in both2:
                c.release()
            ctx2.close()
        if args.extras:
            single = single or {}
            nn0 = ctx.profile_nn_kernel(src, tgt, guesses, params, reps=args.nn_reps)
            single["first_iteration_nn_launch_ms_full_batch"] = round(nn0["avg_ms"], 4)
            # the same batch through the other algorithm of the path (GICP <-> point-to-plane), for the reader
            other = s3d.ALG_ICP if alg == s3d.ALG_GICP else s3d.ALG_GICP
            op2 = s3d.default_params(registration_algorithm=other, point_cloud_density=args.density,
                                     maximum_iterations=args.iters, max_correspondence_distance=2.5,
                                     correspondence_randomness=20)
            ctx.align_batch(src, tgt, guesses, op2, opts)
            t2 = time.perf_counter()
            for _ in range(3):
                ctx.align_batch(src, tgt, guesses, op2, opts)
            ms2 = (time.perf_counter() - t2) / 3 * 1e3
            single["other_algorithm"] = {"algorithm": "icp (point-to-plane)" if other == s3d.ALG_ICP else "gicp",
                                         "ms_per_step": round(ms2, 3),
                                         "registrations_per_s": round(args.pairs / ms2 * 1e3, 2)}
        # ---- the reference's own data and defaults (north_star's parity clause is on test/cloud*.bin): 96 registrations
        # of consecutive fixture scans (tests/golden/cloud*.npz = the reference's test/cloud1..4.bin), default
        # RegistrationParameters (RegistrationParameters.hpp:36-97: GICP, 0.2 m voxels, 50 iterations, early exit ON,
        # all gates), one s3d_align_batch call; three of them against the oracle's align() on the same inputs
        real = None
        if not args.no_real and world == 1:
            try:
                G = os.path.join(ROOT, "tests", "golden")
                fc = [np.load(os.path.join(G, "cloud%d.npz" % i))["xyzi"].astype(np.float32) for i in range(1, 5)]
                fdev = [ctx.upload(c) for c in fc]
                rs, rt = [], []
                for _ in range(32):
                    for a, b in ((0, 1), (1, 2), (2, 3)):
                        rs.append(fdev[a]); rt.append(fdev[b])
                rp = s3d.default_params()
                ro = s3d.ExecOptions(force_iterations=0, check_interval=0, profile=0, cache_prepass=0)
                for _ in range(2):
                    rrec, rinfo = ctx.align_batch(rs, rt, None, rp, ro, want_infos=True)
                reps = 5
                t1 = time.perf_counter()
                for _ in range(reps):
                    rrec = ctx.align_batch(rs, rt, None, rp, ro)
                real_ms = (time.perf_counter() - t1) / reps * 1e3
                real = {"workload": "96 registrations of consecutive scans of the reference's test/cloud1..4.bin (124 k points "
                                    "each; the three distinct pairs 32 times in one s3d_align_batch call, so the pre-pass of "
                                    "the four clouds is shared), default RegistrationParameters: GICP, 0.2 m voxel filter, "
                                    "<= 50 outer iterations with PCL's early exit, all gates",
                        "ms_per_batch": round(real_ms, 3), "registrations_per_s": round(len(rs) / real_ms * 1e3, 1),
                        "status_ok": int((rrec[:, 15] == 0).sum()),
                        "median_outer_iterations": float(np.median([i["iterations"] for i in rinfo])),
                        "filtered_points": [int(rinfo[0]["n_source_filtered"]), int(rinfo[0]["n_target_filtered"])]}
                if not args.no_cpu:
                    import oracle
                    oracle.set_eval_precision(2)    # the smooth-objective variant, as in the parity tests
                    dts, drs, its = [], [], []
                    for k, (a, b) in enumerate(((0, 1), (1, 2), (2, 3))):
                        st_o, T_o, info_o = oracle.align(fc[a], fc[b], np.eye(4), oracle.default_params())
                        d = np.linalg.inv(T_o) @ s3d.api.record_transform(rrec[k])
                        w = np.array([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) / 2.0
                        dts.append(float(np.linalg.norm(d[:3, 3])))
                        drs.append(float(np.arctan2(np.linalg.norm(w), (np.trace(d[:3, :3]) - 1) / 2)))
                        its.append([int(rinfo[k]["iterations"]), int(info_o["iterations"]), int(rrec[k, 15]), int(st_o)])
                    oracle.set_eval_precision(0)
                    real["vs_oracle"] = {"pairs": 3, "max_dt_m": max(dts), "max_dr_rad": max(drs),
                                         "iterations_status_gpu_oracle": its, "oracle": "oracle/s3d_oracle.c align(), unpinned"}
                # -- the reference's actual call (missing 4): ONE pair per incoming scan (ScanSensor.cpp:113),
                # default RegistrationParameters, early exit on, one call each; and the loop-closure form through
                # createConstraint (PointCloudSensor.cpp:286-292: coarse + fine) for cloud1 -> cloud4 from a 2 m odometry
                # guess.  Median of 20 calls, pre-pass cache off (the reference's behaviour) and on.
                def _median_ms(fn, reps=20):
                    fn()
                    ts_ = []
                    for _ in range(reps):
                        t_ = time.perf_counter(); fn(); ts_.append((time.perf_counter() - t_) * 1e3)
                    return float(np.median(ts_))
                sp = {}
                ident = np.eye(4)
                odo = np.eye(4); odo[0, 3] = 2.0
                coarse = s3d.default_params(point_cloud_density=0.5, maximum_iterations=20)
                for label, cache in (("cache_off", 0), ("cache_on", 1)):
                    ctx.cache_control(clear=True)
                    o1 = s3d.ExecOptions(cache_prepass=cache)
                    st_a = ctx.align_clouds(fdev[0], fdev[1], ident, rp, o1)
                    sp["align_cloud1_cloud2_ms_" + label] = round(_median_ms(lambda: ctx.align_clouds(fdev[0], fdev[1], ident, rp, o1)), 4)
                    st_c = ctx.create_constraint_clouds(fdev[0], ident, fdev[3], ident, odo, True, rp, coarse, 1.0, o1)
                    sp["create_constraint_loop_cloud1_cloud4_ms_" + label] = round(_median_ms(
                        lambda: ctx.create_constraint_clouds(fdev[0], ident, fdev[3], ident, odo, True, rp, coarse, 1.0, o1)), 4)
                    sp["status_" + label] = [int(st_a[0]), int(st_c[0])]
                    sp["outer_iterations_" + label] = [int(st_a[2]["iterations"]), int(st_c[3]["iterations"])]
                ctx.cache_control(clear=True)
                sp["workload"] = ("ONE call per registration, the reference's defaults (GICP, 0.2 m voxels, <= 50 iterations, "
                                  "early exit): s3d_align_clouds(cloud1, cloud2) and s3d_create_constraint_clouds(cloud1, "
                                  "cloud4, odometry 2 m, loop = true: coarse 0.5 m + fine); median of 20 calls")
                real["single_pair"] = sp
                # -- the same 96 registrations with NO cloud shared between pairs: every copy moved by its own small rigid
                # motion (so that the pre-pass is paid 192 times, as for 192 different scans)
                rng = np.random.default_rng(5)
                ds, dt_ = [], []
                for k in range(32):
                    for a, b in ((0, 1), (1, 2), (2, 3)):
                        for which, lst in ((a, ds), (b, dt_)):
                            ang = rng.normal(0, 0.01, 3); tr = rng.normal(0, 0.05, 3)
                            cz, sz = np.cos(ang[2]), np.sin(ang[2])
                            R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array(
                                [[1, 0, ang[1]], [0, 1, -ang[0]], [-ang[1], ang[0], 1]])
                            c = fc[which].copy()
                            c[:, :3] = (c[:, :3].astype(np.float64) @ R.T + tr).astype(np.float32)
                            lst.append(c)
                ddev = ctx.upload_many([np.ascontiguousarray(c[:, :3]) for c in ds + dt_])
                dsrc, dtgt = ddev[:96], ddev[96:]
                for _ in range(2):
                    drec = ctx.align_batch(dsrc, dtgt, None, rp, ro)
                t1 = time.perf_counter()
                for _ in range(reps):
                    drec = ctx.align_batch(dsrc, dtgt, None, rp, ro)
                d_ms = (time.perf_counter() - t1) / reps * 1e3
                real["distinct"] = {"workload": "the same 96 registrations, every one on its own two clouds (each copy of a fixture "
                                                "scan moved by a small random rigid motion): 192 clouds through the pre-pass",
                                    "ms_per_batch": round(d_ms, 3), "registrations_per_s": round(96 / d_ms * 1e3, 1),
                                    "status_ok": int((drec[:, 15] == 0).sum()),
                                    "median_outer_iterations": float(np.median(drec[:, 13]))}
                for c in ddev:
                    c.release()
                for c in fdev:
                    c.release()
            except Exception as e:   # never let a secondary block take the contract line down
                real = {"error": str(e)[:200]}
        # ---- N > 1: the same sweep through the C ABI in ONE process (s3d_align_batch_multi: one rank = context + host
        # thread per device, RCCL all-gather of the records) - the layout a C++ ScanSensor::linkToNeighbors binds
        # (INTEGRATION.md).  Run by rank 0 after the timed region while the other ranks wait at the final barrier;
        # with fewer GPUs than ranks the device list repeats devices and the gather is device-to-device copies.
        sweep_abi = None
        if world > 1:
            try:
                ndev_all = max(torch.cuda.device_count(), 1)
                devs = [r % ndev_all for r in range(world)]
                sw = s3d.Sweep(devs)
                s_src = [sw.upload(p[0]) for p in pairs]
                s_tgt = [sw.upload(p[1]) for p in pairs]
                n_all = world * args.pairs                       # every rank's block: the pair list of rank 0 again
                g_all = np.tile(np.eye(4), (n_all, 1, 1))
                sw.align_batch(s_src * world, s_tgt * world, g_all, params, opts)          # uploads + warm-up
                reps = 2
                ts = time.perf_counter()
                for _ in range(reps):
                    rec_sw = sw.align_batch(s_src * world, s_tgt * world, g_all, params, opts)
                ms_sw = (time.perf_counter() - ts) / reps * 1e3
                sweep_abi = {"entry_point": "s3d_align_batch_multi (one process, one rank per device)",
                             "ranks": sw.ranks, "devices": devs, "collective": sw.collective, "pairs": n_all,
                             "ms_per_sweep": round(ms_sw, 3), "registrations_per_s": round(n_all / ms_sw * 1e3, 2),
                             "equals_single_context": bool(np.array_equal(rec_sw[:args.pairs], rec_local[:args.pairs])),
                             "note": "not the driver's metric: measured after the timed region, other ranks idle"}
                sw.close()
            except Exception as e:   # never let the secondary measurement take the contract line down
                sweep_abi = {"error": str(e)[:200]}
        # ---- CPU baseline: the oracle (a port of the reference path), one thread, same inputs/iterations
        cpu = None
        cpu_par = None
        cpu_pcl = None
        pcl_state = "not checked"
        if not args.no_cpu and world == 1:
            import oracle
            # The port allocates and frees its work arrays per call; with glibc's defaults every large block is an mmap /
            # munmap pair and its pages are faulted in again - from many threads at once that is one process-wide lock
            # (8 threads: 3.2x one thread; with the blocks kept on the heap 7.3x).  Keep them: M_MMAP_THRESHOLD,
            # M_TRIM_THRESHOLD, M_ARENA_MAX.  A fairer all-cores figure, and 4 % for the one-thread figure.
            try:
                import ctypes
                _libc = ctypes.CDLL("libc.so.6")
                _libc.mallopt(-3, 1 << 30); _libc.mallopt(-1, (1 << 31) - 1); _libc.mallopt(-8, max(os.cpu_count() or 1, 8))
            except OSError:
                pass
            op = oracle.default_params(registration_algorithm=alg, point_cloud_density=args.density,
                                       maximum_iterations=args.iters, max_correspondence_distance=2.5,
                                       correspondence_randomness=20)
            times = []
            for i in range(min(args.cpu_pairs, args.pairs)):
                tc = time.perf_counter()
                oracle.align(pairs[i][0], pairs[i][1], np.eye(4), op, force_iterations=True)
                times.append(time.perf_counter() - tc)
            # the same port on many cores at once (independent pairs, one thread each; ctypes releases the GIL)
            # the cores this process may actually use: its affinity mask and its cgroup's CPU quota (the GPU boxes of this
            # pool show 256 CPUs and grant 16: 256 threads there are throttled to ~8 cores' worth of progress)
            usable, quota = _usable_cpus()
            nthr = max(1, min(args.cpu_threads if args.cpu_threads > 0 else usable, usable, args.pairs))
            cpu_par = None
            if nthr > 1:
                def _one(i):
                    oracle.align(pairs[i][0], pairs[i][1], np.eye(4), op, force_iterations=True)
                npar = nthr * max(1, min(4, args.pairs // nthr))      # a few pairs per thread: >= 10 s of sample
                tp0 = time.perf_counter()
                with ThreadPool(nthr) as pool:
                    pool.map(_one, range(npar), chunksize=1)
                tpar = time.perf_counter() - tp0
                cpu_par = {"value": round(npar / tpar, 3), "unit": "registrations/s", "cores": nthr, "kind": "port",
                           "host_cpus": os.cpu_count(), "cpu_quota": quota,
                           "sample": "%d pairs of this workload on %d oracle threads at once (glibc kept from returning "
                                     "the work arrays to the kernel: mallopt), %.1f s" % (npar, nthr, tpar)}
            # the reference's own arithmetic, where this host has PCL (oracle/pcl, built by __graft_entry__.build())
            from oracle import pcl_pin
            pcl_state = pcl_pin.status()
            if pcl_state != "absent":
                pb = pcl_pin.bench(pairs[0][0], pairs[0][1], args.density, args.iters, reps=2)
                if pb:
                    cpu_pcl = {"value": round(pb["registrations_per_s"], 4), "unit": "registrations/s", "cores": 1,
                               "kind": "reference", "sample": "pair 0 of this workload through pcl::GeneralizedIterative"
                               "ClosestPoint (oracle/pcl/pcl_gicp bench), %d repetitions" % pb.get("reps", 2)}
            cpu = {"value": round(1.0 / float(np.median(times)), 4), "unit": "registrations/s", "cores": 1,
                   "kind": "port",
                   "sample": "%d of the %d pairs of this workload, oracle/s3d_oracle.c align() (kd-tree + "
                             "PCL-structured %s), median over the %d pairs, %.1f s total" %
                             (len(times), args.pairs, args.algorithm.upper(), len(times), sum(times)),
                   "host_cpus": os.cpu_count()}
        line = {
            "metric": "scan-pair registrations/sec (100k-pt clouds, 20 ICP iters)",
            "value": round(value, 2), "unit": "registrations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 points / f64 accumulators", "data": "synthetic",
            "config": {"workload": "batch of %d independent %dk-pt synthetic scan pairs per GPU, %d outer "
                                   "iterations (early exit disabled), %s, voxel leaf %.2f m, "
                                   "max_correspondence_distance 2.5 m, k=20, no cross-call caching (every step runs voxel filter, grid and "
                                   "k-NN pre-pass of all clouds again)" %
                                   (args.pairs, args.points // 1000, args.iters,
                                    "GICP (reference default)" if alg == s3d.ALG_GICP else "point-to-plane ICP",
                                    args.density),
                       "pairs_per_gpu": args.pairs, "points": args.points, "iterations": args.iters,
                       "algorithm": args.algorithm, "parallelism": "pair-sharded x%d" % world,
                       "collective": ("none" if dist is None else
                                      "all_gather of 128-B edge records (%s)" %
                                      ("RCCL over xGMI" if coll_backend == "nccl" else "gloo on host tensors")),
                       "ranks_share_devices": shared_devices},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_parallel": cpu_par,
            "cpu_baseline_pcl": cpu_pcl,
            "pcl": pcl_state,
            "real_scans": real,
            "upload_ms": round(upload_ms, 3),
            "upload_single_ms": round(upload_single_ms, 3),
            "value_incl_upload": round(args.pairs * world / (elapsed / args.steps + upload_ms * 1e-3), 2),
            "upload": "host -> HBM hand-over of the %d clouds of one batch (%.0f MB of packed xyz, pageable host memory), "
                      "measured once before the timed region: upload_ms = one s3d_cloud_upload_many (one allocation, host "
                      "threads -> pinned slots -> expansion kernels reading over PCIe), upload_single_ms = one "
                      "s3d_cloud_upload per cloud; `value` has the clouds resident in HBM, value_incl_upload re-uploads "
                      "all of them (upload_ms) for every batch" %
                      (2 * args.pairs, 2 * args.pairs * args.points * 12 / 1e6),
            "single_pair": single,
            "mapper_pattern": mapper,
            "two_in_flight": inflight,
            "sweep_abi": sweep_abi,
            "step_ms": step_ms,
            "step_ms_per_rank": step_ms_per_rank,
            "host_threads_per_rank": host_threads,
            "usable_cpus": usable_cpus,
            "cpu_quota": cpu_quota,
            "stage_ms": {k: round(v, 3) for k, v in prof.items() if k.endswith("_ms") and k != "nn_launch_ms"},
            "nn_launch_ms": prof["nn_launch_ms"],
            "distinct_pairs_gathered": distinct,
            "accuracy": {"status_ok": n_ok, "median_err_m": float(np.median(errs)), "max_err_m": float(np.max(errs))},
            "input_generation_s": round(gen_s, 1),
        }
        print(json.dumps(line), flush=True)
    for c in src + tgt:
        c.release()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
