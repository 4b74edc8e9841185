#!/usr/bin/env python3
"""Benchmark of the calibration hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: synthetic 20 poses x 300k points per cloud,
10^6-candidate combinatorial grid per GPU.  One "step" = one complete discrete
search over this rank's candidates: candidate 4-vector generation, the batched
residual kernel over every (candidate, point) pair, split reduction, arg-min,
near-minimum re-score with the reference arithmetic -- and, with N > 1, the
(cost, index) all-gather over RCCL that merges the shards.  Clouds are already
resident in HBM when the timed region starts.

Scaling is weak: every rank scores 10^6 candidates of an N x 10^6 grid (axis 0
is N times longer), so `value` = N * 10^6 * 6*10^6 residuals / step time.

Printed on rank 0 as ONE JSON line; extra objects:
  roofline      the residual kernel against the pipe that executes it: matrix FLOP
                per residual x residuals / HIP-event kernel time (events on the
                library's own stream) over the dense peak of that pipe (bf16: 2.5
                PFLOP/s; FP32-input MFMA: 157.3 TFLOP/s).  The 8-FLOP algorithmic
                figure against the FP32 peak is the named sub-object algorithmic_fp32.
  roofline_f32  the same for a few untimed steps of the FP32-input MFMA kernel.
  roofline_gd   the exact path's part kernel during the descent (FP64 VALU issue).
  cpu_baseline  the CPU oracle (a port of the reference's serial loop) timed on
                a bounded candidate sample of the same workload, rank 0, N = 1.
  north_star_1gpu (N = 1)  the 100 x 10^6-point configurations on this one GPU: upload, separable
                search of the 10^7 grid, 6-D descent to convergence, the dense kernel's rate on a
                stated candidate cut, the 10^8 grid's search, RANSAC (roofline_ransac) and the plane
                fit's moment pass (roofline_hbm), 64 lock-step descents.
  c3_strong (N > 1)  a step of the 100 x 10^6-point, 10^7-candidate grid cut over the N ranks.
The arg-min exchange of every step goes through the library's own RCCL collective
(ccal_comm_create + ccal_argmin_allgather, csrc/comm.hip), in a world of one rank too;
torch.distributed only launches the ranks, hands rank 0's communicator id to the others and
takes the maximum of the step times.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X FP32 vector == FP32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA peak
PEAK_FP64_LANE_OPS = 256 * 4 * 16 * 2.4e9   # FP64-rate VALU lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
FLOP_PER_RESIDUAL = 8             # 4 mul + 3 add (K=4 dot) + 1 add into the running sum
MATRIX_FLOP_PER_RESIDUAL = {      # what the matrix pipe executes per residual
    "bf16x3": 2 * 32,             # one 16x16x32 bf16 MFMA (16384 FLOP) per 256 residuals
    "f32": 2 * 4,                 # one 16x16x4 f32 MFMA (2048 FLOP) per 256 residuals
    "bf16x3wide": 2 * 32,         # two 32x32x16 bf16 MFMAs per 1024 residuals
    "bf16x3n8": 2 * 32,           # the 16x16x32 kernel with 8 candidate tiles per wave
}
MATRIX_PEAK_TFLOPS = {"bf16x3": PEAK_BF16_TFLOPS, "f32": PEAK_FP32_TFLOPS, "bf16x3wide": PEAK_BF16_TFLOPS,
                      "bf16x3n8": PEAK_BF16_TFLOPS}
KERNEL_NAME = {"bf16x3": "residual_bf16x3_kernel", "f32": "residual_mfma_kernel",
               "bf16x3wide": "residual_bf16x3_wide_kernel", "bf16x3n8": "residual_bf16x3_kernel"}
KERNEL_KIND = {"bf16x3": 0, "f32": 1, "bf16x3wide": 2, "bf16x3n8": 3}
# VALU instructions the part kernel of the exact path executes per (pose, point), from the
# SQ_INSTS_VALU counter of the committed PMC pass (profiles/): the unit of the GD leg's roofline
EXACT_VALU_PER_POINT_FILE = os.path.join(ROOT, "profiles", "r3_pmc_split_chain_kernel.json")
RESIDUAL_PMC_FILE = os.path.join(ROOT, "profiles", "r3_pmc_residual_kernel.json")
PEAK_HBM_GBS = 8000.0             # MI355X HBM3E


def quoted_counter(path, key, kernel_key):
    """(value, note): a derived counter of a committed PMC record, or (None, why) when the record is
    missing or was measured on other kernel sources than the ones in this tree."""
    from cameracalibration_amd.provenance import kernel_source_sha
    try:
        rec = json.load(open(path))
        v = rec["derived"][key]
    except Exception:
        return None, "no record"
    want = kernel_source_sha(kernel_key)
    if rec.get("kernel_source_sha") != want:
        return None, (f"stale: profiled on kernel sources {rec.get('kernel_source_sha')}, this tree has {want}")
    return v, "profiles/" + os.path.basename(path) + f" (kernel sources {want})"


def hbm_traffic_per_launch(args):
    """HBM bytes per launch of the residual kernel, QUOTED from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as the MI355X guide prescribes for gfx950, plus WRITE_SIZE): counters are
    collected in their own runs (tools/profile_r3.sh), never inside the benchmark, so the figure is
    given only for the workload it was measured on and only while the kernel's sources are the
    profiled ones; otherwise null, with the reason."""
    if (args.workload, args.clouds, args.points, args.axis, args.kernel) != ("c2", 20, 300_000, 10, "bf16x3"):
        return None, "not the profiled workload"
    return quoted_counter(RESIDUAL_PMC_FILE, "hbm_traffic_bytes_per_launch", "residual")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clouds", type=int, default=20)
    ap.add_argument("--points", type=int, default=300_000)
    ap.add_argument("--axis", type=int, default=10, help="grid steps per axis (10 -> 10^6 candidates)")
    ap.add_argument("--cpu-candidates", type=int, default=512,
                    help="candidates of the workload scored by the CPU oracle for cpu_baseline (0 = skip)")
    ap.add_argument("--kernel", choices=["bf16x3", "f32", "bf16x3wide", "bf16x3n8"], default="bf16x3",
                    help="fast-path kernel: exact bf16x3 operand split on the bf16 matrix pipe, or FP32-input MFMA")
    ap.add_argument("--gd-iterations", type=int, default=-1,
                    help="iteration cap of the untimed GD refinement (-1 = reference's 10000, 0 = skip)")
    ap.add_argument("--workload", choices=["c2", "c3"], default="c2",
                    help="c2: 20 x 300k points, 10^6 candidates per GPU (weak scaling, the metric's configuration); "
                         "c3: 100 x 1M points, ONE 10^7-candidate grid cut into N shards (strong scaling)")
    ap.add_argument("--search", choices=["dense", "separable"], default="dense",
                    help="how the timed step scores the grid (value and roofline are only reported for dense)")
    ap.add_argument("--f32-steps", type=int, default=3,
                    help="untimed steps of the FP32-input MFMA kernel for roofline_f32 (0 = skip)")
    ap.add_argument("--north-star", type=int, default=1,
                    help="N = 1: the 100 x 10^6-point configurations on this GPU after the benchmark (0 = skip)")
    ap.add_argument("--c3-steps", type=int, default=1,
                    help="N > 1: timed dense steps of the strong-scaled 10^7-candidate grid after the benchmark (0 = skip)")
    ap.add_argument("--c3-shard", default="",
                    help="R/W: rehearse on one GPU the c3_strong leg of rank R in a world of W ranks")
    a = ap.parse_args()
    if a.workload == "c3":
        a.clouds, a.points = 100, 1_000_000
    return a


def north_star(cc, opt, scene, grid3, upload_s, args):
    """The 100 poses x 10^6 points configurations on ONE GPU (the sequence of the reference's main,
    src/camera_calibration_optimization.cpp:576-578: discrete search, then the 6-D descent), with the
    search done by the separable pass; the dense kernel's sustained rate is measured on a stated cut of
    the same grid around the winner."""
    import torch
    o = {"workload": f"synthetic 100 poses x 1000000 pts/cloud ({scene.n_points} points), grid {grid3.size} candidates",
         "upload_s": upload_s, "upload_GBps": scene.n_points * 12 / upload_s / 1e9}
    opt.set_search_mode(1)
    opt.grid_search(scene.pose_init, grid3)                         # first call: buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, cost, pose, st = opt.grid_search(scene.pose_init, grid3)
    t1 = time.perf_counter()
    g = cc.GdOptions.default()
    if args.gd_iterations > 0:
        g.max_iterations = args.gd_iterations
    r = opt.gradient_descent(pose, g)
    t2 = time.perf_counter()
    o["separable_search_s"] = t1 - t0
    o["gradient_descent_s"] = t2 - t1
    o["time_to_converged_s"] = t2 - t0
    o["best_index"], o["best_cost"] = idx, cost
    o["gd"] = {"iterations": r.iterations, "converged": r.converged, "fval": r.fval, "evals": r.n_evals,
               "iteration_cap": g.max_iterations, "ms_per_iteration": (t2 - t1) * 1e3 / max(r.iterations, 1),
               "pose_error_vs_truth": (r.xval - scene.pose_true).tolist()}
    # the dense kernel on a cut of 2*10^5 consecutive candidates that holds the winner
    cut = 200_000
    b = max(0, min(grid3.size - cut, idx - cut // 2))
    opt.set_search_mode(0)
    i_d, c_d, _, st_d = opt.grid_search(scene.pose_init, grid3, b, b + cut)
    o["dense_cut"] = {"candidates": [b, b + cut], "residuals": st_d.residuals, "kernel_ms": st_d.kernel_ms,
                      "residuals_per_s": st_d.residuals / (st_d.kernel_ms * 1e-3),
                      "same_winner": bool((i_d, c_d) == (idx, cost)),
                      "whole_grid_would_take_s": grid3.size * scene.n_points / (st_d.residuals / (st_d.kernel_ms * 1e-3))}
    # BASELINE configs[3]'s grid (10^8 candidates) on this one GPU, separable pass
    counts4 = [20, 20, 25, 20, 20, 25]
    grid4 = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (c - 1) for c in counts4[:3]] + [10.0 / (c - 1) for c in counts4[3:]], counts4)
    opt.set_search_mode(1)
    t0 = time.perf_counter()
    i4, c4, _, _ = opt.grid_search(scene.pose_init, grid4)
    o["c4_grid_separable_search_s"] = time.perf_counter() - t0
    o["c4_grid"] = {"candidates": grid4.size, "best_index": i4, "best_cost": c4}
    opt.set_search_mode(0)
    # BASELINE configs[4]: RANSAC 4096 hypotheses x 10^8 points, the plane fit's moment pass, 64 descents
    opt.set_profiling(True)
    opt.ransac_plane(scene.pose_true, 4096, 0.01, 7, which=cc.CCAL_DOWNSAMPLED)       # first call: buffers
    opt.fit_plane(scene.pose_true, which=cc.CCAL_DOWNSAMPLED)
    opt.profile(reset=True)
    t0 = time.perf_counter()
    plane, inl, win = opt.ransac_plane(scene.pose_true, 4096, 0.01, 7, which=cc.CCAL_DOWNSAMPLED)
    t1 = time.perf_counter()
    for _ in range(5):
        fit = opt.fit_plane(scene.pose_true, which=cc.CCAL_DOWNSAMPLED)
    t2 = time.perf_counter()
    pr = opt.profile()
    opt.set_profiling(False)
    rate = pr.ransac_tests / (pr.ransac_score_ms * 1e-3)
    o["roofline_ransac"] = {
        "bound": "mfma", "kernel": "inlier_mfma_kernel", "kernel_avg_ms": pr.ransac_score_ms / max(pr.ransac_calls, 1),
        "tests_per_s": rate, "flop_per_test": MATRIX_FLOP_PER_RESIDUAL["bf16x3"],
        "achieved": rate * MATRIX_FLOP_PER_RESIDUAL["bf16x3"] / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": rate * MATRIX_FLOP_PER_RESIDUAL["bf16x3"] / 1e12 / PEAK_BF16_TFLOPS, "traffic": None,
        "pipe": "dense bf16 matrix pipe (operands as exact three-way bf16 splits, as the residual kernel); one "
                "(hypothesis, point) test = one K=4 contraction and one threshold compare; the compares and "
                "add-with-carries on the vector unit (8 per matrix instruction) set the pace, not the pipe",
        # the yardstick of the task: 8 algorithmic FLOP per test against the FP32 (vector = FP32-input matrix) peak
        "algorithmic_fp32": {"flop_per_test": FLOP_PER_RESIDUAL, "achieved": rate * FLOP_PER_RESIDUAL / 1e12,
                             "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "ratio": rate * FLOP_PER_RESIDUAL / 1e12 / PEAK_FP32_TFLOPS},
        "call_s": t1 - t0, "hypotheses": 4096, "inliers": inl, "winner": win,
        "exact_recount": {"hypotheses": int(pr.ransac_recounted), "kernel_ms": pr.ransac_recount_ms},
        "plane_error_vs_truth": (np.asarray(plane, dtype=np.float64) - scene.plane.astype(np.float64)).tolist()}
    bytes_per_s = pr.moments_points * 12 / (pr.moments_ms * 1e-3)
    o["roofline_hbm"] = {
        "bound": "hbm", "kernel": "moments_kernel", "kernel_avg_ms": pr.moments_ms / max(pr.moments_calls, 1),
        "bytes_per_point": 12, "achieved": bytes_per_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": bytes_per_s / 1e9 / PEAK_HBM_GBS, "traffic": None, "call_s": (t2 - t1) / 5,
        "plane_error_vs_truth": (np.asarray(fit, dtype=np.float64) - scene.plane.astype(np.float64)).tolist()}
    rng = np.random.default_rng(1)
    starts = scene.pose_true + rng.uniform(-1, 1, (64, 6)) * np.array([0.01] * 3 + [0.04] * 3)
    g2 = cc.GdOptions.default()
    g2.max_iterations = 10
    t0 = time.perf_counter()
    rs = opt.gradient_descent(starts, g2)
    o["multi_start_gd"] = {"starts": 64, "iteration_cap": 10, "seconds": time.perf_counter() - t0,
                           "best_fval": min(x.fval for x in rs), "evals": int(sum(x.n_evals for x in rs))}
    return o


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist

    import cameracalibration_amd as cc
    from cameracalibration_amd.synthetic import make_scene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # torch.distributed is the control plane only (rendezvous, barriers, the maximum of the step times)
    # and runs over gloo on the host; the data path's one collective is the library's own RCCL
    # all-gather (below).  CCAL_BENCH_BACKEND=gloo rehearses the N>1 code path on a box with fewer
    # GPUs than ranks: the ranks then share devices, which RCCL refuses, so the exchange falls back to
    # torch's all_gather and the line says so.
    backend = os.environ.get("CCAL_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    xdev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on the process's standard output; this program's standard output
        # carries one JSON line and nothing else, so the announcements go to standard error
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --- inputs: identical scene on every rank (seeded), resident before timing ----
    scene = make_scene(args.clouds, args.points)
    opt = cc.Optimizer(dev_index)
    opt.upload_clouds(cc.CCAL_DOWNSAMPLED, scene.clouds, scene.base_T_flange)
    opt.set_plane(scene.plane)
    opt.set_fast_kernel(KERNEL_KIND[args.kernel])

    if args.workload == "c3":
        # one grid for the whole job, (20,20,25 | 10,10,10) = 10^7 candidates, contiguous shards
        counts = [20, 20, 25, 10, 10, 10]
        grid = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (c - 1) for c in counts[:3]] + [10.0 / (c - 1) for c in counts[3:]],
                            counts)
        total = grid.size
        begin, end = (total * rank) // world, (total * (rank + 1)) // world
        scaling = "strong"
    else:
        n = args.axis
        counts = [n * world, n, n, n, n, n]
        steps_mm = 40.0 / (n - 1)
        steps_deg = 10.0 / (n - 1)
        grid = cc.Grid.make([-20] * 3 + [-5] * 3, [steps_mm] * 3 + [steps_deg] * 3, counts)
        total = grid.size
        begin, end = rank * (total // world), (rank + 1) * (total // world)      # total = world * per_rank exactly
        scaling = "weak"
    per_rank = end - begin
    opt.set_search_mode(1 if args.search == "separable" else 0)

    from cameracalibration_amd.distributed import merge_argmin

    # The one exchange of the path is the library's own: ccal_comm_create + ccal_argmin_allgather
    # (ncclAllGather of 16-byte records on the context's stream).  torch.distributed only carries rank
    # 0's communicator id to the other ranks.  The gloo rehearsal (ranks sharing one GPU, which RCCL
    # refuses) falls back to torch's all_gather and says so in the line.
    exchange = {"through": None, "rccl_ranks": None, "us": []}

    def join_library_comm(o):
        """Collective.  Returns RCCL's own count of the communicator's ranks, or None when any rank could
        not join (every rank then uses the fallback; the reason goes into the line)."""
        err = ""
        try:
            uid = cc.comm_unique_id() if rank == 0 else bytes(128)
        except Exception as e:                      # librccl missing on rank 0: the others must not wait in RCCL
            uid, err = bytes(128), f"comm_unique_id: {e}"
        if world > 1:
            t = torch.tensor(list(uid) + [1 if err else 0], dtype=torch.uint8, device=xdev)
            dist.broadcast(t, 0)
            v = t.cpu().tolist()
            uid = bytes(v[:128])
            if v[128] and not err:
                err = "rank 0 could not make a communicator id"
        n = None
        if not err:
            try:
                o.comm_create(world, rank, uid)
                n, r = o.comm_info()
                if (n, r) != (world, rank):
                    err = f"RCCL reports rank {r} of {n}, the launcher {rank} of {world}"
            except Exception as e:
                err = f"comm_create: {e}"
        if world > 1:
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=xdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and not err:
                err = "another rank could not join the communicator"
        if err:
            o.comm_destroy()
            exchange["fallback_reason"] = err
            return None
        return n

    exchange["fallback_reason"] = None
    if backend == "nccl":
        exchange["rccl_ranks"] = join_library_comm(opt)
    if exchange["rccl_ranks"]:
        exchange["through"] = "libccal: ccal_argmin_allgather (ncclAllGather of 16-byte records, csrc/comm.hip)"
    elif backend == "nccl":
        exchange["through"] = ("torch.distributed all_gather over gloo -- FALLBACK, the library's RCCL communicator "
                               "could not be made: " + str(exchange["fallback_reason"]))
    else:
        exchange["through"] = "torch.distributed all_gather over gloo (rehearsal: NOT the product's collective)"
    exchange["control_plane"] = "torch.distributed over gloo (id broadcast, barriers, max of step times)" if world > 1 else None
    use_library = [bool(exchange["rccl_ranks"])]

    def merge(o, cost, idx):
        t0 = time.perf_counter()
        if use_library[0]:
            cost, idx, _ = o.argmin_allgather(cost, idx)
        elif world > 1:
            cost, idx, _ = merge_argmin(cost, idx, device=xdev)
        exchange["us"].append((time.perf_counter() - t0) * 1e6)
        return cost, idx

    def step():
        idx, cost, pose, st = opt.grid_search(scene.pose_init, grid, begin, end)
        cost, idx = merge(opt, cost, idx)
        pose = cc.grid_pose(grid, scene.pose_init, idx) if idx >= 0 else pose
        return idx, cost, pose, st

    for _ in range(args.warmup):
        step()
    exchange["us"].clear()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    residuals_rank = 0.0
    for _ in range(args.steps):
        idx, cost, pose, st = step()
        kernel_ms += st.kernel_ms
        launches += st.kernel_launches
        residuals_rank += st.residuals
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    residuals_step = float(total) * scene.n_points if args.workload == "c3" else float(per_rank) * scene.n_points * world
    value = residuals_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    res_per_s = residuals_rank / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0      # this rank's kernel rate
    traffic, traffic_source = hbm_traffic_per_launch(args)

    def roofline_of(kernel, rate, avg_ms):
        mf, peak = MATRIX_FLOP_PER_RESIDUAL[kernel], MATRIX_PEAK_TFLOPS[kernel]
        ach = rate * mf / 1e12
        return {
            "bound": "mfma",
            "achieved": ach,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": ach / peak,
            "kernel": KERNEL_NAME[kernel],
            "kernel_avg_ms": avg_ms,
            "flop_per_residual": mf,
            "residuals_per_s": rate,
            "pipe": "FP32-input matrix instruction (16x16x4), same peak as the FP32 vector unit" if kernel == "f32"
                    else "dense bf16 matrix pipe; both FP32 operands enter as exact three-way bf16 splits",
            # the yardstick the task names: 8 algorithmic FLOP per residual against the FP32 peak.  Not a
            # roofline fraction for the bf16 kernels (they do not run on that pipe), hence a field of its own.
            "algorithmic_fp32": {"flop_per_residual": FLOP_PER_RESIDUAL, "achieved": rate * FLOP_PER_RESIDUAL / 1e12,
                                 "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                 "ratio": rate * FLOP_PER_RESIDUAL / 1e12 / PEAK_FP32_TFLOPS},
        }

    out = {
        "metric": "point-to-plane residuals/sec (whole node) + time-to-converged flange_T_camera",
        "value": value,
        "unit": "residuals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        # operands are FP32; the default kernel feeds them to the bf16 matrix pipe as an exact
        # three-way bf16 split and accumulates in FP32 (same 1e-5 tolerance as the FP32 kernel)
        "dtype": "f32" if args.kernel == "f32" else "f32 (bf16x3 split operands, f32 accumulate)",
        "data": "synthetic",
        "config": {
            "workload": (f"synthetic {args.clouds} poses x {args.points} pts/cloud, one {total}-candidate grid in "
                         f"{world} contiguous shards (discrete search step)") if args.workload == "c3" else
                        (f"synthetic {args.clouds} poses x {args.points} pts/cloud, "
                         f"{per_rank}-candidate combinatorial grid per GPU (discrete search step)"),
            "clouds": args.clouds,
            "points_per_cloud": args.points,
            "candidates_per_gpu": per_rank,
            "candidates_total": total,
            "search": args.search,
            "parallelism": f"candidate shards x{world}, RCCL all-gather argmin" if world > 1 else "single GPU",
        },
        # the arg-min exchange inside every timed step (a world of one rank at N = 1)
        "exchange": {"through": exchange["through"], "rccl_ranks": exchange["rccl_ranks"],
                     "us_per_step_rank0": float(np.mean(exchange["us"])) if exchange["us"] else None,
                     "us_max_rank0": float(np.max(exchange["us"])) if exchange["us"] else None,
                     "control_plane": exchange["control_plane"]},
        "value_note": None if args.search == "dense" else
                      "separable search: candidates x points per second as if every residual were evaluated; none is "
                      "(one histogram pass per rotation answers all its translations) -- not the metric, no roofline",
        "best_index": idx,
        "best_cost": cost,
        "n_rescored": int(st.n_rescored),
        "rescore_band": {"abs": st.band_abs, "rel": st.band_rel},
    }
    if args.search == "dense":
        out["roofline"] = roofline_of(args.kernel, res_per_s, kernel_ms / max(launches, 1))
        out["roofline"]["traffic"] = traffic
        out["roofline"]["traffic_source"] = traffic_source
    else:
        out["roofline"] = None       # the separable pass evaluates no residual one by one: nothing to price

    if rank == 0 and world == 1 and args.workload == "c2":
        # ---- untimed: the FP32-input MFMA kernel on the same workload (its own roofline line) ----
        if args.f32_steps > 0 and args.kernel != "f32" and args.search == "dense":
            opt.set_fast_kernel(KERNEL_KIND["f32"])
            opt.grid_search(scene.pose_init, grid, begin, end)
            ms = res = 0.0
            n_l = 0
            for _ in range(args.f32_steps):
                i32, c32, _, st32 = opt.grid_search(scene.pose_init, grid, begin, end)
                ms += st32.kernel_ms
                res += st32.residuals
                n_l += st32.kernel_launches
            opt.set_fast_kernel(KERNEL_KIND[args.kernel])
            out["roofline_f32"] = roofline_of("f32", res / (ms * 1e-3), ms / max(n_l, 1))
            out["roofline_f32"]["steps"] = args.f32_steps
            out["roofline_f32"]["same_result"] = bool(i32 == idx and c32 == cost)
        opt.set_search_mode(0)
        # ---- untimed: the same search by the separable pass (one histogram pass per rotation
        # answers all its translations; no residual is evaluated one by one, so this is NOT
        # part of `value` or of the roofline) ----
        opt.set_search_mode(1)
        opt.grid_search(scene.pose_init, grid, begin, end)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            idx_s, cost_s, _, st_s = opt.grid_search(scene.pose_init, grid, begin, end)
        sep_ms = (time.perf_counter() - t1) / reps * 1e3
        opt.set_search_mode(0)
        out["separable_search"] = {
            "ms_per_search": sep_ms,
            "kernel_ms": st_s.kernel_ms,
            "same_result_as_dense": bool(idx_s == idx and cost_s == cost),
            "equivalent_residuals_per_s": residuals_step / (sep_ms * 1e-3),
            "note": "same arg-min and cost as the dense search; work is O(rotations x points), not one "
                    "residual per (candidate, point): reported beside the metric, never as the metric",
        }
        # ---- untimed: GD refinement from the discrete optimum (time-to-converged) ----
        if args.gd_iterations != 0:
            o = cc.GdOptions.default()
            if args.gd_iterations > 0:
                o.max_iterations = args.gd_iterations
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            idx2, cost2, pose2, _ = opt.grid_search(scene.pose_init, grid, begin, end)
            t2 = time.perf_counter()
            r = opt.gradient_descent(pose2, o)
            t3 = time.perf_counter()
            # the same descent once more with device timers on: roofline of the exact path's part kernel
            opt.set_profiling(True)
            opt.profile(reset=True)
            t_p0 = time.perf_counter()
            r_p = opt.gradient_descent(pose2, o)
            t_p1 = time.perf_counter()
            pr = opt.profile()
            opt.set_profiling(False)
            vpp, vpp_source = quoted_counter(EXACT_VALU_PER_POINT_FILE, "valu_instructions_per_pose_point", "split_chain")
            if pr.split_chain_ms > 0:
                rate = pr.split_pose_points / (pr.split_chain_ms * 1e-3)
                out["roofline_gd"] = {
                    "bound": "fp64_valu_issue",
                    "kernel": "split_chain_kernel",
                    "kernel_avg_ms": pr.split_chain_ms / max(pr.split_requests, 1),
                    "requests": int(pr.split_requests),
                    "pose_points_per_s": rate,
                    "us_per_pose": pr.split_chain_ms * 1e3 / (pr.split_pose_points / scene.n_points),
                    "valu_instructions_per_pose_point": vpp,
                    "valu_source": vpp_source,
                    "achieved": rate * vpp / 1e12 if vpp else None,
                    "peak": PEAK_FP64_LANE_OPS / 1e12,
                    "unit": "T lane-instructions/s (FP64-rate VALU: 16 lanes per SIMD and cycle)",
                    "frac": rate * vpp / PEAK_FP64_LANE_OPS if vpp else None,
                    "traffic": None,
                    "other_kernels_ms": {"split_finish_kernel": pr.split_finish_ms, "exact_chain_kernel": pr.whole_chain_ms,
                                         "screen_prep+cost+reduce": pr.screen_ms},
                    # where an iteration's wall time goes (this profiled descent): device time of the
                    # kernels, host time before and after the stream waits, and the waits themselves
                    # ("wall" is this profiled run's: with the timers on, the waits go through stream events;
                    # "wall_unprofiled" is the descent timed for time_to_converged, which watches the result words)
                    "us_per_iteration": {
                        "wall": (t_p1 - t_p0) * 1e6 / max(r_p.iterations, 1),
                        "wall_unprofiled": (t3 - t2) * 1e6 / max(r.iterations, 1),
                        "split_chain_kernel": pr.split_chain_ms * 1e3 / max(r_p.iterations, 1),
                        "split_finish_kernel": pr.split_finish_ms * 1e3 / max(r_p.iterations, 1),
                        "screening_kernels": pr.screen_ms * 1e3 / max(r_p.iterations, 1),
                        "host_outside_waits": (pr.exact_host_ms + pr.screen_host_ms) * 1e3 / max(r_p.iterations, 1),
                        "waits": (pr.exact_wait_ms + pr.screen_wait_ms) * 1e3 / max(r_p.iterations, 1)},
                    "same_iterates": bool(np.array_equal(r_p.xval, r.xval) and r_p.iterations == r.iterations),
                }
            out["time_to_converged_s"] = t3 - t1
            out["time_to_converged_phases_s"] = {"dense_search": t2 - t1, "gradient_descent": t3 - t2,
                                                 "separable_search_instead": sep_ms * 1e-3}
            # the shipped CLI's default: the separable pass for the search (same winner, checked above)
            out["time_to_converged_separable_s"] = sep_ms * 1e-3 + (t3 - t2)
            out["gd"] = {"iterations": r.iterations, "converged": r.converged, "fval": r.fval,
                         "evals": r.n_evals, "iteration_cap": o.max_iterations,
                         "pose_error_vs_truth": (r.xval - scene.pose_true).tolist()}
        # ---- CPU baseline: the oracle's serial loop on a bounded sample ---------------
        if args.cpu_candidates > 0:
            from oracle import oracle as O
            osc = O.OracleScene(scene.clouds, scene.base_T_flange)
            og = O.make_grid([grid.min[k] for k in range(6)], [grid.step[k] for k in range(6)], counts)
            nc = args.cpu_candidates
            t1 = time.perf_counter()
            bi, bc, _, costs = osc.grid_search(scene.plane, scene.pose_init, og, 0, nc, want_costs=True)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": nc * scene.n_points / dt,
                "unit": "residuals/s",
                "cores": 1,
                "kind": "port",
                "sample": f"first {nc} grid candidates x {scene.n_points} points, serial oracle (oracle/ccal_oracle.c)",
                "seconds": dt,
            }
            nthreads = O.omp_threads()
            nc2 = nc * min(nthreads, 16)
            t1 = time.perf_counter()
            osc.grid_search(scene.plane, scene.pose_init, og, 0, nc2, omp=True)
            dt2 = time.perf_counter() - t1
            out["cpu_baseline_all_cores"] = {
                "value": nc2 * scene.n_points / dt2, "unit": "residuals/s", "cores": nthreads,
                "kind": "port", "sample": f"first {nc2} grid candidates, OpenMP over candidates", "seconds": dt2,
            }
            # the sample doubles as a parity spot check of the fast path
            fast = opt.get_error_fast(np.stack([cc.grid_pose(grid, scene.pose_init, i) for i in range(nc)]))
            out["fast_vs_cpu_max_rel_err"] = float(np.max(np.abs(fast - costs) / costs))

    # ---- beyond the metric's configuration: the 100 x 10^6-point cases ---------------------------
    shard = None
    if args.c3_shard:
        shard = tuple(int(v) for v in args.c3_shard.split("/"))
    if args.workload == "c2" and ((world == 1 and rank == 0 and (args.north_star or shard)) or (world > 1 and args.c3_steps > 0)):
        opt.comm_destroy()
        opt.close()
        del scene
        big = make_scene(100, 1_000_000)
        opt3 = cc.Optimizer(dev_index)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt3.upload_clouds(cc.CCAL_DOWNSAMPLED, big.clouds, big.base_T_flange)
        upload_s = time.perf_counter() - t1
        opt3.set_plane(big.plane)
        opt3.set_fast_kernel(KERNEL_KIND[args.kernel])
        counts3 = [20, 20, 25, 10, 10, 10]
        grid3 = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (c - 1) for c in counts3[:3]] + [10.0 / (c - 1) for c in counts3[3:]],
                             counts3)
        if world > 1 or shard:
            # BASELINE configs[2]: ONE 10^7-candidate grid cut into contiguous shards, strong scaling
            w3, r3 = (world, rank) if world > 1 else (shard[1], shard[0])
            b3, e3 = (grid3.size * r3) // w3, (grid3.size * (r3 + 1)) // w3
            if use_library[0]:
                use_library[0] = bool(join_library_comm(opt3))

            def step3(mode):
                opt3.set_search_mode(mode)
                i3, c3, _, st3 = opt3.grid_search(big.pose_init, grid3, b3, e3)
                c3, i3 = merge(opt3, c3, i3)
                return i3, c3, st3

            leg = {}
            for mode, name, n_steps, n_warm in ((1, "separable", 3, 1), (0, "dense", args.c3_steps, 0)):
                if n_steps <= 0:
                    continue
                for _ in range(n_warm):
                    step3(mode)
                barrier()
                t1 = time.perf_counter()
                k_ms = res3 = 0.0
                for _ in range(n_steps):
                    i3, c3, st3 = step3(mode)
                    k_ms += st3.kernel_ms
                    res3 += st3.residuals
                barrier()
                dt = time.perf_counter() - t1
                if world > 1:
                    t = torch.tensor([dt], dtype=torch.float64, device=xdev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                leg[name] = {"steps": n_steps, "ms_per_step": dt / n_steps * 1e3, "best_index": i3, "best_cost": c3,
                             "residuals_per_s_whole_job": float(grid3.size) * big.n_points * n_steps / dt}
                if mode == 0:
                    leg[name]["kernel_residuals_per_s_this_rank"] = res3 / (k_ms * 1e-3) if k_ms > 0 else None
            leg["same_winner"] = bool("dense" not in leg or "separable" not in leg or
                                      (leg["dense"]["best_index"], leg["dense"]["best_cost"]) ==
                                      (leg["separable"]["best_index"], leg["separable"]["best_cost"]))
            out["c3_strong"] = {
                "workload": f"synthetic 100 poses x 1000000 pts/cloud, one {grid3.size}-candidate grid in {w3} contiguous shards",
                "scaling": "strong", "ranks": w3, "rehearsed_rank": None if world > 1 else r3,
                "candidates_this_rank": e3 - b3, "upload_s": upload_s, **leg,
                "note": "dense: every (candidate, point) residual evaluated; separable: one histogram pass per "
                        "rotation (same winner), reported as if every residual were evaluated"}
            opt3.comm_destroy()
        else:
            out["north_star_1gpu"] = north_star(cc, opt3, big, grid3, upload_s, args)
        opt3.close()

    if rank == 0:
        print(json.dumps(out), flush=True)
    try:
        opt.close()
    except Exception:
        pass
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
