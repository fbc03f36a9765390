#!/usr/bin/env python3
"""bench.py -- throughput of the EKF/EKS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (forward EKF kernel, pinv grid, backward EKS kernel, scenario-scoring tail) over
BASELINE.json's headline sweep: SIAlphaModelEKFOptControlled over 300 regions x 250 NPI-cost weights x (400 observed +
120 horizon) days = 75 000 chains x 520 days = 39.0 M region-day EKF steps, all 11 reference outputs written (1376
algorithmic bytes per region-day step).  Inputs are resident in HBM before the timed region.

N > 1 (one process per GPU, torch.distributed over RCCL): the SAME fixed sweep is sharded by contiguous chain blocks
(batch.shard_chains, 75 000 / N chains per rank) -- strong scaling, which is what BASELINE's "1 -> 8 GPU scaling" of
the 300 x 400 x 250 sweep means.  Chains are independent, so there is no collective on the data path; at the end of
every pass each rank's per-chain Pareto coordinates (J0, J1) -- 16 B per chain -- are gathered to rank 0, which filters
the front per region (the path's only collective).  `--scaling weak` runs the round-1 variant instead (every rank a
full 75 000-chain shard of a 300 N-region sweep).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_STEP_FULL = {3: 632, 6: 1376}      # inputs 112 B + all reference outputs
# split of the algorithmic bytes by kernel (DESIGN.md "Algorithmic bytes"): forward kernel reads
# x,u,R (112 B) and writes S-,S+,P-,P+,K,innov,rho,u_opt; smoother writes S_s,P_s,u_opt_smooth
BYTES_FWD = {3: 112 + 8 * (3 + 3 + 9 + 9 + 3 + 1 + 1 + 12), 6: 112 + 8 * (6 + 6 + 36 + 36 + 6 + 1 + 1 + 12)}
BYTES_BWD = {3: 8 * (3 + 9 + 12), 6: 8 * (6 + 36 + 12)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=300)
    ap.add_argument("--eps", type=int, default=250)
    ap.add_argument("--t-hist", type=int, default=400)
    ap.add_argument("--horizon", type=int, default=120)
    ap.add_argument("--workload", default="cfg4", choices=["cfg4", "cfg4-live", "cfg3", "cfg5", "newcase"],
                    help="cfg4 = the series SURVEY.md 8(d) specifies (one wave, extinct long before day 400; the headline); "
                         "cfg4-live = the same sweep on a living multi-wave epidemic (synth.make_cfg4(live=True))")
    ap.add_argument("--outputs", default="all", choices=["all", "reduced"],
                    help="reduced = u_opt_smooth + S_SMOOTH only (what TrainPredictPrescribeNPI.m:460-493 consumes)")
    ap.add_argument("--time-pipe", type=int, default=0, choices=[-1, 0, 1],
                    help="epi_batch_desc.time_pipe: 0 = the library decides whether the forward kernel runs in time segments "
                         "with the pinv grid of each beside the next (it does for batches that leave SIMDs idle), 1 = on, -1 = off")
    ap.add_argument("--lane-block", type=int, default=-1,
                    help="output layout (epi_batch_desc.lane_block): -1 = chain-blocked with one block per wavefront of the "
                         "launch (epi_ekf_preferred_lane_block; default), n > 0 = blocks of n chains, 0 = classic [T][rows][B]")
    ap.add_argument("--no-score", action="store_true",
                    help="skip the scenario-scoring tail (SIalpha_Controlled + NPICost on the horizon) after each pass")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the fixed sweep sharded over the ranks (default); weak = every rank a full sweep "
                         "of its own regions")
    ap.add_argument("--shape", default="auto", choices=["auto", "lane", "quad", "wave", "hex"],
                    help="lane mapping of the 6-state kernels (epi_batch_desc.shape): auto = by batch size")
    ap.add_argument("--storage", default="f64", choices=["f64", "f32"],
                    help="f32 = BASELINE config 5's fp32: outputs STORED as float32, arithmetic and the smoother's inputs fp64 "
                         "(SURVEY.md 0: fp32 covariance arithmetic is not viable at cond(P) up to 1e8)")
    ap.add_argument("--placement-tries", type=int, default=5,
                    help="one-time set-up before the first pass (EkfRunner.tune_placement): time a staged pass on this many "
                         "allocations of the outputs + workspace and keep the fastest (where the allocator puts the ~14 concurrently "
                         "streamed arrays changes a pass by up to 15 %, DESIGN.md 5); 1 = take what the allocator gives")
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="set-up, before the W warm-up passes: keep running passes until this much wall time has gone by, so that the "
                         "device is at its steady clocks when the timed region starts -- after an idle second the first ~50-150 ms of "
                         "work run 3-15 %% slower (profiles/r05/clock_ramp.txt), which a 2.5 ms pass would carry into ten timed passes "
                         "and a 15.6 ms pass would not; 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-chains", type=int, default=0, help="0 = sized for ~15 s")
    return ap.parse_args()


def make_workload(args, rank):
    from epidemicmodeling_amd import synth
    if args.workload in ("cfg4", "cfg4-live"):
        live = args.workload == "cfg4-live"
        w = synth.make_cfg4(args.regions, args.eps, args.t_hist, args.horizon,
                            region_offset=rank * args.regions if args.scaling == "weak" else 0, live=live)
        name = (f"{args.workload}: SIAlphaModelEKFOptControlled sweep, {args.regions} regions x {args.eps} cost weights x "
                f"({args.t_hist}+{args.horizon}) days" + (", living multi-wave epidemic (reactive NPI history)" if live else ""))
    elif args.workload == "newcase":
        w = synth.make_newcase_sweep(args.regions, args.eps, args.t_hist, args.horizon)
        name = (f"newcase: NewCaseEKFEstimatorWithOptimalNPI sweep, {args.regions} regions x {args.eps} cost weights x "
                f"({args.t_hist}+{args.horizon}) days")
    elif args.workload == "cfg3":
        w = synth.make_cfg3(args.regions, args.t_hist, region_offset=rank * args.regions)
        name = f"cfg3: SIAlphaModelEKF, {args.regions} regions x {args.t_hist} days"
    else:
        w = synth.make_cfg5(args.regions, args.eps, args.t_hist)
        name = f"cfg5: 3-state MC-EKS, {args.regions} regions x {args.eps} draws x {args.t_hist} days"
    return w, name


# stage times of the headline workload in the fast placement mode (profiles/r06/bench_distribution.txt; r04/alloc/): a run whose
# forward stage or smoother is above these ran on an allocation whose arrays collide in the L2 ("Placement")
PLACEMENT_SLOW_MS = {"ekf_fwd": 6.4, "eks_bwd": 7.1}


def placement_report(placement, args, ms, wname):
    """What the set-up did about the allocation-dependent stage times, and whether the run still sits in the slow mode."""
    rep = {"tries": None if placement is None else [round(t["sum_ms"], 3) for t in placement["tries"]],
           "chosen": None if placement is None else placement["chosen"]}
    headline = args.workload in ("cfg4", "cfg4-live") and (args.regions, args.eps, args.t_hist, args.horizon) == (300, 250, 400, 120) \
        and args.outputs == "all" and args.storage == "f64" and args.gpus == 1
    if headline:
        slow = [k for k, lim in PLACEMENT_SLOW_MS.items() if ms[k] > lim]
        rep["mode"] = "slow: " + ", ".join(slow) + " above the fast mode's stage times" if slow else "fast"
        rep["fast_mode_limits_ms"] = PLACEMENT_SLOW_MS
    return rep


def pmc_summary(args):
    """The committed rocprofv3 PMC summary of THIS workload and THESE kernel sources (profiles/r*/traffic_summary*.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of profiles/traffic_probe.py, corrected by the
    calibration copy as MI355X_MICROARCH.md prescribes), or None."""
    if args.workload not in ("cfg4", "cfg4-live") or (args.regions, args.eps, args.t_hist, args.horizon) != (300, 250, 400, 120):
        return None
    import glob
    tag = {"cfg4": "", "cfg4-live": "_live"}[args.workload] + ("" if args.outputs == "all" else "_reduced")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"traffic_summary{tag}.json")))
    if not files:
        return None
    summ = json.load(open(files[-1]))
    from epidemicmodeling_amd import _build
    if summ.get("kernel_src_sha16") != _build.source_hash():
        return None          # measured on other kernel sources than the ones this run executes: not quoted (stale)
    return summ


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from pmc_summary(); None if the bench is not running a profiled workload."""
    summ = pmc_summary(args)
    if summ is None:
        return None
    for name, v in summ["kernels"].items():
        if name.startswith(kernel):
            return v["hbm_bytes"]
    return None


def measured_copy_bandwidth(dev):
    """What this box's HBM delivers on a plain copy (state it beside the nominal 8 TB/s): 2 GiB read +
    2 GiB written per launch, once with the filter kernels' own access shape (8 B per lane, the library's calibration
    kernel) and once with torch's vectorised copy; HIP events, best of three, in GB/s of bytes moved."""
    import ctypes as C
    import torch
    from epidemicmodeling_amd import _lib
    n = 1 << 28
    src = torch.rand(n, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)

    def own():
        _lib.check(_lib.lib().epi_calib_copy_f64_device(src.data_ptr(), dst.data_ptr(), n, C.c_void_p(st.cuda_stream), err), err)
    out = {}
    for name, fn in (("copy_8B_per_lane_GBs", own), ("copy_torch_GBs", lambda: dst.copy_(src))):
        fn(); torch.cuda.synchronize(dev)
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(dev)
            best = min(best, a.elapsed_time(b))
        out[name] = 16.0 * n / (best * 1e-3) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    return out


def cpu_baseline(w, args):
    """The CPU oracle (C restatement of Tools/*.m, MATLAB unavailable) timed on this host's cores over a
    bounded sample of the same workload: every k-th chain, all days, all outputs."""
    from tests import helpers as H
    # a one-GPU box grants this job a 16-CPU share; never spawn more OpenMP threads than that
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    n = args.cpu_sample_chains
    if n <= 0:
        # pilot run to size the sample for ~12 s of CPU work on this host (capped by the whole workload)
        pilot = w.select(np.linspace(0, w.B - 1, min(w.B, cores * 64)).astype(np.int64))
        H.oracle_batch(pilot.select(np.arange(min(pilot.B, cores))), n_threads=cores)     # warm the library / threads
        t0 = time.perf_counter()
        H.oracle_batch(pilot, n_threads=cores)
        rate = pilot.B * pilot.T / (time.perf_counter() - t0)
        n = int(max(cores * 4, min(w.B, rate * 12.0 / w.T)))
    idx = np.linspace(0, w.B - 1, n).astype(np.int64)
    ws = w.select(idx)
    t0 = time.perf_counter()
    H.oracle_batch(ws, n_threads=cores)
    dt = time.perf_counter() - t0
    # the same restatement on ONE thread (SURVEY.md 8d asks for both), ~3 s worth of chains
    n1 = int(max(4, min(ws.B, ws.B * 3.0 / (dt * cores))))
    w1 = w.select(np.linspace(0, w.B - 1, n1).astype(np.int64))
    t1 = time.perf_counter()
    H.oracle_batch(w1, n_threads=1)
    dt1 = time.perf_counter() - t1
    return {"value": ws.B * ws.T / dt, "unit": "region-day EKF steps/s", "cores": cores, "kind": "port",
            "sample": f"{ws.B} of {w.B} chains (every {max(1, w.B // n)}th) x {ws.T} days, all 11 outputs, "
                      f"OpenMP over chains, {dt:.1f} s; C restatement of Tools/*.m (MATLAB unavailable)",
            "single_thread": {"value": w1.B * w1.T / dt1, "cores": 1, "sample": f"{w1.B} chains x {w1.T} days, {dt1:.1f} s"}}


def self_launch(n):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE): start the N ranks as fresh child processes -- one
    torch.distributed.run on 127.0.0.1 with a free port, same arguments -- BEFORE this process has imported torch or
    touched a GPU, relay their output (rank 0 prints the one JSON line) and return the launcher's exit code, which is
    non-zero as soon as any rank fails."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    import torch
    import torch.distributed as dist
    from epidemicmodeling_amd import batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (or start bench.py bare)")
    # one process per GPU; EPI_BENCH_BACKEND=gloo + fewer devices than ranks is only for rehearsing the N > 1
    # code path on a one-GPU box (ranks then share device 0 and the gather goes through host memory)
    backend = os.environ.get("EPI_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    w, wname = make_workload(args, rank)
    strong = args.scaling == "strong" and world > 1
    B_total = w.B * (1 if (strong or world == 1) else world)
    regions_total = w.Sx
    if strong:
        # the fixed sweep, sharded by contiguous chain blocks; every rank builds the same workload and
        # keeps its own block of chains (and only the series those chains use)
        if args.workload == "cfg3":
            raise SystemExit("cfg3 (300 chains) does not shard: replicas only (DESIGN.md 6)")
        lo, hi = batch.shard_chains(w.B, rank, world)
        w = w.select(np.arange(lo, hi))
    m = w.m
    outputs = None if args.outputs == "all" else ["u_opt_smooth", "S_SMOOTH"]
    dw = batch.DeviceWorkload(w, dev)
    runner = batch.EkfRunner(dw, outputs=outputs, extras=False, time_pipe=args.time_pipe, lane_block="auto" if args.lane_block < 0 else args.lane_block,
                             shape=args.shape, storage=args.storage)
    steps_per_pass = w.B * w.T
    t_hist_idx = w.meta.get("T_hist", w.T) - 1
    # set-up, outside the timed region like the allocation itself: keep the fastest of a few placements of the arrays
    placement = runner.tune_placement(args.placement_tries) if args.placement_tries > 1 else None

    # scenario-scoring tail of the sweep (TrainPredictPrescribeNPI.m:481-493): per-chain (J0, J1) are what leaves
    # the GPU at the end of a pass; with N > 1 they are gathered to rank 0 (the path's only collective)
    score = (not args.no_score) and args.workload in ("cfg4", "cfg4-live") and "u_opt_smooth" in runner.out
    score_state = {}

    def prepare_scoring():
        from epidemicmodeling_amd import layout as L_
        n, Bc = w.n_npi, w.B
        sp = torch.zeros((batch.SIM_PRM_COUNT, Bc), dtype=torch.float64, device=dev)
        prm = dw.prm
        sp[3], sp[4], sp[5] = prm[L_.PRM_ALPHA_MIN], prm[L_.PRM_ALPHA_MAX], prm[L_.PRM_GAMMA]
        sp[6], sp[7], sp[11] = prm[L_.PRM_B], prm[L_.PRM_BETA], 1.0
        sp[batch.SIM_A:batch.SIM_A + n] = prm[L_.PRM_A:L_.PRM_A + n]
        sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n] = prm[L_.PRM_U_MAX:L_.PRM_U_MAX + n]
        sp[batch.SIM_W:batch.SIM_W + n] = 1.0                       # npi_weights = ones (testPrescribeXPRIZE02.m:56)
        S = runner.unblocked("S_SMOOTH")
        th = t_hist_idx + 1
        # The scoring step's inputs -- s/i/alpha_historic(end) and the historic prefixes of the two NPICost sums -- come
        # from the region's 3-state run in the reference (TrainPredictPrescribeNPI.m:351-362, 481); here they are taken
        # once, outside the timed region, from the smoothed states of the first pass.
        sp[0:3].copy_(runner.unblocked_at("S_SMOOTH", t_hist_idx)[0:3])
        score_state["J0p"] = (S[:th, 0] * S[:th, 1] * S[:th, 2]).sum(dim=0)
        score_state["J1p"] = runner.unblocked("u_opt_smooth")[:th].sum(dim=(0, 1))
        score_state["sp"] = sp

    def one_step(events=None):
        """One pass of the hot path.  events None: what a user calls -- ONE library call (epi_sweep_run_device: forward
        kernel, pinv grid, smoother, and beside the smoother's pass over the observed days the scoring tail and the
        Pareto filter).  With events: the same work enqueued stage by stage, bracketed by HIP events."""
        whole = not strong                 # this rank holds whole regions: the library filters the front too
        if events is None and score and score_state:
            sc = runner.run_sweep(t_hist_idx + 1, score_state["sp"], score_state["J0p"], score_state["J1p"],
                                  n_regions=w.Sx if whole else None)
            if whole:
                score_state["front"] = (sc["on_front"], sc["i_opt"])
        else:
            if events is None:
                runner.run()
            else:
                e0, e1, e2, e3 = events
                e0.record(); runner.run(phase=1); e1.record(); runner.run(phase=3); e2.record()
                runner.run(phase=4); e3.record()
            sc = None
            if score and score_state:
                sc = batch.score_sweep(runner.out["u_opt_smooth"], t_hist_idx + 1, score_state["sp"], score_state["J0p"],
                                       score_state["J1p"], B=w.B)
                if whole:
                    score_state["front"] = batch.pareto_front(sc["J0"], sc["J1"], w.Sx)
        if world > 1:
            # the path's only collective, bracketed so that a scaling curve can be decomposed: HIP events on the launch
            # stream (over RCCL the collective is ordered against it) and the host's own clock (gloo rehearsal: the gather
            # goes through host memory and blocks the host)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tg = time.perf_counter()
            g0.record()
        if score and score_state:
            if strong:
                # a region's 250 cost weights may straddle two ranks: (J0, J1) of all shards are gathered to rank 0
                # (the path's only collective; shards padded to the common block length) and filtered there
                allj = batch.gather_shards_to_root(sc["JJ"] if "JJ" in sc else torch.stack([sc["J0"], sc["J1"]]), B_total)
                if world > 1:
                    g1.record()
                if rank == 0:
                    score_state["front"] = batch.pareto_front(allj[0].contiguous(), allj[1].contiguous(), regions_total)
            elif world > 1:
                batch.gather_to_root(sc["JJ"] if "JJ" in sc else torch.stack([sc["J0"], sc["J1"]]))
                g1.record()
        elif world > 1:
            # no scoring: gather the per-chain smoothed state at the last observed day to rank 0
            batch.gather_to_root(runner.unblocked_at("S_SMOOTH", t_hist_idx).contiguous())
            g1.record()
        if world > 1 and gather_log["on"]:
            gather_log["events"].append((g0, g1))
            gather_log["host_s"] += time.perf_counter() - tg

    gather_log = {"on": False, "events": [], "host_s": 0.0}
    # EPI_BENCH_STAGED=1: every pass -- the very first and the warm-up included -- is enqueued stage by stage, so that a
    # `rocprofv3 --stats` of the run sees ONE kind of launch per kernel and its averages are the per-kernel durations `roofline`
    # quotes (round 5's CSV held one overlapped call: its monitor launch, beside the pinv grid and the smoother on the helper
    # stream, spans both -- 9.45 ms against 0.67 alone; profiles/r06/monitor_outlier.txt)
    staged = os.environ.get("EPI_BENCH_STAGED") == "1"
    one_step([torch.cuda.Event(enable_timing=True) for _ in range(4)] if staged else None)
    torch.cuda.synchronize(dev)
    if score:
        prepare_scoring()
    spin_passes, spin_ms = 0, 0.0
    if args.spinup_ms > 0:        # set-up: the device at its steady clocks before the warm-up passes (see --spinup-ms)
        # passes until the wall clock says so (the first pass after set-up may be cold -- code objects, lazy initialisation --
        # so the count is not extrapolated from it); with N > 1 the passes contain the gather, so the ranks agree on every
        # further pass by an all-reduced flag
        t_spin = time.perf_counter()
        while True:
            one_step([torch.cuda.Event(enable_timing=True) for _ in range(4)] if staged else None)
            torch.cuda.synchronize(dev)
            spin_passes += 1
            more = (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms
            if world > 1:
                tn = torch.tensor([1 if more else 0], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                more = bool(tn.item())
            if not more:
                break
        spin_ms = (time.perf_counter() - t_spin) * 1e3
    for _ in range(args.warmup):
        one_step([torch.cuda.Event(enable_timing=True) for _ in range(4)] if staged else None)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    # The timed passes are the call a user makes (one epi_sweep_run_device per pass): inside it the library overlaps what is
    # off the critical path (monitor, scoring tail, Pareto filter; for a batch that leaves SIMDs idle also the pinv grids of
    # the forward pass's time segments).  The per-kernel durations come from K further passes enqueued stage by stage, each
    # bracketed by HIP events on the launch stream.  EPI_BENCH_STAGED=1: time the staged passes instead (round-1 behaviour).
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    gather_log["on"] = True
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(evs[k] if staged else None)
    torch.cuda.synchronize(dev)
    own_elapsed = time.perf_counter() - t0          # this rank's K passes, before it waits for the others
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gather_log["on"] = False
    per_rank = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # per-rank figures for decomposing a scaling curve: every rank's own ms per pass and its gather time
        g_ev = float(np.mean([a.elapsed_time(b) for a, b in gather_log["events"]])) if gather_log["events"] else 0.0
        mine = torch.tensor([own_elapsed / args.steps * 1e3, g_ev, gather_log["host_s"] / args.steps * 1e3], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        per_rank = {"ms_per_step_by_rank": [float(v) for v in allr[:, 0]],
                    "ms_per_step_min": float(allr[:, 0].min()), "ms_per_step_max": float(allr[:, 0].max()),
                    "gather_ms_hip_events_by_rank": [float(v) for v in allr[:, 1]],
                    "gather_ms_host_clock_by_rank": [float(v) for v in allr[:, 2]],
                    "backend": backend, "rccl_ranks": int(dist.get_world_size()),
                    "note": "ms_per_step_by_rank = each rank's own K passes (kernels + its part of the gather) before the closing "
                            "barrier; gather_ms = the end-of-sweep gather of (J0, J1) alone, per pass"}
    if not staged:
        for k in range(args.steps):
            one_step(evs[k])
        torch.cuda.synchronize(dev)
    ms_fwd = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    ms_pinv = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    ms_bwd = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))
    if rank == 0:
        total_steps = (B_total * w.T if strong else steps_per_pass * world) * args.steps
        value = total_steps / elapsed
        full = args.outputs == "all"
        # algorithmic bytes per region-day step, split by kernel: eks_pinv produces no reference
        # output of its own -- its algorithmic bytes are the P_MINUS it must read (m*m doubles)
        alg = {"ekf_fwd": BYTES_FWD[m] if full else 112, "eks_pinv": 8 * m * m,
               "eks_bwd": BYTES_BWD[m] if full else 8 * (m + 12)}
        if args.storage == "f32":       # outputs are 4-byte elements; the inputs x, u, R stay fp64 (112 B)
            alg["ekf_fwd"] = 112 + (alg["ekf_fwd"] - 112) // 2
            alg["eks_bwd"] //= 2
        if w.model.startswith("NewCase"):     # no u_opt_smooth output, no separate pinv stage (mrdivide inline)
            alg["eks_bwd"] = 8 * (m + m * m) if full else 8 * m
            alg["eks_pinv"] = 0
        ms = {"ekf_fwd": ms_fwd, "eks_pinv": ms_pinv, "eks_bwd": ms_bwd}
        dom = max(ms, key=ms.get)
        dom_bytes = alg[dom] * steps_per_pass
        achieved = dom_bytes / (ms[dom] * 1e-3) / 1e9
        step_bytes = (alg["ekf_fwd"] + alg["eks_bwd"]) * steps_per_pass
        step_gbs = step_bytes / (sum(ms.values()) * 1e-3) / 1e9
        traffic = pmc_traffic(dom, args)
        copy_bw = measured_copy_bandwidth(dev)
        res = {
            "metric": "region-day EKF steps/sec (300 regions x 400 days x 250 costs)",
            "value": value, "unit": "region-day EKF steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) and args.scaling == "strong" else "weak",
            "vs_baseline": None, "dtype": "f64" if args.storage == "f64" else "f64 arithmetic, f32 storage", "data": "synthetic",
            "config": {"workload": wname, "chains_per_gpu": w.B, "days": w.T, "outputs": args.outputs, "storage": args.storage, "time_pipe": args.time_pipe, "spinup_ms": args.spinup_ms, "spinup_passes": spin_passes, "spinup_ms_actual": round(spin_ms, 1),
                       "lane_block": runner.blk, "shape": ("wave (one wavefront per chain)" if m == 6 else "wave (seven 9-lane chains per wavefront)") if runner.blk == 1 and w.B > 1 else (("quad (4 lanes per chain)" if runner.blk == 16 else ("hex (6 lanes per chain)" if runner.blk == 10 else "lane (1 lane per chain)")) if m == 6 else "lane (1 lane per chain)"),
                       "sweep_chains_total": B_total,
                       "region_day_steps_per_pass_per_gpu": steps_per_pass,
                       "historic_only_steps_per_pass_per_gpu": w.B * (t_hist_idx + 1),
                       "scoring_tail": bool(score),
                       "placement": placement_report(placement, args, ms, wname),
                       "parallelism": (f"the fixed sweep's {B_total} chains sharded over {world} GPU(s) by contiguous blocks; "
                                       "end-of-sweep gather of (J0, J1) to rank 0, Pareto filter there") if strong or world == 1 else
                                      f"every rank its own full sweep ({world} x {w.B} chains); end-of-sweep gather of (J0, J1) to rank 0"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": dom,
                         # what the kernel actually moves (PMC bytes of the committed profile / this run's duration):
                         # the smoother reads S+, P+, X back (round 6: no longer S-, P-), ~1.9x the algorithmic figure
                         "traffic_GBs": None if traffic is None else traffic / (ms[dom] * 1e-3) / 1e9,
                         "traffic_frac_of_peak": None if traffic is None else traffic / (ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "measured_copy": copy_bw,     # this box, this process: bytes read + written per second
                         "kernel_ms": ms[dom], "algorithmic_bytes_per_launch": dom_bytes,
                         "limiter": {"ekf_fwd": "HBM writes: 32.4 GB at the box's fill rate; a lone wave per SIMD issues 73 % of the time (profiles/r06/valu_summary.json)",
                                     "eks_pinv": "fp64 VALU issue (SIMD VALU busy 82 % of the kernel's duration, three waves per SIMD) on top of 13.5 GB of traffic; profiles/r06/valu_summary.json",
                                     "eks_bwd": "instruction issue of a lone wave per SIMD in two rounds (a wave issues 70 % of its cycles, waits for memory 9 %) and HBM (32.6 GB at 4.9-5.1 TB/s, the first round at the box's copy rate); profiles/r06/valu_summary.json, traffic_summary.json"}[dom]},
            "kernels": {**{k + "_ms": v for k, v in ms.items()},
                        **{k + "_GBs": alg[k] * steps_per_pass / (ms[k] * 1e-3) / 1e9 for k in ms},
                        "whole_step_algorithmic_bytes": step_bytes, "whole_step_GBs": step_gbs,
                        "whole_step_frac_of_hbm_peak": step_gbs / HBM_PEAK_GBS},
        }
        if per_rank is not None:
            res["ranks"] = per_rank
        # a figure that can be compared across boxes of different HBM speed: the pass time in units of what this box needs
        # to move the pass's MEASURED traffic at its own copy bandwidth (1.0 = the pass runs at the copy rate on the bytes
        # its kernels really move; profiles/r*/traffic_summary*.json, same kernel sources)
        summ = pmc_summary(args) if world == 1 else None
        if summ is not None:
            tot = float(sum(v["hbm_bytes"] for v in summ["kernels"].values()))
            res["box_normalised"] = {"traffic_bytes_per_pass": tot, "copy_GBs": copy_bw["copy_8B_per_lane_GBs"],
                                     "pass_ms_at_copy_rate": tot / (copy_bw["copy_8B_per_lane_GBs"] * 1e9) * 1e3,
                                     "pass_over_copy_floor": (elapsed / args.steps) / (tot / (copy_bw["copy_8B_per_lane_GBs"] * 1e9))}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(w, args)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
