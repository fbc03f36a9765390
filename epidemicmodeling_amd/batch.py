"""Batched device-resident execution of the EKF/EKS hot path.

PyTorch is used here only as plumbing -- device memory (torch tensors), the current HIP stream and,
for N > 1 GPUs, torch.distributed over RCCL.  All arithmetic happens in libepiekf.so's HIP kernels,
reached through the C ABI of include/epiekf.h with raw device pointers."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import layout as L

OUT_NAMES = ["u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS", "P_SMOOTH",
             "K_GAIN", "innovations", "rho"]


def out_mask_of(names) -> int:
    m = 0
    for n in names:
        m |= L.OUT_BITS[n]
    return m


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class DeviceWorkload:
    """Inputs of one batched filter problem resident in HBM (built once, run many times)."""

    def __init__(self, w, device="cuda:0"):
        self.device = torch.device(device)
        self.model, self.T, self.n_npi, self.L, self.order, self.obs_type = w.model, w.T, w.n_npi, w.L, w.order, w.obs_type
        self.m = L.MODEL_DIM[w.model]
        self.B, self.Sx, self.Su = w.B, w.Sx, w.Su
        f = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.device)
        i = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(self.device)
        self.x, self.u, self.R_series, self.R_scalar = f(w.x), f(w.u), f(w.R_series), f(w.R_scalar)
        self.x_series, self.u_series = i(w.x_series), i(w.u_series)
        self.prm, self.s_init, self.Ps_init = f(w.prm), f(w.s_init), f(w.Ps_init)
        self.s_final, self.Ps_final, self.Q = f(w.s_final), f(w.Ps_final), f(w.Q)
        self.r_mode = 1 if w.R_series is not None else 0
        self.q_mode = 1 if np.ndim(w.Q) == 3 else 0     # Q [T][m*m][B]: Q(:,:,k) per filter step

    def inputs_struct(self):
        s = _lib.Inputs()
        for n in ("x_series", "u_series", "x", "u", "R_series", "R_scalar", "prm", "s_init", "Ps_init",
                  "s_final", "Ps_final", "Q"):
            setattr(s, n, _ptr(getattr(self, n)))
        return s

    def input_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.x, self.u, self.R_series, self.R_scalar, self.prm,
                                                          self.s_init, self.Ps_init, self.s_final, self.Ps_final,
                                                          self.Q, self.x_series, self.u_series) if t is not None)


class EkfRunner:
    """Pre-allocated outputs + workspace for a DeviceWorkload; run() only enqueues kernels."""

    def __init__(self, dw: DeviceWorkload, outputs=None, extras=False, time_pipe=0, precheck=True, lane_block=0, shape=0,
                 storage="f64", exact_nonfinite=None, slab=None, test_window=0, test_flags=0):
        """time_pipe: epi_batch_desc.time_pipe (0 = the library decides whether a full call runs its forward kernel in time
        segments with the pinv grid of each segment beside the next, 1 = on, -1 = off).  precheck: ask the
        library once (synchronously) whether the batch qualifies for the symmetric-packed kernels, so that
        run() enqueues only the variant that will actually execute.  lane_block: 0 = classic [T][rows][B] outputs;
        8 / "auto" (= chains per wavefront of the launch) = chain-blocked outputs (epi_batch_desc.lane_block): `out` then holds the raw blocked tensors
        [T, nblk, rows, blk] ([T, nblk*blk] for one-row arrays) and unblocked() returns [T, rows, B] copies."""
        self.dw = dw
        names = list(OUT_NAMES) if outputs is None else list(outputs)
        if dw.model.startswith("NewCase") and "u_opt_smooth" in names:
            names.remove("u_opt_smooth")          # the reference has no such output (NewCase...m:1)
        self.names = names
        self.mask = out_mask_of(names)
        self.desc = _lib.make_desc(dw.model, dw.B, dw.T, dw.Sx, dw.Su, dw.n_npi, dw.L, dw.order, dw.obs_type,
                                   dw.r_mode, self.mask, dw.q_mode)
        # epi_batch_desc.shape: 0 = by batch size, 1 / 2 = one / four lanes per chain (6-state generic models)
        self.desc.shape = {"auto": 0, "lane": 1, "quad": 2, "wave": 3, "hex": 4}.get(shape, shape)
        # epi_batch_desc.storage: "f32" = outputs stored as float32 (each the fp64 result rounded once; BASELINE config 5)
        self.desc.storage = {"f64": 0, "f32": 1}[storage]
        # epi_batch_desc.exact_nonfinite: chains whose covariance overflows are run again by the dense kernels, in place
        # (their Inf / NaN pattern is then the dense evaluation's, i.e. the reference's)
        # None = the library's default (on whenever the smoother runs), True = always, False = off
        self.desc.exact_nonfinite = 0 if exact_nonfinite is None else (1 if exact_nonfinite else -1)
        # test hooks (epi_batch_desc.test_window / test_flags, 0 in production): short addressing windows, forced reverse-time pipeline
        self.desc.test_window, self.desc.test_flags = int(test_window), int(test_flags)
        odt = torch.float32 if storage == "f32" else torch.float64
        if lane_block == "auto":       # one block per wavefront of the launch
            lane_block = int(_lib.lib().epi_ekf_preferred_lane_block(C.byref(self.desc)))
        self.blk = dw.B if (lane_block <= 0 or lane_block >= dw.B) else int(lane_block)
        self.nblk = (dw.B + self.blk - 1) // self.blk
        self.desc.lane_block = 0 if self.blk == dw.B else self.blk
        self.err = C.create_string_buffer(256)
        h = _lib.lib()
        _lib.check(h.epi_ekf_validate(C.byref(self.desc), self.err), self.err)
        dev = dw.device
        self.out = {}
        shapes = {}
        for n in names:
            rows = L.out_rows(n, dw.m, dw.n_npi)
            if self.blk == dw.B:
                shapes[n] = (dw.T, dw.B) if rows == 0 else (dw.T, rows, dw.B)
            else:
                shapes[n] = (dw.T, self.nblk * self.blk) if rows == 0 else (dw.T, self.nblk, rows, self.blk)
        self.ws_bytes = int(h.epi_ekf_workspace_bytes(C.byref(self.desc)))
        self._shapes, self._odt, self._slab_opt = shapes, odt, slab
        self._allocate()
        self.pinv_rank = torch.empty((dw.T, self.nblk * self.blk), dtype=torch.int32, device=dev) if extras else None
        self.status = torch.zeros((dw.B,), dtype=torch.int32, device=dev) if extras else None
        self.ins = dw.inputs_struct()
        self.desc.time_pipe = int(time_pipe)
        self._sweep = None
        if precheck:
            ok = C.c_int(0)
            st = torch.cuda.current_stream(dev)
            rc = h.epi_ekf_precheck_device(C.byref(self.desc), C.byref(self.ins), C.c_void_p(st.cuda_stream),
                                           C.byref(ok), self.err)
            _lib.check(rc, self.err)
            self.desc.path_hint = 1 if ok.value else 2
        self._bind()

    def _allocate(self):
        """The outputs and the workspace (fresh device memory; `out` / `ws` are replaced)."""
        dw, names, shapes, odt, slab = self.dw, self.names, self._shapes, self._odt, self._slab_opt
        dev = dw.device
        self.out = {}
        ws_elems = (max(self.ws_bytes, 8) + 7) // 8
        if slab is None:
            for n in names:
                self.out[n] = torch.empty(shapes[n], dtype=odt, device=dev)
            self.ws = torch.empty(ws_elems, dtype=torch.float64, device=dev)
            self._slab = None
        else:
            # ONE device allocation for every output and the workspace (slab = {"align": bytes, "stagger": bytes}): array i
            # starts at a multiple of `align` plus i * `stagger` -- the arrays' relative placement is then the caller's choice
            # instead of the allocator's (profiles/alloc_probe.py)
            align, stagger = int(slab.get("align", 2 << 20)), int(slab.get("stagger", 0))
            isz = torch.empty((), dtype=odt).element_size()
            offs, off = {}, 0
            for i, n in enumerate(names + ["__ws__"]):
                nbytes = ws_elems * 8 if n == "__ws__" else int(np.prod(shapes[n])) * isz
                off = (off + align - 1) // align * align + i * stagger
                offs[n] = (off, nbytes)
                off += nbytes
            self._slab = torch.empty(off, dtype=torch.uint8, device=dev)
            for n in names:
                o, nb = offs[n]
                self.out[n] = self._slab[o:o + nb].view(odt).view(shapes[n])
            o, nb = offs["__ws__"]
            self.ws = self._slab[o:o + nb].view(torch.float64)

    def _bind(self):
        self.outs = _lib.Outputs()
        for n in OUT_NAMES:
            setattr(self.outs, n, _ptr(self.out.get(n)))
        self.outs.pinv_rank = _ptr(self.pinv_rank)
        self.outs.status = _ptr(self.status)

    def stage_ms(self, passes=2, min_ms=0.0, stream=None):
        """(forward, pinv, smoother) milliseconds of a pass enqueued stage by stage -- epi_ekf_time_stages_device: HIP events on
        the stream, one untimed round, then the mean over as many rounds as fill `min_ms` of device time (at least `passes`)."""
        st = torch.cuda.current_stream(self.dw.device) if stream is None else stream
        ms = (C.c_double * 3)()
        best = None
        # the library times until min_ms is filled; `passes` rounds at least: ask again while fewer were averaged
        need = float(min_ms)
        for _ in range(max(1, int(passes))):
            rc = _lib.lib().epi_ekf_time_stages_device(C.byref(self.desc), C.byref(self.ins), C.byref(self.outs), _ptr(self.ws),
                                                       self.ws_bytes, C.c_void_p(st.cuda_stream), need, ms, self.err)
            _lib.check(rc, self.err)
            cur = (float(ms[0]), float(ms[1]), float(ms[2]))
            best = cur if best is None else tuple((a + b) for a, b in zip(best, cur))
            if need > 0:
                return cur
        return tuple(x / max(1, int(passes)) for x in best)

    def tune_placement(self, tries=3, spinup_ms=150.0):
        """Where the allocator puts the ~14 arrays a pass streams concurrently changes the forward kernel's and the smoother's
        time by up to 15 % (they meet in the physically indexed L2's sets and banks or they do not: DESIGN.md 5, "where the
        arrays lie"), it is a property of the ALLOCATION -- the same arrays give the same time run after run -- and the caller
        cannot see it.  So: time a staged pass on this allocation, allocate the outputs and the workspace again (`tries` - 1
        times, each while the earlier ones are still held, so that other memory is handed out), keep the fastest and free the
        rest.  One-time set-up cost: a few passes and, transiently, `tries` x the outputs' memory (skipped when that does not
        fit).  The device is brought to its steady clocks first (`spinup_ms` of passes: after idle the first 50-150 ms of work
        run up to 15 % slower, which would make the FIRST try look bad whatever its placement) and every try is timed over at
        least 15 ms of work.  Returns {"tries": [...ms per try...], "chosen": i}.
        Call it BEFORE anything keeps a reference to `out` / `ws`: when another allocation than the first is kept (`generation`
        is then incremented), tensors taken from `out` earlier and HIP graphs captured from earlier run() calls still point at
        memory this runner no longer writes.  A later allocation is kept only if it wins by more than 1 %."""
        import time
        dev = self.dw.device
        if int(tries) > 1 and spinup_ms > 0:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < spinup_ms:      # stage by stage, like the timed tries: one kind of launch per kernel
                self.run(phase=1); self.run(phase=3); self.run(phase=4)
                torch.cuda.synchronize(dev)
        held, log = [], []
        need = sum(t.numel() * t.element_size() for t in self.out.values()) + self.ws.numel() * 8
        for i in range(max(1, int(tries))):
            if i > 0:
                free, _ = torch.cuda.mem_get_info(dev)
                if free < need * 1.05:
                    break
                self._allocate()
                self._bind()
            f, p, b = self.stage_ms(min_ms=15.0)
            log.append({"fwd_ms": f, "pinv_ms": p, "bwd_ms": b, "sum_ms": f + p + b})
            held.append((self.out, self.ws, self._slab))
        best = int(np.argmin([x["sum_ms"] for x in log]))
        if best != 0 and log[best]["sum_ms"] > 0.99 * log[0]["sum_ms"]:
            best = 0                      # inside the timing noise: the first allocation stays
        self.generation = getattr(self, "generation", 0) + (1 if best != 0 else 0)
        self.out, self.ws, self._slab = held[best]
        self._bind()
        del held
        torch.cuda.empty_cache()
        return {"tries": log, "chosen": best}

    def run(self, stream=None, phase: int = 0):
        """Enqueue forward + backward kernels on `stream` (default: torch's current stream).
        phase 1 / 2 enqueue only the forward / only the smoother kernel (per-kernel timing)."""
        st = torch.cuda.current_stream(self.dw.device) if stream is None else stream
        self.desc.phase = phase
        rc = _lib.lib().epi_ekf_run_device(C.byref(self.desc), C.byref(self.ins), C.byref(self.outs),
                                           _ptr(self.ws), self.ws_bytes, C.c_void_p(st.cuda_stream), self.err)
        _lib.check(rc, self.err)
        return self.out

    def run_sweep(self, t_hist, sp, J0_prefix, J1_prefix, n_regions=None, stream=None):
        """epi_sweep_run_device: the full filter call followed by the sweep's scoring tail (TrainPredictPrescribeNPI.m:
        481-493) on the last T - t_hist days of the u_opt_smooth it wrote and, when `n_regions` is given (the batch then
        holds n_regions x P chains, region-major), the Pareto filter and optimum per region (:624-633) -- one library
        call, scoring and filter enqueued beside the smoother's pass over the observed days.  sp [48, B], J0_prefix /
        J1_prefix [B] as for score_sweep.  Returns dict J0, J1 [B] (views of JJ [2, B]) (+ on_front bool-able int32 [R, P],
        i_opt int32 [R])."""
        dev = self.dw.device
        B = self.dw.B
        if self._sweep is None:
            jj = torch.empty((2, B), dtype=torch.float64, device=dev)     # (J0; J1) side by side: what the end-of-sweep gather sends
            self._sweep = {"JJ": jj, "J0": jj[0], "J1": jj[1]}
        res = self._sweep
        sd = _lib.SweepDesc()
        sd.abi_version, sd.t_hist = _lib.ABI_VERSION, int(t_hist)
        if n_regions:
            P = B // int(n_regions)
            if P * int(n_regions) != B:
                raise ValueError("the batch does not hold the same number of cost weights for every region")
            sd.R, sd.P = int(n_regions), P
            if "on_front" not in res or tuple(res["on_front"].shape) != (int(n_regions), P):
                res["on_front"] = torch.empty((int(n_regions), P), dtype=torch.int32, device=dev)
                res["i_opt"] = torch.empty((int(n_regions),), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev) if stream is None else stream
        self.desc.phase = 0
        rc = _lib.lib().epi_sweep_run_device(C.byref(self.desc), C.byref(self.ins), C.byref(self.outs), _ptr(self.ws), self.ws_bytes,
                                             C.byref(sd), _ptr(sp), _ptr(J0_prefix), _ptr(J1_prefix), _ptr(res["J0"]), _ptr(res["J1"]),
                                             _ptr(res.get("on_front")) if n_regions else None,
                                             _ptr(res.get("i_opt")) if n_regions else None, C.c_void_p(st.cuda_stream), self.err)
        _lib.check(rc, self.err)
        return res

    def unblocked(self, name):
        """[T, rows, B] ([T, B]) tensor of output `name` whatever the layout run() wrote it in."""
        t = self.pinv_rank if name == "pinv_rank" else self.out[name]
        if self.blk == self.dw.B:
            return t
        if t.dim() == 2:
            return t[:, :self.dw.B]
        T, nb, rows, blk = t.shape
        return t.permute(0, 2, 1, 3).reshape(T, rows, nb * blk)[:, :, :self.dw.B]

    def unblocked_at(self, name, t):
        """[rows, B] slice of output `name` at time index t (no copy of the whole array)."""
        x = self.out[name][t]
        if self.blk == self.dw.B:
            return x
        if x.dim() == 1:
            return x[:self.dw.B]
        nb, rows, blk = x.shape
        return x.permute(1, 0, 2).reshape(rows, nb * blk)[:, :self.dw.B]

    def output_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.out.values())


def run_workload(w, outputs=None, device="cuda:0", extras=True, time_pipe=0, precheck=True, lane_block=0, shape=0, storage="f64",
                 exact_nonfinite=None, test_window=0, test_flags=0):
    """Convenience: upload `w`, run once, return dict name -> numpy array [T, rows, B] (+ pinv_rank/status)."""
    dw = DeviceWorkload(w, device)
    r = EkfRunner(dw, outputs, extras=extras, time_pipe=time_pipe, precheck=precheck, lane_block=lane_block, shape=shape, storage=storage,
                  exact_nonfinite=exact_nonfinite, test_window=test_window, test_flags=test_flags)
    r.run()
    torch.cuda.synchronize(dw.device)
    res = {n: r.unblocked(n).cpu().numpy() for n in r.out}
    if extras:
        res["pinv_rank"] = r.unblocked("pinv_rank").cpu().numpy()
        res["status"] = r.status.cpu().numpy()
    return res


def shard_chains(B: int, rank: int, world: int):
    """Contiguous block partition of the chain axis: chains are independent, so a rank
    only ever needs its own block; returns (start, stop)."""
    per = (B + world - 1) // world
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def gather_to_root(t: torch.Tensor, group=None, dst: int = 0):
    """The path's only collective: gather per-rank result shards (chain-minor tensors of equal shape)
    to rank `dst` at the end of a sweep.  Over RCCL this is one send per peer on its own xGMI link.  The collective is
    issued whatever the world size (a one-rank world runs it too: same code path as N ranks)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl":
        t = t.cpu()                     # gloo rehearsal / CPU tests: the gather goes through host memory
    bufs = [torch.empty_like(t) for _ in range(world)] if dist.get_rank(group) == dst else None
    dist.gather(t, bufs, dst=dst, group=group)      # RCCL: one send per peer, each on its own xGMI link
    return bufs


def blocks_side_by_side(buf: torch.Tensor) -> torch.Tensor:
    """[world, ..., per] (what all_gather_into_tensor fills: one padded block per rank) -> [..., world * per], the blocks
    next to each other in rank (= chain) order."""
    world, per = buf.shape[0], buf.shape[-1]
    return buf.movedim(0, -2).reshape(tuple(buf.shape[1:-1]) + (world * per,))


_GATHER_BUFS = {}       # (device, dtype, shape of a padded block, world) -> (send block, receive buffer): allocated once per sweep shape


def gather_shards_to_root(t: torch.Tensor, B_total: int, group=None, dst: int = 0):
    """Strong-scaling form of the end-of-sweep gather: `t` [..., n_r] holds this rank's block of a chain-minor result whose
    blocks come from shard_chains(B_total, rank, world) -- all of length ceil(B_total / world) except a shorter (possibly
    empty) last one.  Blocks are padded to the common length, gathered to rank `dst` (one message per peer) and
    reassembled there in chain order; returns the [..., B_total] tensor on `dst`, None elsewhere.  The padded send block and
    the receive buffer are allocated on the first call for a shape and reused: a pass costs one copy into the send block
    (only when the block is short or not contiguous) and the collective, no allocation."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = (B_total + world - 1) // world
    nccl = dist.get_backend(group) == "nccl"
    if t.shape[-1] == per and t.is_contiguous():
        send = t
        key = (t.device, t.dtype, tuple(t.shape), world)
        if nccl and key not in _GATHER_BUFS:
            _GATHER_BUFS[key] = (None, torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device))
    else:
        shape = tuple(t.shape[:-1]) + (per,)
        key = (t.device, t.dtype, shape, world)
        if key not in _GATHER_BUFS or _GATHER_BUFS[key][0] is None:
            _GATHER_BUFS[key] = (torch.zeros(shape, dtype=t.dtype, device=t.device),
                                 torch.empty((world,) + shape, dtype=t.dtype, device=t.device) if nccl else None)
        send = _GATHER_BUFS[key][0]
        send[..., :t.shape[-1]].copy_(t)          # the padding stays zero
    if nccl:
        # RCCL: one ncclAllGather of the small per-chain summaries (16 B per chain) -- every rank receives
        # them, rank `dst` uses them.  Issued for every world size, one rank included.
        buf = _GATHER_BUFS[key][1]
        dist.all_gather_into_tensor(buf, send, group=group)
        if dist.get_rank(group) != dst:
            return None
        return blocks_side_by_side(buf)[..., :B_total]
    parts = gather_to_root(send, group=group, dst=dst)
    if dist.get_rank(group) != dst:
        return None
    return torch.cat([p_.to(t.device) for p_ in parts], dim=-1)[..., :B_total]


# ---------------------------------------------------------------------------
# forward simulators (Tools/SIalpha_Controlled.m, Tools/SEIRP.m, Tools/NPICost.m) on the device
# ---------------------------------------------------------------------------
SIM_FIELDS = {"s0": 0, "i0": 1, "alpha0": 2, "alpha_min": 3, "alpha_max": 4, "gamma": 5, "b": 6, "beta": 7,
              "s_noise_std": 8, "i_noise_std": 9, "alpha_noise_std": 10, "dt": 11}
SIM_A, SIM_U_MAX, SIM_W, SIM_PRM_COUNT = 12, 24, 36, 48


def sialpha_sim(u, sp, z=None, u_series=None, with_cost=False, store=True, device="cuda:0"):
    """Batched SIalpha_Controlled (+ fused NPICost).  u [K, n_npi, Su], sp [48, B] (SIM_* rows), z [K, 3, B]
    standard-normal draws or None.  Returns dict of torch tensors s,i,alpha [K,B] (+ J0,J1 [B])."""
    dev = torch.device(device)
    t = lambda a, dt=torch.float64: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev) \
        if not isinstance(a, torch.Tensor) else a
    u, sp, z = t(u), t(sp), t(z)
    us = t(u_series, torch.int32)
    K, n_npi, Su = u.shape
    B = sp.shape[1]
    d = _lib.SimDesc()
    d.abi_version, d.B, d.K, d.Su, d.n_npi = _lib.ABI_VERSION, B, K, Su, n_npi
    d.noise, d.with_cost, d.prefix_days = int(z is not None), int(with_cost), 0
    out = {}
    if store:
        for n in ("s", "i", "alpha"):
            out[n] = torch.empty((K, B), dtype=torch.float64, device=dev)
    if with_cost:
        out["J0"] = torch.empty((B,), dtype=torch.float64, device=dev)
        out["J1"] = torch.empty((B,), dtype=torch.float64, device=dev)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_sialpha_sim_device(C.byref(d), _ptr(us), _ptr(u), _ptr(sp), _ptr(z), _ptr(out.get("s")),
                                           _ptr(out.get("i")), _ptr(out.get("alpha")), _ptr(out.get("J0")),
                                           _ptr(out.get("J1")), C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return out


def random_npi_mc(sp, u_min, n_scen, K, seed=0, z=None, J0_prefix=None, J1_prefix=None, prefix_days=0,
                  store_u=False, device="cuda:0"):
    """Random-NPI Monte-Carlo scenarios (Tools/TrainPredictPrescribeNPI.m:496-521) on the device.

    sp [48, R] per-region SIM_* rows (end-of-history state, model constants, SIM_U_MAX = NPI_MAXES, SIM_W = NPICost
    weights), u_min [n_npi, R] = NPI_MINS, z [K, 3, n_scen*R] standard-normal draws or None; J0_prefix/J1_prefix [R]
    sequential historic sums over prefix_days.  Returns dict J0, J1 [n_scen, R] (+ u [K, n_npi, n_scen*R])."""
    dev = torch.device(device)
    t = lambda a: None if a is None else (a if isinstance(a, torch.Tensor) else
                                          torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev))
    sp, u_min, z, J0p, J1p = t(sp), t(u_min), t(z), t(J0_prefix), t(J1_prefix)
    n_npi, R = u_min.shape
    d = _lib.McDesc()
    d.abi_version, d.R, d.n_scen, d.K, d.n_npi = _lib.ABI_VERSION, R, int(n_scen), int(K), n_npi
    d.noise, d.prefix_days = int(z is not None), int(prefix_days)
    d.seed_lo, d.seed_hi = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    out = {"J0": torch.empty((n_scen, R), dtype=torch.float64, device=dev),
           "J1": torch.empty((n_scen, R), dtype=torch.float64, device=dev)}
    if store_u:
        out["u"] = torch.empty((K, n_npi, n_scen * R), dtype=torch.float64, device=dev)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_random_npi_mc_device(C.byref(d), _ptr(sp), _ptr(u_min), _ptr(z), _ptr(J0p), _ptr(J1p),
                                             _ptr(out.get("u")), _ptr(out["J0"]), _ptr(out["J1"]),
                                             C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return out


def pareto_front(J0, J1, n_regions):
    """Pareto-front filter and optimum of the sweep (Tools/TrainPredictPrescribeNPI.m:624-633) per region.
    J0, J1: torch [B] in the sweep's chain order (region-major).  Returns (on_front bool [R, P], i_opt int [R],
    0-based)."""
    dev = J0.device
    B = J0.numel()
    P = B // n_regions
    if P * n_regions != B:
        raise ValueError("J0 does not hold the same number of points for every region")
    on = torch.empty((n_regions, P), dtype=torch.int32, device=dev)
    io = torch.empty((n_regions,), dtype=torch.int32, device=dev)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_pareto_front_device(n_regions, P, _ptr(J0.contiguous()), _ptr(J1.contiguous()), _ptr(on),
                                            _ptr(io), C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return on.bool(), io


def si_controlled(alpha, beta, s0, i0, K, dt, alpha_series=None, device="cuda:0"):
    """Batched Tools/SI_Controlled.m: alpha [K-1, Sa], beta / s0 / i0 [B].  Returns (s, i) torch [K, B]."""
    dev = torch.device(device)
    t = lambda a, dt_=torch.float64: None if a is None else (a.contiguous() if isinstance(a, torch.Tensor) else
                                                             torch.as_tensor(np.ascontiguousarray(a), dtype=dt_).to(dev))
    al, ser = t(alpha), t(alpha_series, torch.int32)
    prm = t(np.stack([np.asarray(beta, dtype=np.float64), np.asarray(s0, dtype=np.float64), np.asarray(i0, dtype=np.float64)]))
    B = prm.shape[1]
    if al.shape[0] < K - 1:
        raise IndexError("Index exceeds the number of array elements (alpha).")
    s = torch.empty((K, B), dtype=torch.float64, device=dev); i = torch.empty_like(s)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_si_controlled_device(B, int(K), al.shape[1], float(dt), _ptr(ser), _ptr(al), _ptr(prm), _ptr(s), _ptr(i),
                                             C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return s, i


def npi_cost(newcases, inputs, weights, u_series=None, device="cuda:0"):
    """Batched Tools/NPICost.m: newcases [T, B], inputs [T, n_npi, Su], weights [T, n_npi, B] or [n_npi, B] (the same
    every day).  Returns (J0, J1) torch [B]."""
    dev = torch.device(device)
    t = lambda a, dt=torch.float64: None if a is None else (a.contiguous() if isinstance(a, torch.Tensor) else
                                                            torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev))
    nc, u, w, us = t(newcases), t(inputs), t(weights), t(u_series, torch.int32)
    T, B = nc.shape
    n_npi, Su = u.shape[1], u.shape[2]
    if u.shape[0] != T or w.shape[-1] != B or w.shape[-2] != n_npi or (w.dim() == 3 and w.shape[0] != T):
        raise ValueError("NPICost: newcases, inputs and weights do not agree in size")
    J0 = torch.empty((B,), dtype=torch.float64, device=dev); J1 = torch.empty_like(J0)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_npi_cost_device(B, T, n_npi, Su, int(w.dim() == 3), _ptr(us), _ptr(nc), _ptr(u), _ptr(w), _ptr(J0),
                                        _ptr(J1), C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return J0, J1


def preprocess(cases, population, deaths=None, ip=None, W=7, min_cases=1.0, first_num_days=7, outputs=None,
               device="cuda:0"):
    """Per-region preprocessing on the device (Tools/TrainPredictPrescribeNPI.m:142-198,201-202,240).

    cases / deaths: cumulative confirmed counts [T, S] (NaN = missing), population [S], ip [T, n_npi, S] (NaN = N/A).
    Returns dict of torch tensors in the filters' input layout: x_new / x_total / R_v / new_refined / new_smoothed /
    zero_lag / fatality [T, S], I0 [S], ip_filled [T, n_npi, S] (those in `outputs`; default: all that apply)."""
    dev = torch.device(device)
    t = lambda a: None if a is None else (a if isinstance(a, torch.Tensor) else
                                          torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev))
    cases, population, deaths, ip = t(cases), t(population), t(deaths), t(ip)
    T, S = cases.shape
    names = list(_lib.PRE_OUT_NAMES) if outputs is None else list(outputs)
    if deaths is None and "fatality" in names:
        names.remove("fatality")
    if ip is None and "ip_filled" in names:
        names.remove("ip_filled")
    d = _lib.PreDesc()
    d.abi_version, d.S, d.T, d.n_npi = _lib.ABI_VERSION, S, T, 0 if ip is None else ip.shape[1]
    d.W, d.first_num_days, d.min_cases = int(W), int(first_num_days), float(min_cases)
    out = {}
    for n in names:
        shape = (S,) if n == "I0" else (tuple(ip.shape) if n == "ip_filled" else (T, S))
        out[n] = torch.empty(shape, dtype=torch.float64, device=dev)
    outs = _lib.PreOutputs()
    for n in _lib.PRE_OUT_NAMES:
        setattr(outs, n, _ptr(out.get(n)))
    h = _lib.lib()
    wsb = int(h.epi_preprocess_workspace_bytes(C.byref(d)))
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.float64, device=dev)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = h.epi_preprocess_device(C.byref(d), _ptr(cases), _ptr(deaths), _ptr(population), _ptr(ip), C.byref(outs),
                                 _ptr(ws), wsb, C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return out


def nnls_affine_fit(X, y, max_iters=100, device="cuda:0"):
    """Regression between the EKF rounds on the device (Tools/TrainPredictPrescribeNPI.m:251-276, 'NONNEGATIVELS').
    X [D, n, S] = NPI_MAXES - InterventionPlans over the regression window, y [D, S] = smoothed alpha.
    Returns dict of torch tensors a [n, S], b [S], min_err [S], iters [S], flag [S]."""
    dev = torch.device(device)
    t = lambda v: v if isinstance(v, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float64).to(dev)
    X, y = t(X).contiguous(), t(y).contiguous()
    D, n, S = X.shape
    d = _lib.NnlsDesc()
    d.abi_version, d.S, d.D, d.n, d.max_iters = _lib.ABI_VERSION, S, D, n, int(max_iters)
    out = {"a": torch.empty((n, S), dtype=torch.float64, device=dev), "b": torch.empty((S,), dtype=torch.float64, device=dev),
           "min_err": torch.empty((S,), dtype=torch.float64, device=dev),
           "iters": torch.empty((S,), dtype=torch.int32, device=dev), "flag": torch.empty((S,), dtype=torch.int32, device=dev)}
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_nnls_affine_fit_device(C.byref(d), _ptr(X), _ptr(y), _ptr(out["a"]), _ptr(out["b"]),
                                               _ptr(out["min_err"]), _ptr(out["iters"]), _ptr(out["flag"]),
                                               C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return out


RT_OUT_ROWS = {"S_MINUS": 2, "S_PLUS": 2, "P_MINUS": 4, "P_PLUS": 4, "K_GAIN": 2, "S_SMOOTH": 2, "P_SMOOTH": 4,
               "innovations": 0, "rho": 0}


class RtRunner:
    """Batched Tools/Rt_ExpFitEKF.m on the device: inputs of a synth.RtWorkload resident in HBM, outputs
    pre-allocated; run() only enqueues rt_expfit_fwd (+ rt_expfit_bwd when a smoothed output is selected)."""

    def __init__(self, w, device="cuda:0", outputs=None):
        self.device = dev = torch.device(device)
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev)
        self.x, self.rp = f(w.x), f(w.rp)
        self.x_series = None if w.x_series is None else torch.as_tensor(np.ascontiguousarray(w.x_series), dtype=torch.int32).to(dev)
        self.T, self.B = w.T, w.B
        names = list(_lib.RT_OUT_NAMES) if outputs is None else list(outputs)
        for n in ("S_MINUS", "S_PLUS", "P_MINUS", "P_PLUS"):      # the smoother reads them back
            if n not in names:
                names.append(n)
        self.out = {n: torch.empty((w.T, w.B) if RT_OUT_ROWS[n] == 0 else (w.T, RT_OUT_ROWS[n], w.B), dtype=torch.float64,
                                   device=dev) for n in names}
        self.desc = _lib.RtDesc()
        self.desc.abi_version, self.desc.B, self.desc.T, self.desc.Sx = _lib.ABI_VERSION, w.B, w.T, w.x.shape[1]
        self.desc.L, self.desc.order = int(w.L), int(w.order)
        self.outs = _lib.RtOutputs()
        for n in _lib.RT_OUT_NAMES:
            setattr(self.outs, n, _ptr(self.out.get(n)))
        self.err = C.create_string_buffer(256)
        _lib.check(_lib.lib().epi_rt_expfit_validate(C.byref(self.desc), self.err), self.err)

    def run(self, stream=None):
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        rc = _lib.lib().epi_rt_expfit_run_device(C.byref(self.desc), _ptr(self.x_series), _ptr(self.x), _ptr(self.rp),
                                                 C.byref(self.outs), C.c_void_p(st.cuda_stream), self.err)
        _lib.check(rc, self.err)
        return self.out


def rt_expfit(w, device="cuda:0", outputs=None):
    """Convenience: upload a synth.RtWorkload, run once, return dict name -> numpy array."""
    r = RtRunner(w, device, outputs)
    r.run()
    torch.cuda.synchronize(r.device)
    return {n: t.cpu().numpy() for n, t in r.out.items()}


def seirp_sim(par, init, dt, K, sat=None, integrator="euler", device="cuda:0"):
    """Batched SEIRP / SEIRPSaturatedResource.  par [K or 1, 7, B], init [5, B], sat [6, B] or None.
    Returns torch tensor [K, 5, B] (s,e,i,r,p rows; row 0 is the initial condition, SEIRP.m:20-24)."""
    dev = torch.device(device)
    t = lambda a: None if a is None else (a.contiguous() if isinstance(a, torch.Tensor) else
                                          torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev))
    par, init, sat = t(par), t(init), t(sat)
    B = init.shape[1]
    out = torch.empty((K, 5, B), dtype=torch.float64, device=dev)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_seirp_sim_device(B, K, par.shape[0], float(dt), int(sat is not None),
                                         {"euler": 0, "rk4": 1}[integrator], _ptr(par), _ptr(init), _ptr(sat),
                                         _ptr(out), C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return out


def score_sweep(u_opt_smooth, t_hist, sp, J0_prefix, J1_prefix, store=False, B=None):
    """Scenario scoring tail of the Pareto sweep (Tools/TrainPredictPrescribeNPI.m:481-493) on the device.

    u_opt_smooth : torch [T, n_npi, B] (the smoother's output, left in HBM) or, chain-blocked, [T, nblk, n_npi, blk]
                   (then pass the number of chains B); its last T - t_hist days drive
                   SIalpha_Controlled from the end-of-history state given in `sp` (SIM_* rows, [48, B]);
    J0_prefix/J1_prefix : [B] sequential sums over the t_hist historic days (newcases; weights.*inputs).
    Returns dict with J0, J1 [B] (NPICost over the whole span) and, if store, the simulated s, i, alpha [H, B]."""
    dev = u_opt_smooth.device
    if u_opt_smooth.dim() == 4:
        T, _, n_npi, u_block = u_opt_smooth.shape
        if B is None:
            raise ValueError("blocked u_opt_smooth needs the number of chains B")
    else:
        T, n_npi, B = u_opt_smooth.shape
        u_block = 0
    H = T - t_hist
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev) if not isinstance(a, torch.Tensor) else a
    sp, J0p, J1p = t(sp), t(J0_prefix), t(J1_prefix)
    u_h = u_opt_smooth[t_hist:]                 # contiguous view: [H, n_npi, B]
    d = _lib.SimDesc()
    d.abi_version, d.B, d.K, d.Su, d.n_npi = _lib.ABI_VERSION, B, H, B, n_npi
    d.noise, d.with_cost, d.prefix_days, d.u_block = 0, 1, int(t_hist), int(u_block)
    out = {"J0": torch.empty((B,), dtype=torch.float64, device=dev), "J1": torch.empty((B,), dtype=torch.float64, device=dev)}
    if store:
        for n in ("s", "i", "alpha"):
            out[n] = torch.empty((H, B), dtype=torch.float64, device=dev)
    err = C.create_string_buffer(256)
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().epi_sialpha_score_device(C.byref(d), None, _ptr(u_h), _ptr(sp), None, _ptr(J0p), _ptr(J1p),
                                             _ptr(out.get("s")), _ptr(out.get("i")), _ptr(out.get("alpha")),
                                             _ptr(out["J0"]), _ptr(out["J1"]), C.c_void_p(st.cuda_stream), err)
    _lib.check(rc, err)
    return out
