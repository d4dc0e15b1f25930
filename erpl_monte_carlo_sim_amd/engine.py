"""TrajectoryEngine — thin Python driver of the C ABI (include/erpl_mc.h).

PyTorch-ROCm is used only as plumbing: it owns the device buffers (SoA tensors) and the HIP
stream; every number is produced by the hand-written HIP kernels behind `erpl_mc_run_batch`.
No CPU execution path exists here: if the HIP library or a GPU is missing, construction raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi


class DeviceBatch:
    """Per-sample SoA tensors resident in HBM (the layout `erpl_batch` points at)."""

    def __init__(self, ic, rocket, motor, alt_grid, wind, precision):
        self.ic, self.rocket, self.motor = ic, rocket, motor
        self.alt_grid, self.wind = alt_grid, wind
        self.precision = precision
        self.n = int(ic.shape[1])
        self.k_wind = 0 if wind is None else int(wind.shape[0])
        # the kernels index these by (row, sample) and the knots by row: a wrong shape is an out-of-bounds read on the GPU
        if wind is not None:
            if alt_grid is None or tuple(wind.shape) != (self.k_wind, 3, self.n) or alt_grid.numel() != self.k_wind:
                raise ValueError(f"wind must be (K, 3, n) with K altitudes: got wind {tuple(wind.shape)}, "
                                 f"altitudes {None if alt_grid is None else tuple(alt_grid.shape)}, n = {self.n}")
            if not 1 <= self.k_wind <= _abi.MAX_WIND_KNOTS:
                raise ValueError(f"1..{_abi.MAX_WIND_KNOTS} wind knots, got {self.k_wind}")
            if alt_grid.dtype != torch.float64 or wind.dtype != (torch.float32 if precision == _abi.PREC_F32 else torch.float64):
                raise ValueError("altitudes are float64; the wind table is float32 for the fp32 build, float64 otherwise")
            if not (wind.is_contiguous() and alt_grid.is_contiguous()):
                raise ValueError("wind and altitudes must be contiguous")

    def input_bytes(self):
        b = self.ic.numel() * 8 + self.rocket.numel() * 8 + self.motor.numel() * 8
        if self.wind is not None:
            b += self.wind.numel() * self.wind.element_size() + self.alt_grid.numel() * 8
        return b

    @staticmethod
    def from_host(hb, device, precision=_abi.PREC_F64):
        """Upload a flatten.HostBatch.  The wind table is stored in the working precision."""
        if hb.k_wind > _abi.MAX_WIND_KNOTS:
            raise _abi.ErplError(f"{hb.k_wind} wind knots exceed the ABI limit {_abi.MAX_WIND_KNOTS}")
        if hb.k_wind and not np.all(np.diff(hb.alt_grid) > 0):
            raise _abi.ErplError("altitude_profile must be strictly increasing")
        # (one reduction instead of an element-wise mask: a NaN or inf anywhere makes the sum non-finite; a finite table
        # whose sum overflows is re-checked element by element)
        if hb.k_wind and not (np.all(np.isfinite(hb.alt_grid)) and (np.isfinite(np.sum(hb.wind)) or np.all(np.isfinite(hb.wind)))):
            raise _abi.ErplError("wind profile must be finite")
        # The kernels assume what the reference silently assumes: finite inputs, positive masses and
        # mass flow, a finite burn time (a non-finite burn time would never leave the launch rail).
        if hb.n >= 2 ** 31:
            raise _abi.ErplError("at most 2**31 - 1 samples per batch")
        if not (np.all(np.isfinite(hb.ic)) and np.all(np.isfinite(hb.rocket)) and np.all(np.isfinite(hb.motor))):
            raise _abi.ErplError("initial conditions, masses and motor parameters must be finite")
        if not (np.all(hb.rocket[0] > 0) and np.all(hb.rocket[1] > 0)):
            raise _abi.ErplError("dry_mass and propellant_mass must be positive")
        if not (np.all(hb.motor[2] > 0) and np.all(hb.motor[3] >= 0)):
            raise _abi.ErplError("mass_flow_rate must be positive and burn_time non-negative")
        wdt = torch.float32 if precision == _abi.PREC_F32 else torch.float64
        f64 = dict(dtype=torch.float64, device=device)
        ic = torch.as_tensor(np.ascontiguousarray(hb.ic), **f64)
        rocket = torch.as_tensor(np.ascontiguousarray(hb.rocket), **f64)
        motor = torch.as_tensor(np.ascontiguousarray(hb.motor), **f64)
        if hb.k_wind:
            alt = torch.as_tensor(np.ascontiguousarray(hb.alt_grid), **f64)
            wind = torch.as_tensor(np.ascontiguousarray(hb.wind), device=device).to(wdt).contiguous()
        else:
            alt, wind = None, None
        return DeviceBatch(ic, rocket, motor, alt, wind, precision)


class TrajectoryEngine:
    """One engine (= one `erpl_ctx`) per GPU / torch.distributed rank."""

    def __init__(self, device=None, lib_path=None):
        self.lib = _abi.load_library(lib_path)   # lib_path: an experiment build (tools/), never the product
        if not torch.cuda.is_available():
            raise _abi.ErplError("no GPU visible to PyTorch-ROCm; this engine has no CPU fallback")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._ctx = C.c_void_p()
        _abi.check(self.lib, self.lib.erpl_mc_create(idx, C.byref(self._ctx)), "erpl_mc_create")
        self._cfg = None

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.erpl_mc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_config(self, cfg):
        _abi.check(self.lib, self.lib.erpl_mc_set_config(self._ctx, C.byref(cfg)), "erpl_mc_set_config")
        self._cfg = cfg

    def set_launch(self, block_threads=256, max_blocks=0, refill_threshold=1):
        _abi.check(self.lib, self.lib.erpl_mc_set_launch(self._ctx, block_threads, max_blocks, refill_threshold),
                   "erpl_mc_set_launch")

    def set_waves_per_simd(self, waves):
        """fp32 flight-kernel build: 2 (256 VGPRs), 3 (168 VGPRs, three resident waves), 0 = by batch size."""
        _abi.check(self.lib, self.lib.erpl_mc_set_waves_per_simd(self._ctx, int(waves)), "erpl_mc_set_waves_per_simd")

    def set_chunk(self, chunk_steps):
        """Step-chunked launches with per-GPU compaction in between (0 = single launch; < 0 = the library decides
        per batch from the trajectory lengths of the batches it has finished: erpl_mc_set_chunk)."""
        _abi.check(self.lib, self.lib.erpl_mc_set_chunk(self._ctx, int(chunk_steps)), "erpl_mc_set_chunk")

    def set_adopt(self, lanes):
        """Lane adoption: waves down to `lanes` flying trajectories hand them to fuller waves (0 = off, < 0 = the
        library decides per batch: erpl_mc_set_adopt)."""
        _abi.check(self.lib, self.lib.erpl_mc_set_adopt(self._ctx, int(lanes)), "erpl_mc_set_adopt")

    def reserve(self, n):
        _abi.check(self.lib, self.lib.erpl_mc_reserve(self._ctx, n), "erpl_mc_reserve")

    def alloc_outputs(self, n):
        summary = torch.empty((_abi.SUMMARY_DIM, n), dtype=torch.float64, device=self.device)
        status = torch.empty((n,), dtype=torch.int32, device=self.device)
        return summary, status

    def set_short_flight_overlap(self, depth):
        """How many batches of short flights start side by side (erpl_mc_set_short_flight_overlap; default 4, 0 = no limit
        besides set_overlap).  Scheduling only."""
        _abi.check(self.lib, self.lib.erpl_mc_set_short_flight_overlap(self._ctx, int(depth)), "erpl_mc_set_short_flight_overlap")

    def get_overlap(self):
        """Batches `submit()` keeps in flight at once (3, or 8 when the process has the hardware queues for it)."""
        return int(self.lib.erpl_mc_get_overlap(self._ctx))

    def set_overlap(self, depth):
        """Batches `submit()` keeps in flight at once (erpl_mc_set_overlap; 1..8)."""
        _abi.check(self.lib, self.lib.erpl_mc_set_overlap(self._ctx, int(depth)), "erpl_mc_set_overlap")

    def submit(self, db, **kw):
        """Like run(), but on one of the library's internal streams (erpl_mc_submit_batch): the batch
        starts after everything enqueued on the current stream so far and may overlap earlier batches;
        the current stream does NOT wait for it.  Call wait() before touching the outputs.  The ticket
        of the batch is kept in `self.last_ticket`."""
        return self.run(db, overlap=True, **kw)

    def wait(self, ticket=-1, stream=None):
        """Make `stream` (default: the current torch stream) wait on the device for a submitted batch
        (ticket < 0: for all of them).  The host does not block."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        _abi.check(self.lib, self.lib.erpl_mc_wait_batch(self._ctx, int(ticket), C.c_void_p(st.cuda_stream)),
                   "erpl_mc_wait_batch")

    def synchronize(self):
        """Host-blocking wait for everything enqueued; raises _abi.IncompleteBatch if a lane hand-over timed out."""
        _abi.check(self.lib, self.lib.erpl_mc_synchronize(self._ctx), "erpl_mc_synchronize")

    def check(self, ticket=-1):
        """Where results are consumed: block the host until batch `ticket` (< 0: every submitted batch) has finished
        and raise _abi.IncompleteBatch if one of its lane hand-overs timed out (erpl_mc_check_batch).  wait() only
        orders a stream on the device; it reports batches that had ALREADY finished incomplete when it is called."""
        _abi.check(self.lib, self.lib.erpl_mc_check_batch(self._ctx, int(ticket)), "erpl_mc_check_batch")

    def set_adopt_spin(self, polls):
        """Test knob (erpl_mc_set_adopt_spin): < 0 makes every adopting lane give up at once."""
        _abi.check(self.lib, self.lib.erpl_mc_set_adopt_spin(self._ctx, int(polls)), "erpl_mc_set_adopt_spin")

    @staticmethod
    def raise_if_incomplete(status):
        """Results whose status words still carry ST_INCOMPLETE were never integrated: refuse to hand them on."""
        bad = (status & _abi.ST_INCOMPLETE) != 0
        n_bad = int(bad.sum())
        if n_bad:
            raise _abi.IncompleteBatch(f"{n_bad} sample(s) carry ERPL_ST_INCOMPLETE: a lane hand-over timed out")

    def debug_eval(self, db, what, inputs):
        """Device-side known-answer evaluation (erpl_mc_debug_eval): inputs [rows, m] -> outputs [rows', m]
        float64, through the device functions of the kernel build `db.precision` selects."""
        x = torch.as_tensor(np.ascontiguousarray(np.atleast_2d(inputs), dtype=np.float64), device=self.device)
        m = int(x.shape[1])
        rows = {_abi.DBG_ATMOSPHERE: 4, _abi.DBG_AERO: 5, _abi.DBG_RHS: 15}[what]
        out = torch.full((rows, m), float("nan"), dtype=torch.float64, device=self.device)
        b = _abi.ErplBatch()
        b.n, b.precision, b.k_wind, b.flags = db.n, db.precision, db.k_wind, 0
        b.ic, b.rocket, b.motor = db.ic.data_ptr(), db.rocket.data_ptr(), db.motor.data_ptr()
        b.alt_grid = db.alt_grid.data_ptr() if db.k_wind else None
        b.wind = db.wind.data_ptr() if db.k_wind else None
        st = torch.cuda.current_stream(self.device)
        rc = self.lib.erpl_mc_debug_eval(self._ctx, C.byref(b), int(what), m, C.c_void_p(x.data_ptr()),
                                         C.c_void_p(out.data_ptr()), C.c_void_p(st.cuda_stream))
        _abi.check(self.lib, rc, "erpl_mc_debug_eval")
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()

    def run(self, db, flags=0, summary=None, status=None, traj_ids=None, traj_stride=1, traj_cap=0,
            stream=None, overlap=False):
        """Enqueue rail + flight kernels for the batch on the current torch stream.
        Returns (summary [16, n] f64, status [n] i32[, traj [m, cap, 15] f64, traj_len [m] i64]);
        asynchronous with respect to the host."""
        if summary is None or status is None:
            summary, status = self.alloc_outputs(db.n)
        b = _abi.ErplBatch()
        b.n, b.precision, b.k_wind, b.flags = db.n, db.precision, db.k_wind, flags
        b.ic, b.rocket, b.motor = db.ic.data_ptr(), db.rocket.data_ptr(), db.motor.data_ptr()
        b.alt_grid = db.alt_grid.data_ptr() if db.k_wind else None
        b.wind = db.wind.data_ptr() if db.k_wind else None
        o = _abi.ErplOut()
        o.summary, o.status = summary.data_ptr(), status.data_ptr()
        traj = tlen = ids = None
        if traj_ids is not None and len(traj_ids):
            ids = torch.as_tensor(np.asarray(traj_ids, dtype=np.int64), device=self.device)
            traj = torch.full((len(ids), traj_cap, _abi.TRAJ_DIM), float("nan"), dtype=torch.float64,
                              device=self.device)
            tlen = torch.zeros((len(ids),), dtype=torch.int64, device=self.device)
            o.n_traj, o.traj_ids, o.traj_stride, o.traj_cap = len(ids), ids.data_ptr(), traj_stride, traj_cap
            o.traj, o.traj_len = traj.data_ptr(), tlen.data_ptr()
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if overlap:
            t = C.c_int64(0)
            rc = self.lib.erpl_mc_submit_batch(self._ctx, C.byref(b), C.byref(o), C.c_void_p(st.cuda_stream), C.byref(t))
            _abi.check(self.lib, rc, "erpl_mc_submit_batch")
            self.last_ticket = t.value
        else:
            rc = self.lib.erpl_mc_run_batch(self._ctx, C.byref(b), C.byref(o), C.c_void_p(st.cuda_stream))
            _abi.check(self.lib, rc, "erpl_mc_run_batch")
        if traj is not None:
            self._keep = ids
            return summary, status, traj, tlen
        return summary, status

    def extract_histories(self, db, sample, traj, time_offset, stream=None):
        """Per-step diagnostic histories (simulator.py:496-552) of the records `traj` [m, 15] (device
        tensor, as produced by run(..., traj_ids=...)) of sample `sample`: returns [m, 17] f64."""
        traj = traj.contiguous()
        m = int(traj.shape[0])
        out = torch.empty((m, _abi.DIAG_DIM), dtype=torch.float64, device=self.device)
        b = _abi.ErplBatch()
        b.n, b.precision, b.k_wind, b.flags = db.n, db.precision, db.k_wind, 0
        b.ic, b.rocket, b.motor = db.ic.data_ptr(), db.rocket.data_ptr(), db.motor.data_ptr()
        b.alt_grid = db.alt_grid.data_ptr() if db.k_wind else None
        b.wind = db.wind.data_ptr() if db.k_wind else None
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        rc = self.lib.erpl_mc_extract_histories(self._ctx, C.byref(b), int(sample), C.c_void_p(traj.data_ptr()), m,
                                                float(time_offset), C.c_void_p(out.data_ptr()),
                                                C.c_void_p(st.cuda_stream))
        _abi.check(self.lib, rc, "erpl_mc_extract_histories")
        return out

    def set_profiling(self, enable=True):
        """Record HIP events around the two kernels on the launch stream (erpl_mc_set_profiling)."""
        _abi.check(self.lib, self.lib.erpl_mc_set_profiling(self._ctx, int(bool(enable))), "erpl_mc_set_profiling")

    def last_kernel_ms(self):
        """(rail_ms, flight_ms) device durations of the last profiled run (synchronises on its end)."""
        a, b = C.c_float(), C.c_float()
        _abi.check(self.lib, self.lib.erpl_mc_last_kernel_ms(self._ctx, C.byref(a), C.byref(b)),
                   "erpl_mc_last_kernel_ms")
        return a.value, b.value

    def kernel_ms_history(self, max_runs=_abi.PROFILE_RING):
        """Per-launch (rail_ms[], flight_ms[]) of the most recent profiled runs, oldest first."""
        m = min(int(max_runs), _abi.PROFILE_RING)
        ra, fa, n = (C.c_float * m)(), (C.c_float * m)(), C.c_int(0)
        _abi.check(self.lib, self.lib.erpl_mc_kernel_ms_history(self._ctx, m, ra, fa, C.byref(n)),
                   "erpl_mc_kernel_ms_history")
        return list(ra[:n.value]), list(fa[:n.value])

    def debug_counters(self):
        out = (C.c_double * 16)()
        torch.cuda.synchronize(self.device)
        _abi.check(self.lib, self.lib.erpl_mc_debug_counters(self._ctx, out), "erpl_mc_debug_counters")
        return list(out)

    def last_stats(self):
        """(physics RK4 steps integrated, wave-iterations) of the last run (synchronises)."""
        a, b = C.c_double(), C.c_double()
        torch.cuda.synchronize(self.device)
        _abi.check(self.lib, self.lib.erpl_mc_last_stats(self._ctx, C.byref(a), C.byref(b)), "erpl_mc_last_stats")
        return a.value, b.value

    def ticket_stats(self, ticket):
        """(physics RK4 steps integrated, wave-iterations) of ONE submitted batch (waits for that batch alone)."""
        a, b = C.c_double(), C.c_double()
        _abi.check(self.lib, self.lib.erpl_mc_ticket_stats(self._ctx, C.c_int64(ticket), C.byref(a), C.byref(b)), "erpl_mc_ticket_stats")
        return a.value, b.value
