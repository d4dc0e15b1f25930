"""Sample sharding over ranks + gather of per-sample summaries.

Samples are independent, so the data path has NO collective: rank r integrates a contiguous block
of the batch on its own GPU.  The only exchange is the result collection that replaces the
reference's ProcessPoolExecutor fan-in (monte_carlo.py:76-83): one all-gather of the
[16, n/world] summary block and the [n/world] status word per rank (RCCL over xGMI when the
process group is "nccl"; "gloo" in the CPU tests).
"""
import numpy as np
import torch


def world():
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d.get_rank(), d.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`; every rank's block has ceil(n/world) slots, the
    last ones possibly empty, so that the gathered tensor is rectangular."""
    per = -(-n // world_size)
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi, per


def all_gather_summaries(summary, status, n_total, group=None, force_collective=False):
    """summary [S, n_local], status [n_local] of this rank -> ([S, n_total], [n_total]) on every
    rank.  Tensors stay on their device (GPU for nccl/RCCL, CPU for gloo).  A single-rank world returns
    its inputs unless `force_collective` (tests: run the collective itself in a one-rank RCCL group)."""
    rank, ws = world()
    if ws == 1 and not (force_collective and torch.distributed.is_available() and torch.distributed.is_initialized()):
        return summary, status
    d = torch.distributed
    _, _, per = shard_bounds(n_total, rank, ws)
    S = summary.shape[0]
    pad_s = torch.full((S, per), float("nan"), dtype=summary.dtype, device=summary.device)
    pad_t = torch.zeros((per,), dtype=status.dtype, device=status.device)
    pad_s[:, : summary.shape[1]] = summary
    pad_t[: status.shape[0]] = status
    # outputs are the rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
    out_s = torch.empty((ws * S, per), dtype=summary.dtype, device=summary.device)
    out_t = torch.empty((ws * per,), dtype=status.dtype, device=status.device)
    d.all_gather_into_tensor(out_s, pad_s.contiguous(), group=group)
    d.all_gather_into_tensor(out_t, pad_t.contiguous(), group=group)
    full_s = out_s.view(ws, S, per).permute(1, 0, 2).reshape(S, ws * per)[:, :n_total].contiguous()
    full_t = out_t.reshape(ws * per)[:n_total].contiguous()
    return full_s, full_t


def run_local_shard(n_total, local_batch, runner, group=None):
    """`local_batch` holds THIS rank's contiguous shard [lo, hi) of an n_total-sample batch (None or empty
    when the shard is empty): integrate it with `runner` and all-gather.  runner(HostBatch) -> (summary
    [S, m] tensor, status [m] tensor).  Returns NumPy (summary [S, n_total], status [n_total]), identical
    on every rank.  Host preparation per rank is proportional to n_total / world."""
    rank, ws = world()
    err = None
    if local_batch is not None and local_batch.n > 0:
        try:
            summ, stat = runner(local_batch)
        except Exception as e:   # noqa: BLE001 - re-raised below, after the collective every other rank is waiting in
            if ws == 1:
                raise
            err = e
            summ, stat = failed_shard(local_batch.n, group)
    else:
        summ = stat = None
    if ws == 1:
        return summ.cpu().numpy(), stat.cpu().numpy()
    if summ is None:  # empty shard (n < world): contribute padding only
        ref_dev = torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend(group) == "nccl" else torch.device("cpu")
        summ = torch.empty((16, 0), dtype=torch.float64, device=ref_dev)
        stat = torch.empty((0,), dtype=torch.int32, device=ref_dev)
    elif torch.distributed.get_backend(group) != "nccl":   # gloo (CPU tests, single-GPU rehearsals) gathers host tensors
        summ, stat = summ.cpu(), stat.cpu()
    full_s, full_t = all_gather_summaries(summ, stat, n_total, group)
    if err is not None:
        raise err
    return full_s.cpu().numpy(), full_t.cpu().numpy()


def failed_shard(m, group=None):
    """What a rank whose integration raised contributes to the gather (it must still take part, or the
    other ranks wait in the collective for ever): NaN summaries and status words that carry ERPL_ST_INCOMPLETE, which
    every rank refuses to hand on (TrajectoryEngine.raise_if_incomplete); the failing rank re-raises its own error."""
    from . import _abi
    nccl = torch.distributed.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    return (torch.full((_abi.SUMMARY_DIM, m), float("nan"), dtype=torch.float64, device=dev),
            torch.full((m,), _abi.ST_INCOMPLETE, dtype=torch.int32, device=dev))


def run_sharded(host_batch, runner, group=None):
    """Integrate `host_batch` (all samples, identical on every rank) with this rank's `runner` on
    its contiguous shard and gather (see run_local_shard, which avoids building the other ranks' samples)."""
    rank, ws = world()
    lo, hi, _ = shard_bounds(host_batch.n, rank, ws)
    local = host_batch.take(np.arange(lo, hi)) if hi > lo else None
    return run_local_shard(host_batch.n, local, runner, group)
