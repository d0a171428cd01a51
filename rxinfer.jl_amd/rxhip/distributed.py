"""Multi-GPU host logic: one process per GPU, chains sharded, no data-path collective.

The path partitions by independent factor graphs (chains): rank r owns a contiguous block of chains
and runs the unmodified single-GPU engine on it.  The ONLY exchange is the global Bethe free energy
(reference: one scalar per iteration out of `score(model, BetheFreeEnergy, …)`,
src/model/plugins/reactivemp_free_energy.jl:119-123) — an all-reduce(sum) of one double per sweep
(`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests)."""
import numpy as np


def shard_bounds(n_chains, rank, world):
    """Contiguous, balanced partition: the first n_chains % world ranks get one extra chain."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_chains), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_observations(y_chain_major, rank, world):
    """y: [chain][T][dy] (host).  Returns this rank's block."""
    lo, hi = shard_bounds(y_chain_major.shape[0], rank, world)
    return y_chain_major[lo:hi]


def allreduce_free_energy(fe_local, dist=None, device=None):
    """Sum the per-rank batch free energies (per iteration).  fe_local: array-like [iterations] or a
    torch tensor already on the right device.  Deterministic for a fixed world size (ring/tree order is
    fixed by the backend); the result is identical on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return fe_local
    import torch

    if isinstance(fe_local, torch.Tensor):
        dist.all_reduce(fe_local, op=dist.ReduceOp.SUM)
        return fe_local
    t = torch.as_tensor(np.asarray(fe_local, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allgather_ordered_sum_(t, dist, scratch=None):
    """In place: t ← Σ over ranks of t, as an all-gather into [world][n] followed by ONE local reduction over the rank axis.  Every rank reduces the same
    bytes with the same kernel, so the result is bit-identical on every rank and from run to run, whatever ring or tree the backend would pick for an
    all-reduce (whose summation order is the backend's) — what the 1e-8 run-to-run free-energy tolerance needs.  Returns the scratch
    buffer for reuse (allocate once, outside a timed region)."""
    import torch

    world = dist.get_world_size()
    flat = t.view(-1)
    if scratch is None or scratch.numel() != world * flat.numel() or scratch.device != t.device:
        scratch = torch.empty(world * flat.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(scratch, flat)
    torch.sum(scratch.view(world, -1), dim=0, out=flat)
    return scratch


def gather_per_chain(values_local, n_chains, dist=None):
    """All-gather a per-chain array ([local_chains, ...]) into chain order (used by tests / result
    assembly; not on the timed path)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return values_local
    import torch

    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, np.asarray(values_local))
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == n_chains
    return out


def sharded_infer(run_shard, y_chain_major, dist=None):
    """Drive one sharded inference: `run_shard(y_block) -> (mean, cov, fe_per_chain)` is the single-GPU
    engine on this rank's chains.  Returns (mean_block, cov_block, fe_per_chain_block, fe_total_global)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    yb = shard_observations(y_chain_major, rank, world)
    mean, cov, fe = run_shard(yb)
    fe_total = allreduce_free_energy(np.array([np.sum(fe)]), dist)[0]
    return mean, cov, fe, float(fe_total)


# ---- mean-field mixture: the path's one real exchange step ------------------------------------------------
class DeviceMixtureShard:
    """Adapter of a `GMMEngine` holding this rank's points to the split-phase protocol of `sharded_mixture_vmp`.

    If the engine was created on torch's CURRENT stream (`GMMEngine(..., stream=torch.cuda.current_stream().cuda_stream)`)
    the three steps of an iteration — accumulate kernel, RCCL all-reduce, update kernel — are simply enqueued in order on
    that one stream and the host never waits.  Otherwise the host synchronises around the collective."""

    def __init__(self, engine):
        self.engine = engine
        self._stats = None

    def _same_stream(self):
        import torch

        return int(self.engine.stream() or 0) == int(torch.cuda.current_stream().cuda_stream or 0)

    def begin(self, iterations):
        self.engine.begin_run(iterations)
        self._stats = self.engine.statistics_tensor()  # aliases the device buffer (fixed for the engine's lifetime)

    def accumulate(self):
        """Streams the shard; returns a tensor ALIASING the 3K'+1 statistics (all-reduced in place)."""
        self.engine.accumulate()
        if not self._same_stream():
            self.engine.sync()  # the statistics are produced on the engine's stream; the collective must come after them
        return self._stats

    def update(self, want_fe):
        import torch

        if not self._same_stream():
            torch.cuda.current_stream().synchronize()  # the in-place all-reduce (torch's stream) before the update kernel
        self.engine.update(want_fe)


def sharded_mixture_vmp(shard, iterations, want_fe=True, dist=None):
    """VMP for a mixture whose points are sharded over ranks (C5): per iteration every rank accumulates the
    responsibility-weighted statistics of its points (3K+1 numbers: Σπ, Σπy, Σπy² per component and Σ H[q(z_i)]),
    ONE exchange (all-gather + a local reduction over the rank axis: bit-identical everywhere) makes them global, and every rank applies the identical update — so all ranks hold the same
    q(m), q(p), q(s) and the same (global) free energy without any further exchange.
    Reference semantics: the messages toward m[k], p[k], s are products over ALL points
    (src/model/plugins/reactivemp_inference.jl:365-374); the sum over shards is that product in the natural
    parameters.  `shard`: begin(iterations) / accumulate() -> tensor aliasing the statistics / update(want_fe)."""
    shard.begin(iterations)
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    scratch = None
    for _ in range(iterations):
        stats = shard.accumulate()
        if multi:
            scratch = allgather_ordered_sum_(stats, dist, scratch)
        shard.update(want_fe)
