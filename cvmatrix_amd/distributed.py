"""Multi-GPU: one process per GPU, folds sharded across ranks, RCCL over xGMI for the one
exchange the path has (replicating the full-data Gram and column statistics).

The reference has no distributed code at all; its only parallel structure
is that folds are independent given the full-data matrices (cvmatrix.py:1001-1010 reads
nothing else).  Two ways to get those matrices onto every rank:

  row-sharded  (default) every rank owns a block of rows -- and therefore the folds made of
               those rows -- runs the fit-stage kernel on its own rows only and the partial
               [G | H | gstats] are summed with ONE all-reduce.  Fit work scales 1/world.
  replicated   every rank holds all rows; rank `src` runs the fit stage and broadcasts
               [G | H | gstats]; folds are dealt round-robin (fold f -> rank f mod world).

After that exchange the per-fold update needs no communication.  The collectives below
are written against torch.distributed only, so the same code runs on RCCL ("nccl" backend
on ROCm) and, for the CPU tests, on gloo."""

from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .cvmatrix import CVMatrix


def assign_folds(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Deal folds to ranks: longest-processing-time first on the fold row counts (ties by
    fold number), so ragged folds balance; equal folds reduce to a round-robin deal.
    Returns, per rank, the fold numbers it owns in ascending order."""
    sizes = np.asarray(sizes, dtype=np.int64)
    order = sorted(range(len(sizes)), key=lambda f: (-int(sizes[f]), f))
    load = [0] * world
    owned: List[List[int]] = [[] for _ in range(world)]
    for f in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owned[r].append(f)
        load[r] += int(sizes[f]) if sizes[f] > 0 else 1
    return [sorted(o) for o in owned]


def shard_folds(labels, world: int, rank: int):
    """Strong-scaling layout of ONE cross-validation over ``world`` GPUs (the
    reference's parallel shape is a vmap over folds, benchmarks/benchmark.py:144-152): fold
    f -> rank by ``assign_folds`` on the fold sizes; a rank holds exactly the rows its folds
    are made of.  ``labels``: one fold label per row of the whole data set.

    Returns ``(keys, rows, local_labels)``: the fold labels this rank owns (in the
    Partitioner's first-seen order), the global row numbers it holds (ascending) and those
    rows' fold labels -- ``Partitioner(local_labels)`` then yields this rank's folds in LOCAL
    row numbers, in the order of ``keys``."""
    from .partitioner import Partitioner

    p = Partitioner(labels)
    all_keys = list(p.folds_dict)
    sizes = [p.folds_dict[k].size for k in all_keys]
    mine = assign_folds(sizes, world)[rank]
    keys = [all_keys[f] for f in mine]
    rows = (np.sort(np.concatenate([p.folds_dict[k] for k in keys])) if keys
            else np.zeros(0, dtype=np.int64))
    lab = np.asarray(labels)
    local_labels = lab[rows] if lab.dtype != object else np.array([labels[i] for i in rows], dtype=object)
    return keys, rows, local_labels


def pack_globals(G: torch.Tensor, H: Optional[torch.Tensor], gstats: torch.Tensor) -> torch.Tensor:
    """One float64 buffer [G | H | gstats] so the exchange is a single collective
    (K(K+M)+2(K+M)+2 values: 2.2 MB at K=512,M=16; 67 MB at K=4096)."""
    parts = [G.reshape(-1).double()]
    if H is not None:
        parts.append(H.reshape(-1).double())
    parts.append(gstats.reshape(-1).double())
    return torch.cat(parts)


def unpack_globals(buf: torch.Tensor, G: torch.Tensor, H: Optional[torch.Tensor],
                   gstats: torch.Tensor) -> None:
    o = 0
    for t in (G, H, gstats):
        if t is None:
            continue
        n = t.numel()
        t.copy_(buf[o : o + n].reshape(t.shape).to(t.dtype))
        o += n


def allreduce_globals(G, H, gstats, group=None, flat: Optional[torch.Tensor] = None) -> None:
    """Sum the per-rank partial [G | H | gstats] in place on every rank.  ``flat``: the
    contiguous buffer the three tensors are views of (CVMatrix allocates them that way for
    float64): the collective then runs on it directly, nothing is packed or copied back."""
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return
    buf = pack_globals(G, H, gstats)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    unpack_globals(buf, G, H, gstats)


def broadcast_globals(G, H, gstats, src: int = 0, group=None,
                      flat: Optional[torch.Tensor] = None) -> None:
    """Replicate rank `src`'s [G | H | gstats] on every rank."""
    if flat is not None:
        dist.broadcast(flat, src=src, group=group)
        return
    buf = pack_globals(G, H, gstats)
    dist.broadcast(buf, src=src, group=group)
    unpack_globals(buf, G, H, gstats)


class ShardedCVMatrix(CVMatrix):
    """CVMatrix over the GPUs of one node, one process per GPU.

    mode="row_sharded": ``fit`` receives THIS RANK'S rows; fold indices passed to
    ``training_*`` are local row numbers of this rank.  mode="replicated": ``fit`` receives
    all rows on every rank; use ``my_folds`` to pick this rank's share of the folds.

    The path has ONE exchange per fit -- the all-reduce (row-sharded) or broadcast (replicated) of
    ``[G | H | gstats]`` -- and every rank must enter it:

    * default (``lazy_fit=False``): the exchange happens inside ``fit``; every rank calls ``fit``
      (a rank that owns no rows passes zero-row arrays: it contributes zeros, runs no kernel and
      still takes part);
    * ``lazy_fit=True``: the exchange moves to the first call that needs the full-data matrices
      (``training_*``, ``training_*_batched``, ``XTX`` ...).  EVERY rank must then make such a
      call after every ``fit`` -- a rank without folds calls ``ensure_fit()`` -- or the others
      wait in the collective for ever.  Nothing can detect a rank that does not call (that is what
      a collective is); this mode is for loops like bench.py's, where all ranks run the same
      program."""

    def __init__(self, *args, mode: str = "row_sharded", group=None, src: int = 0, lazy_fit: bool = False,
                 **kw):
        # lazy_fit=True moves the exchange of the full-data matrices from ``fit`` to the first
        # call that needs them: every rank must then make that call (it is collective)
        super().__init__(*args, lazy_fit=lazy_fit, **kw)
        if mode not in ("row_sharded", "replicated"):
            raise ValueError("mode must be 'row_sharded' or 'replicated'")
        self.mode, self.group, self.src = mode, group, src
        self._tail_host = self._tail_event = None
        self._tail_pending = self._tail_requested = False
        self._probe = None      # bench.py: callable(label) at "exchange_begin" / "exchange_end"

    def ensure_fit(self) -> None:
        """Take part in a pending lazy exchange (a rank that has no fold to ask for)."""
        self._ensure_fit()

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def my_folds(self, sizes: Sequence[int]) -> List[int]:
        return assign_folds(sizes, self.world)[self.rank]

    def fit(self, X, Y=None, weights=None, folds=None, assume_unchanged=None) -> None:
        if self.world > 1 and self.mode == "replicated" and folds is not None:
            raise ValueError("fit(folds=...) needs mode='row_sharded' (or a single process)")
        self._tail_pending = False
        super().fit(X, Y, weights, folds=folds, assume_unchanged=assume_unchanged)     # row_sharded: this rank's rows (and folds)

    def _launch_fit(self, lib) -> None:
        if self.world > 1 and self.mode == "replicated" and self.rank != self.src:
            self._neg = torch.zeros(1, dtype=torch.int32, device=self.device)
            return                                  # the broadcast fills the matrices
        super()._launch_fit(lib)

    def _lazy_sweep(self, batch) -> None:
        if self.world > 1 and self.mode == "replicated":
            return self._ensure_fit()               # a rank's folds do not partition the rows
        super()._lazy_sweep(batch)

    def _after_globals(self) -> None:
        """The one exchange of the path: sum (row-sharded) or replicate the full-data matrices.
        Every rank issues exactly this one collective per fit, whatever its local data look
        like (no collective may depend on rank-local state)."""
        if self.world == 1:
            return
        probe = self._probe
        if probe is not None:
            probe("exchange_begin")
        self._exchange()
        if probe is not None:
            probe("exchange_end")
        if self.mode == "row_sharded":
            # the global sample / non-zero-weight counts for the host-side validity checks
            # (cvmatrix.py:612-630, 1074-1078) ride in the all-reduced statistics vector
            # ([... | sw | nz]: sw = N when unweighted).  They stay on the device until a check
            # cannot be decided on this rank's own counts (CVMatrix._passes_on_local_counts):
            # then ``_request_totals`` copies them into pinned memory asynchronously and the check
            # awaits that copy -- a step whose folds leave this rank enough rows never waits
            self._tail_pending, self._tail_requested = True, False

    def _request_totals(self) -> None:
        if not self._tail_pending or self._tail_requested:
            return
        K, M = self._Kd, self._Md or 0
        if self._tail_host is None:
            self._tail_host = torch.empty(2, dtype=torch.float64, pin_memory=True)
            self._tail_event = torch.cuda.Event()
        with torch.cuda.device(self.device):
            self._tail_host.copy_(self._gs[2 * K + 2 * M: 2 * K + 2 * M + 2], non_blocking=True)
            self._tail_event.record()
        self._tail_requested = True

    def _exchange(self) -> None:
        """The collective itself (on the current stream's NCCL/RCCL ordering)."""
        if self.mode == "row_sharded":
            allreduce_globals(self._G, self._H, self._gs, self.group, flat=self._globals)
        else:
            broadcast_globals(self._G, self._H, self._gs, self.src, self.group, flat=self._globals)

    def _exchanges_globals(self) -> bool:
        return self.world > 1

    def _totals_in_flight(self) -> bool:
        return bool(self._tail_pending) or super()._totals_in_flight()

    def _resolve_totals(self) -> None:
        super()._resolve_totals()               # (this rank's own weights: a deferred check raises here)
        if not self._tail_pending:
            return
        self._request_totals()
        self._tail_pending = False
        self._tail_event.synchronize()
        sw, nz = float(self._tail_host[0]), float(self._tail_host[1])
        self._nz_total = int(round(nz))
        # (weighted: the sample count is never consulted -- cvmatrix.py:612-615 is the
        #  unweighted branch)
        self._n_total = int(round(sw)) if self.weights is None else None
        self._sum_w = None
