"""One rank of a multi-GPU job on ONE GPU, without a process group (a measurement tool).

``EmulatedRank`` is a ``ShardedCVMatrix`` that reports the world size and rank it is told to and
runs every code path of the real multi-GPU step; the one collective of the path
(``ShardedCVMatrix._exchange``) is replaced by its last arithmetic step -- an in-place add
(row-sharded) or copy (replicated) of a buffer of the same size that holds the OTHER ranks' share
of ``[G | H | gstats]``, computed once up front with ``others_share`` so that the results are the
real job's (to rounding) -- followed by ``comm_us`` microseconds of held stream.  ``bench.py
--emulate-world G`` and tools/emulate_scaling.py predict the strong-scaling curve with it."""

from __future__ import annotations

import torch

from cvmatrix_amd.cvmatrix import CVMatrix
from cvmatrix_amd.distributed import ShardedCVMatrix


class EmulatedRank(ShardedCVMatrix):
    def __init__(self, *args, emu_world: int, emu_rank: int, others=None, sleep_cycles: int = 0, **kw):
        super().__init__(*args, **kw)
        self._emu_world, self._emu_rank = int(emu_world), int(emu_rank)
        self.others = others                # (flat | None, G, H, gstats) of the other ranks
        self.sleep_cycles = int(sleep_cycles)

    @property
    def world(self) -> int:
        return self._emu_world

    @property
    def rank(self) -> int:
        return self._emu_rank

    def _exchange(self) -> None:
        o = self.others
        add = self.mode == "row_sharded"
        if self._globals is not None:
            self._globals.add_(o[0]) if add else self._globals.copy_(o[0])
        else:
            for t, u in zip((self._G, self._H, self._gs), o[1:]):
                if t is not None:
                    t.add_(u) if add else t.copy_(u)
        if self.sleep_cycles:
            torch.cuda._sleep(self.sleep_cycles)


def others_share(flags, dtype, dev, mode, whole_xyw, mine_xyw, ddof=1):
    """``[G | H | gstats]`` of everybody but this rank: (whole problem) - (this rank's rows) for the
    row-sharded layout, the whole problem's (what rank 0 broadcasts) for the replicated one."""
    whole = CVMatrix(*flags, ddof=ddof, dtype=dtype, copy=False, device=dev, lazy_fit=False)
    whole.fit(*whole_xyw)
    names = ("_globals", "_G", "_H", "_gs")
    if mode == "row_sharded" and mine_xyw[0].shape[0]:
        mine = CVMatrix(*flags, ddof=ddof, dtype=dtype, copy=False, device=dev, lazy_fit=False)
        mine.fit(*mine_xyw)
        return [None if getattr(whole, n) is None else getattr(whole, n) - getattr(mine, n) for n in names]
    return [None if getattr(whole, n) is None else getattr(whole, n).clone() for n in names]


def sleep_cycles_for(comm_us: float) -> int:
    """``torch.cuda._sleep`` argument that holds a stream for ``comm_us`` microseconds (calibrated)."""
    if comm_us <= 0:
        return 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(10000000)
    e1.record()
    torch.cuda.synchronize()
    return int(comm_us * 1e-3 / e0.elapsed_time(e1) * 10000000)
