"""Partitioner: fold label -> validation index array.

Host-side mirror of the reference's ``Partitioner`` (cvmatrix/partitioner.py:22-107):
same constructor, ``folds_dict`` (keys in first-seen order, ascending int index arrays)
and ``get_validation_indices`` incl. its ``ValueError("Fold ... not found.")``.  Added for
the batched device path: ``csr()`` exports all folds as one concatenated index array plus
offsets, the layout ``cvm_fold_update`` (include/cvmhip.h) consumes."""

from __future__ import annotations

import threading
import weakref
from collections.abc import Hashable
from typing import Iterable, Optional, Tuple

import numpy as np
import numpy.typing as npt

# id(index array) -> the Partitioner that owns it (entries vanish with the Partitioner).  It lets
# ``CVMatrix.training_*(p.get_validation_indices(fold))`` -- the reference's call pattern,
# README.md:120-141 -- recognise that the indices it was handed are one fold of a known partition
# of the rows, and serve all of that partition's folds from one sweep over the data.
_OWNER: "weakref.WeakValueDictionary[int, Partitioner]" = weakref.WeakValueDictionary()
_REGISTER_MAX_FOLDS = 4096


# id(base array) -> Partitioner, for folds that are views of one array the Partitioner made (integer
# labels: every fold is a row of a matrix or a slice of the sorted order): any number of folds at
# one registry entry; the fold's position follows from its address.
_BASE_OWNER: "weakref.WeakValueDictionary[int, Partitioner]" = weakref.WeakValueDictionary()
# both registries are process-wide and Partitioners may be built and looked up from several threads
# (one model per thread and stream): every access goes through this lock
_REG_LOCK = threading.Lock()


def partitioner_pos(indices):
    """(Partitioner, position of the fold in its ``folds_dict`` order) if ``indices`` is the very
    array object a live Partitioner holds for one of its folds, else (None, None)."""
    with _REG_LOCK:
        p = _OWNER.get(id(indices))
    if p is not None:
        pos = p._fold_pos.get(id(indices))
        if pos is not None and p._fold_arrays[pos] is indices:
            return p, pos
        return None, None
    b = getattr(indices, "base", None)
    if b is None:
        return None, None
    with _REG_LOCK:
        p = _BASE_OWNER.get(id(b))
    if p is None or p._base is not b:
        return None, None
    pos = p._pos_from_address(indices)
    if pos is None or p._fold_arrays[pos] is not indices:
        return None, None
    return p, pos


def partitioner_of(indices) -> "Optional[Partitioner]":
    """The live ``Partitioner`` whose ``folds_dict`` holds exactly this array object, if any."""
    return partitioner_pos(indices)[0]


class Partitioner:
    """Groups sample positions by fold label (Algorithm 1 of Engstrøm & Jensen, as in
    cvmatrix/partitioner.py:89-107).

    Parameters
    ----------
    folds : Iterable of Hashable with N elements
        One label per sample; equal labels form a fold.
    """

    def __init__(self, folds: Iterable[Hashable]) -> None:
        self.folds_dict: dict[Hashable, npt.NDArray[np.int_]] = {}
        self._init_folds_dict(folds)
        self._fold_arrays = list(self.folds_dict.values())
        self._fold_pos: dict = {}
        # folds that are views of one array of this object (integer labels): found by address,
        # whatever their number
        self._base = None
        self._addr_pos = None
        self._starts = getattr(self, "_starts", None)   # ragged folds: their starts inside the sorted order
        if self._fold_arrays:
            b = getattr(self._fold_arrays[0], "base", None)
            if b is not None and all(getattr(a, "base", None) is b for a in self._fold_arrays[:3]):
                self._base = b
                with _REG_LOCK:
                    _BASE_OWNER[id(b)] = self
        if self._base is None and len(self._fold_arrays) <= _REGISTER_MAX_FOLDS:
            # folds that are arrays of their own (labels of any hashable kind): one entry each
            with _REG_LOCK:
                for i, a in enumerate(self._fold_arrays):
                    self._fold_pos[id(a)] = i
                    _OWNER[id(a)] = self

    def _pos_from_address(self, a) -> Optional[int]:
        b = self._base
        off = a.__array_interface__["data"][0] - b.__array_interface__["data"][0]
        if self._addr_pos is None:      # (built on first use; two threads may both build it: same dict)
            sizes = np.fromiter((f.size for f in self._fold_arrays), dtype=np.int64, count=len(self._fold_arrays))
            if self._starts is None:    # rows of a matrix, in order
                st = np.zeros(sizes.size, dtype=np.int64)
                np.cumsum(sizes[:-1], out=st[1:])
            else:
                st = self._starts
            self._addr_pos = dict(zip((st * b.itemsize).tolist(), range(sizes.size)))
        return self._addr_pos.get(off)

    def get_validation_indices(self, fold: Hashable) -> npt.NDArray[np.int_]:
        """Index array of the samples labelled ``fold`` (partitioner.py:61-87)."""
        try:
            return self.folds_dict[fold]
        except KeyError as e:
            raise ValueError(f"Fold {fold} not found.") from e

    def _init_folds_dict(self, folds: Iterable[Hashable]) -> None:
        arr = folds if isinstance(folds, np.ndarray) else None
        if arr is not None and arr.ndim == 1 and arr.dtype.kind in "iub" and arr.size > 0:
            # vectorised grouping for the common integer-label case (N up to 1e6+):
            # a stable sort keeps positions ascending inside each fold; folds are then
            # ordered by first appearance like the reference's dict insertion order.
            # (labels that fit 16 bits -- any realistic number of folds -- sort by radix: 0.7 ms
            #  instead of 2.3 ms for 100 000 rows; the order of a stable sort is the same)
            if self._init_periodic(arr):
                return
            lo, hi = int(arr.min()), int(arr.max())
            key = arr.astype(np.uint16) if (0 <= lo and hi < 65536 and arr.dtype.itemsize > 2) else arr
            order = np.argsort(key, kind="stable")
            sorted_labels = arr[order]
            starts = np.flatnonzero(np.r_[True, sorted_labels[1:] != sorted_labels[:-1]])
            first_pos = order[starts]
            by_first = np.argsort(first_pos, kind="stable")
            keys = arr[first_pos[by_first]]           # iterating yields the labels as NumPy scalars
            P = starts.size
            order = order.astype(int, copy=False)
            if arr.size % P == 0 and (P == 1 or bool(np.all(np.diff(starts) == arr.size // P))):
                # equal folds (arange(N) % P, leave-one-out): rows of one matrix, no Python slicing
                vals = list(order.reshape(P, arr.size // P)[by_first])
            else:
                bounds = np.r_[starts, arr.size]
                self._starts = np.asarray(bounds[by_first], dtype=np.int64)
                lo, hi = bounds[by_first].tolist(), bounds[by_first + 1].tolist()
                vals = [order[a:b] for a, b in zip(lo, hi)]
            out = dict(zip(keys, vals))
            self.folds_dict = out
            return
        buckets: dict = {}
        for pos, label in enumerate(folds):
            buckets.setdefault(label, []).append(pos)
        self.folds_dict = {k: np.asarray(v, dtype=int) for k, v in buckets.items()}

    def _init_periodic(self, arr: np.ndarray) -> bool:
        """Labels that repeat with a period of P distinct values -- ``arange(N) % P``, the folds of the
        reference's benchmark (benchmarks/benchmark.py:233) -- or that count the rows (leave-one-out):
        checked with two vector comparisons and laid out by formula, no sort (N = 1e5, P = 10: 0.15 ms
        against 0.45 ms).  The same dict as the general path builds: keys in first-seen order, ascending
        int index arrays, all views of one array.  False: not such labels, nothing done."""
        n = arr.size
        head = arr[1:min(n, 4097)]
        rep = np.flatnonzero(head == arr[0])
        if rep.size == 0:
            if n > 1 and not (arr[0] == 0 and arr[-1] == n - 1 and np.array_equal(arr, np.arange(n, dtype=arr.dtype))):
                return False
            per = n                                   # arange(N): every row its own fold
        else:
            per = int(rep[0]) + 1
            if not np.array_equal(arr[per:], arr[:-per]) or np.unique(arr[:per]).size != per:
                return False
        keys = arr[:per]                              # iterating yields the labels as NumPy scalars
        if n % per == 0:
            mat = np.ascontiguousarray(np.arange(n, dtype=int).reshape(n // per, per).T)
            vals = list(mat)
        else:
            parts = [np.arange(f, n, per, dtype=int) for f in range(per)]
            base = np.concatenate(parts)
            bounds = np.zeros(per + 1, dtype=np.int64)
            np.cumsum([q.size for q in parts], out=bounds[1:])
            self._starts = bounds[:-1].copy()
            lo, hi = bounds[:-1].tolist(), bounds[1:].tolist()
            vals = [base[a:b] for a, b in zip(lo, hi)]
        self.folds_dict = dict(zip(keys, vals))
        return True

    def csr(self) -> Tuple[np.ndarray, np.ndarray]:
        """All folds in ``folds_dict`` order as (indices int64[n], offsets int64[P+1])."""
        parts = list(self.folds_dict.values())
        sizes = np.fromiter((p.size for p in parts), dtype=np.int64, count=len(parts))
        offsets = np.zeros(len(parts) + 1, dtype=np.int64)
        np.cumsum(sizes, out=offsets[1:])
        idx = (np.concatenate(parts).astype(np.int64, copy=False) if parts
               else np.zeros(0, dtype=np.int64))
        return idx, offsets
