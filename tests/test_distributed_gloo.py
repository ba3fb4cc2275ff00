"""CPU (-m "not gpu"), world_size 2 over gloo: the multi-GPU plumbing of
cvmatrix_amd/distributed.py -- fold assignment, the packed [G | H | gstats] exchange
(all-reduce for the row-sharded fit, broadcast for the replicated fit) and that fold
results computed from the exchanged globals equal the single-process results.  The local
Gram partials come from the oracle here (the HIP kernels need a GPU); on the GPU box the
same collectives run over RCCL."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

from cvmatrix_amd.distributed import (
    allreduce_globals,
    assign_folds,
    broadcast_globals,
    pack_globals,
    unpack_globals,
)


def test_assign_folds_balances_and_covers():
    owned = assign_folds([10] * 10, 8)
    assert sorted(f for o in owned for f in o) == list(range(10))
    assert max(len(o) for o in owned) == 2 and min(len(o) for o in owned) == 1
    owned = assign_folds([5, 100, 3, 50, 50, 1, 0], 3)
    assert sorted(f for o in owned for f in o) == list(range(7))
    loads = [sum([5, 100, 3, 50, 50, 1, 0][f] for f in o) for o in owned]
    assert max(loads) == 100 and min(loads) >= 50
    assert assign_folds([], 4) == [[], [], [], []]
    assert assign_folds([7, 7], 1) == [[0, 1]]


def test_shard_folds_strong_scaling_layout():
    """bench.py --scaling strong: ONE problem's folds dealt over the ranks; a rank holds the
    rows of its folds, renumbered locally, and a Partitioner over its local labels yields
    those folds in the order of its keys."""
    from cvmatrix_amd import Partitioner
    from cvmatrix_amd.distributed import shard_folds

    N, P = 103, 10
    labels = np.arange(N) % P
    for world in (1, 2, 4, 8, 16):
        seen_rows, seen_keys = [], []
        for rank in range(world):
            keys, rows, local = shard_folds(labels, world, rank)
            assert list(rows) == sorted(rows) and np.array_equal(labels[rows], local)
            assert set(np.unique(local)) == set(keys)
            lp = Partitioner(local)
            assert list(lp.folds_dict) == keys
            for k in keys:     # local validation indices map back to the global fold
                assert np.array_equal(rows[lp.get_validation_indices(k)], np.flatnonzero(labels == k))
            seen_rows += list(rows)
            seen_keys += keys
        assert sorted(seen_rows) == list(range(N)) and sorted(seen_keys) == list(range(P))
        assert max(len(shard_folds(labels, world, r)[0]) for r in range(world)) == -(-P // world)
    # ragged folds with non-integer labels: balanced by rows, first-seen order kept
    lab = ["a"] * 50 + ["b"] * 10 + ["c"] * 30 + ["d"] * 10
    k0, r0, _ = shard_folds(lab, 2, 0)
    k1, r1, _ = shard_folds(lab, 2, 1)
    assert k0 == ["a"] and k1 == ["b", "c", "d"] and r0.size == 50 and r1.size == 50


def test_pack_unpack_roundtrip():
    G = torch.arange(9.0, dtype=torch.float64).reshape(3, 3)
    H = torch.arange(6.0, dtype=torch.float32).reshape(3, 2)
    gs = torch.arange(12.0, dtype=torch.float64)
    buf = pack_globals(G, H, gs)
    assert buf.dtype == torch.float64 and buf.numel() == 9 + 6 + 12
    G2, H2, gs2 = torch.zeros_like(G), torch.zeros_like(H), torch.zeros_like(gs)
    unpack_globals(buf, G2, H2, gs2)
    assert torch.equal(G, G2) and torch.equal(H, H2) and torch.equal(gs, gs2)
    G3, gs3 = torch.zeros_like(G), torch.zeros_like(gs)
    unpack_globals(pack_globals(G, None, gs), G3, None, gs3)
    assert torch.equal(G, G3) and torch.equal(gs, gs3)


def _globals_from_oracle(X, Y, w):
    """[G | H | gstats] of a row block, gstats laid out like cvm_gram_fit's output."""
    from oracle.cvmatrix_oracle import fit_globals

    g = fit_globals(X, Y, None if w is None else w.reshape(-1, 1), True, True, True, True)
    N = X.shape[0]
    sw = float(g["sw"]) if w is not None else float(N)
    nz = float(g["nz"]) if w is not None else float(N)
    gs = np.concatenate([g["sX"].ravel(), g["qX"].ravel(), g["sY"].ravel(), g["qY"].ravel(),
                         [sw, nz]])
    return (torch.from_numpy(g["G"].copy()), torch.from_numpy(g["H"].copy()),
            torch.from_numpy(gs))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.cvmatrix_oracle import OracleCVMatrix

        rng = np.random.default_rng(5)
        N, K, M, P = 240, 6, 2, 8                      # P folds over the whole data set
        X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
        w[::17] = 0.0
        # rows are sharded by blocks; fold f lives entirely in one block (weak-scaling layout)
        rows = np.arange(N).reshape(world, N // world)[rank]
        local_folds = [rows[i::P // world] for i in range(P // world)]
        # ---- row-sharded fit: local partial + ONE all-reduce --------------------------
        G, H, gs = _globals_from_oracle(X[rows], Y[rows], w[rows])
        allreduce_globals(G, H, gs)
        Gf, Hf, gsf = _globals_from_oracle(X, Y, w)
        assert torch.allclose(G, Gf, rtol=1e-12, atol=1e-12)
        assert torch.allclose(H, Hf, rtol=1e-12, atol=1e-12)
        assert torch.allclose(gs, gsf, rtol=1e-12, atol=1e-12)
        assert int(round(float(gs[-1]))) == int(np.count_nonzero(w))
        # ---- fold update from the exchanged globals = single-process result ------------
        full = OracleCVMatrix()
        full.fit(X, Y, w)
        shard = OracleCVMatrix()
        shard.fit(X[rows], Y[rows], w[rows])           # local rows, then global statistics
        shard.XTX, shard.XTY = G.numpy(), H.numpy()
        gsn = gs.numpy()
        shard.sum_X, shard.sum_sq_X = gsn[0:K].reshape(1, -1), gsn[K:2 * K].reshape(1, -1)
        shard.sum_Y = gsn[2 * K:2 * K + M].reshape(1, -1)
        shard.sum_sq_Y = gsn[2 * K + M:2 * K + 2 * M].reshape(1, -1)
        shard.sum_w, shard.num_nonzero_w = gsn[-2], int(round(gsn[-1]))
        for v_global in local_folds:
            v_local = v_global - rows[0]
            (a, b), sa = shard.training_XTX_XTY(v_local)
            (c, d), sc = full.training_XTX_XTY(v_global)
            np.testing.assert_allclose(a, c, rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(b, d, rtol=1e-10, atol=1e-10)
            for s, t in zip(sa, sc):
                np.testing.assert_allclose(s, t, rtol=1e-10)
        # ---- the same exchange on ONE contiguous buffer (how CVMatrix lays out float64 globals:
        #      XTX, XTY, gstats are views of it; nothing is packed, one collective) ----------
        Gl, Hl, gsl = _globals_from_oracle(X[rows], Y[rows], w[rows])
        flat = torch.empty(K * K + K * M + gsl.numel(), dtype=torch.float64)
        Gv, Hv, gv = flat[:K * K].view(K, K), flat[K * K:K * K + K * M].view(K, M), flat[K * K + K * M:]
        Gv.copy_(Gl); Hv.copy_(Hl); gv.copy_(gsl)
        allreduce_globals(Gv, Hv, gv, flat=flat)
        assert torch.equal(Gv, G) and torch.equal(Hv, H) and torch.equal(gv, gs)
        if rank != 0:
            flat.zero_()
        broadcast_globals(Gv, Hv, gv, src=0, flat=flat)
        assert torch.equal(Gv, G) and torch.equal(gv, gs)
        # ---- replicated fit: rank 0 computes, broadcast ---------------------------------
        if rank == 0:
            G2, H2, gs2 = Gf.clone(), Hf.clone(), gsf.clone()
        else:
            G2, H2, gs2 = torch.zeros_like(Gf), torch.zeros_like(Hf), torch.zeros_like(gsf)
        broadcast_globals(G2, H2, gs2, src=0)
        assert torch.equal(G2, Gf) and torch.equal(H2, Hf) and torch.equal(gs2, gsf)
        # every rank deals the same folds to itself
        sizes = [N // P] * P
        mine = assign_folds(sizes, world)[rank]
        got = [None] * world
        dist.all_gather_object(got, mine)
        assert sorted(f for o in got for f in o) == list(range(P))
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_process_exchange(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
