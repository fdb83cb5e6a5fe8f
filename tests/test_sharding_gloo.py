"""N>1 path on CPU: two gloo ranks, each owning one slab of the element range.

Checks (a) the slabs tile the global range and each rank's closed-form slab equals the
global arrays' slice, (b) running the path per slab (the oracle stands in for the GPU
kernel here) gives exactly the unsharded result — elements are independent, so no
collective belongs on the data path, (c) the norm reduction and the max-over-ranks
timing used by bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as po

import tinman_sandbox_amd as tsa
from tinman_sandbox_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_elems, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O = po.Oracle()
        nets, nete = sharding.shard_range(total_elems, rank, world)
        slab = tsa.ElementArrays(4, 72, nete - nets, device="cpu").init_data(first_elem=nets).to_numpy()
        glob = O.init_arrays(4, 72, 1, 3, total_elems)
        for n in tsa.ARRAY_NAMES:
            assert np.array_equal(slab[n], glob[n][nets:nete]), n
        sc = po.default_scalars(72)
        Dvv = O.dvv_np4(False)
        O.compute_and_apply_rhs(slab, Dvv, sc)            # this rank's slab, local indices
        O.compute_and_apply_rhs(glob, Dvv, sc)            # unsharded reference
        for n in tsa.ARRAY_NAMES:
            assert np.array_equal(slab[n], glob[n][nets:nete]), n
        # norms: per-element partial sums -> gathered in rank order
        per = np.zeros((nete - nets, 3))
        for e in range(nete - nets):
            per[e, 0] = O.compute_norm(slab["elem_state_v"][e, sc["np1"]]) ** 2
            per[e, 1] = O.compute_norm(slab["elem_state_T"][e, sc["np1"]]) ** 2
            per[e, 2] = O.compute_norm(slab["elem_state_dp3d"][e, sc["np1"]]) ** 2
        got = sharding.gather_slab_norms(torch.from_numpy(per), dist)
        want = tuple(O.state_norms(glob, Dvv, sc))
        assert got == want, (got, want)
        # timing reduction
        t = sharding.max_over_ranks([1.0 + rank, 5.0 - rank], dist)
        assert t == [float(world), 5.0]
        # per-rank figures of bench.py's roofline.per_gpu, in rank order on every rank
        g = sharding.gather_over_ranks([float(nete - nets), 0.25 * (rank + 1)], dist)
        assert g == [[float(sharding.shard_range(total_elems, r, world)[1] - sharding.shard_range(total_elems, r, world)[0]),
                      0.25 * (r + 1)] for r in range(world)]
        dist.barrier()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total_elems", [7, 8])
def test_two_rank_sharding_gloo(tmp_path, total_elems):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total_elems, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_eight_rank_sharding_gloo(tmp_path):
    """The world size the driver's scaling run ends at: eight gloo ranks (CPU), 19 elements — slabs of 3, 3, 3, 3, 3, 3, 1, 0:
    a short and an EMPTY last rank — tile the range, each slab's result equals the unsharded one, and the norm / max / gather
    collectives bench.py uses agree on every rank."""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), 19, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))
    assert [sharding.shard_range(100000, r, 8) for r in (0, 7)] == [(0, 12500), (87500, 100000)]   # configs[2]


def test_single_process_paths():
    per = torch.tensor([[1.0, 4.0, 9.0], [3.0, 12.0, 16.0]])
    assert sharding.gather_slab_norms(per) == (2.0, 4.0, 5.0)
    assert sharding.max_over_ranks([1.5, 2.5]) == [1.5, 2.5]
    assert sharding.gather_over_ranks([1.5, 2.5]) == [[1.5, 2.5]]
