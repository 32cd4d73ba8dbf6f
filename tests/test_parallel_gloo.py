"""world_size-2 gloo tests of the sequence-sharding layer used by bench.py --gpus N (no GPU needed)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from glamr_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    r, w, lr = parallel.env_rank_world()
    lo, hi = parallel.shard_range(7, r, w)
    seeds = parallel.weak_scaling_seeds(4, r)
    parallel.barrier()
    slowest = parallel.max_over_ranks(1.0 + rank)                     # rank 1 is "slower"
    gathered = parallel.gather_results([('seq%d' % i, i * i) for i in range(lo, hi)])
    q.put((rank, (lo, hi), seeds, slowest, gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reductions():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, rng0, seeds0, slow0, g0), (r1, rng1, seeds1, slow1, g1) = res
    assert rng0 == (0, 4) and rng1 == (4, 7)                          # contiguous, balanced, covers everything exactly once
    assert seeds0 == [0, 1, 2, 3] and seeds1 == [4, 5, 6, 7]          # weak scaling: disjoint work per rank
    assert slow0 == slow1 == 2.0                                      # every rank agrees on the slowest rank's time
    assert g1 is None and g0 == [('seq%d' % i, i * i) for i in range(7)]


def test_shard_range_properties():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            blocks = [parallel.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
