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


def _bench_worker(rank, world, port, q, mode):
    import io
    import json
    from contextlib import redirect_stdout
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    buf = io.StringIO()
    argv = ['--gpus', str(world), '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--stub-model', '--batch', '5', '--mode', mode, '--total', '7']
    with redirect_stdout(buf):
        out = bench.run(argv)
    q.put((rank, buf.getvalue(), out))


def _run_bench_skeleton(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_bench_distributed_skeleton_over_gloo():
    """bench.run()'s multi-rank protocol -- process-group init, rank-0 asset creation behind a barrier, disjoint seeds per rank,
    barrier + max-over-ranks timing, ONE JSON line on rank 0 only -- executed with world_size 2 over gloo and a stub model."""
    import json
    (r0, text0, out0), (r1, text1, out1) = _run_bench_skeleton('weak')
    assert out1 is None and text1.strip() == ''                       # only rank 0 reports
    line = json.loads(text0.strip())
    assert line == out0 and line['n_gpus'] == 2 and line['steps'] == 3 and line['scaling'] == 'weak'
    assert line['config']['sequences_total'] == 10                    # 5 per rank, whole-job aggregate
    assert line['config']['seeds_first_last'] == [0, 4]               # rank 0's block; rank 1 works on 5..9
    # 3 steps of 5 sequences at 2 ms each per rank: the whole job is 10 sequences per step in >= 10 ms
    assert 0 < line['value'] <= 10 / 0.010 * 1.05 and abs(line['ms_per_step'] * line['value'] / 1e3 - 10) < 1e-6


def test_bench_strong_scaling_split():
    """--mode strong: a fixed job (BASELINE configs[2]) split over the ranks in balanced contiguous blocks."""
    (r0, text0, out0), _ = _run_bench_skeleton('strong')
    assert out0['scaling'] == 'strong' and out0['config']['sequences_total'] == 7 and out0['config']['seeds_first_last'] == [0, 3]
