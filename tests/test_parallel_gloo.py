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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain_env():
    """A shell the driver would start a command from: no RANK / WORLD_SIZE / MASTER_* set."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE', 'GROUP_RANK')}
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return env


def test_bench_launched_plainly_spawns_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher and no rendezvous variables in the environment (what a SCALE driver that does not know about
    torchrun would type): bench.py re-executes itself under torch.distributed.run with two ranks; rc 0, exactly ONE line on stdout -- rank 0's
    JSON -- carrying the weak line AND the BASELINE configs[2] strong line of the same run."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub-model', '--backend', 'gloo', '--steps', '3', '--warmup', '1',
                        '--batch', '6'], env=_plain_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['process_group_ranks'] == 2 and out['config']['sequences_total'] == 12 and out['scaling'] == 'weak'
    strong = out['configs2_strong_64']
    assert strong['scaling'] == 'strong' and strong['sequences_total'] == 64 and '32 on rank 0' in strong['workload'] and strong['sequences_per_sec'] > 0
    # inside a launcher with the wrong world size the message says what to do instead of a bare exit
    env = dict(_plain_env(), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub-model', '--backend', 'gloo'], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr


def test_run_dataset_sharded_over_two_ranks_prints_the_single_process_line(tmp_path):
    """`python -m glamr_amd.global_recon.run_dataset --gpus 2` (BASELINE configs[4]'s driver) launched plainly: the sequence list is split in
    contiguous blocks, every rank reconstructs and evaluates its own, rank 0 gathers the per-sequence metrics in sequence order and prints ONE
    metric line -- equal to the line of the single-process run.  CPU stand-ins for the optimiser / evaluator (no GPU here); five sequences on
    two ranks = blocks of 3 + 2; result pickles of BOTH ranks' sequences are on disk afterwards."""
    import pickle
    import subprocess
    import sys
    import numpy as np
    from glamr_amd.utils import synth
    md = synth.make_smpl_model()
    out_dir, gt_dir = tmp_path / 'out', tmp_path / 'gt'
    os.makedirs(gt_dir)
    names = ['s%d' % i for i in range(5)]
    for i, seq in enumerate(names):
        d = synth.make_in_dict(seed=50 + i, num_frames=40 + 3 * i, num_persons=1, smpl_model=md, with_gt=True)
        os.makedirs(out_dir / seq / 'pose_est')
        pickle.dump(d['est'], open(out_dir / seq / 'pose_est' / 'pose.pkl', 'wb'))
        pickle.dump({'person_data': d['gt'], 'meta': {}}, open(gt_dir / (seq + '.pkl'), 'wb'))
    base = ['--cfg', 'glamr_static', '--dataset', '', '--seqs'] + names + ['--gt_dir', str(gt_dir), '--seeds', '1', '2', '--stub-model', '--cached', '0']
    from glamr_amd.global_recon import run_dataset
    single = run_dataset.main(base + ['--out_dir', str(out_dir)])
    assert single is not None and 'sequences: s0,s1,s2,s3,s4' in single
    for seq in names:
        os.remove(out_dir / seq / 'grecon' / ('%s_seed1.pkl' % seq))
    r = subprocess.run([sys.executable, '-m', 'glamr_amd.global_recon.run_dataset'] + base + ['--out_dir', str(out_dir), '--gpus', '2'],
                       env=_plain_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if 'G-MPJPE' in l]
    assert lines == [single], (r.stdout, single)
    for seq in names:                                   # written by rank 0 (s0..s2) and rank 1 (s3, s4)
        assert os.path.exists(out_dir / seq / 'grecon' / ('%s_seed1.pkl' % seq))
