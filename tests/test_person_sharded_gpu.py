"""MI355X side of the person-sharded variant of BASELINE configs[3] (glamr_amd/parallel.py): the `frozen` person slots of the stage kernel
against the CPU runtime of the same algorithm, and the launch-by-launch schedule with its collectives under an RCCL process group."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from tests import grecon_common as gc

pytestmark = pytest.mark.gpu


def test_frozen_person_slots_on_the_device_equal_the_cpu_runtime(asset_root):
    """One gradient launch (niters 1, lr 0, grads_out) of the main stage of glamr_static_multi on a 4-person scene in which persons 2 and 3
    belong to "another rank": their world pose is given, they have no residuals of their own and receive no gradient, the camera-only terms
    are left to the owner of the camera.  Device kernel vs the single-threaded host instance of grecon_algo.hpp: same gradients."""
    from oracle.port import build
    from glamr_amd.global_recon import packing
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.utils import synth
    cfg = get_config('glamr_static_multi')
    specs = cfg['grecon_model_specs']
    T, P = 100, 4
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
    ora = build.load_optimizer(asset_root, cfg)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
    jl = gc.j_local_from_oracle(ora.smpl, data)
    spec = cfg['opt_stage_specs']['main_opt']
    got = {}
    for name, (run, dev) in (('cpu', gc.hostsim_runner()), ('gpu', gc.device_runner())):
        packed = packing.PackedScenes([data], [jl], dev)
        frozen = torch.tensor([0, 0, 1, 1], dtype=torch.int32)
        packed.t['frozen'] = frozen.to(dev)
        # the poses "the other rank published": the initial ones, nudged
        bo, bt = packed.t['base_orient'].view(1, P, T, 3), packed.t['base_trans'].view(1, P, T, 3)
        bo[:, 2:] += 0.01
        bt[:, 2:] += 0.02
        sd = packing.stage_desc(spec, specs, False, niters=1)
        sd.lr = 0.0
        sd.flags |= packing.FLAG_NO_CAMERA_TERMS
        before = packed.t['params'].clone()
        grads = run(packed, sd, True)
        assert torch.equal(packed.t['params'].cpu()[:, 9 * T:], before.cpu()[:, 9 * T:])          # lr 0: nothing but the camera block (set from cam_pose) moved
        got[name] = grads.cpu().numpy()[0]
    l = packing.param_layout_py(P, T)
    own = got['gpu'][:l['person0'] + 2 * l['person_stride']]
    assert np.abs(own).max() > 1e-3                                                        # there is a gradient
    assert not got['gpu'][l['person0'] + 2 * l['person_stride']:].any()                    # ... and none for the frozen persons
    scale = np.abs(got['cpu']).max()
    err = np.abs(got['gpu'] - got['cpu']).max()
    print('frozen slots: device vs CPU runtime gradient difference %.2e (largest gradient %.2e)' % (err, scale))
    assert err < 3e-4 * max(1.0, scale)


def test_person_sharded_schedule_under_an_rccl_group(monkeypatch):
    """bench.person_sharded_line inside a `nccl` (= RCCL) process group of one rank: the launch-by-launch schedule with its all-gather / all-reduce
    calls on device tensors next to the default one-workgroup schedule, 5 iterations per stage on two 4-person scenes -- same projections."""
    import torch.distributed as dist
    import bench
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', str(port)), ('RANK', '0'), ('LOCAL_RANK', '0'), ('WORLD_SIZE', '1'), ('HSA_ENABLE_IPC_MODE_LEGACY', '0')):
        monkeypatch.setenv(k, v)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)
    try:
        out = bench.person_sharded_line(bench.ensure_assets(), dev, 0, 1, iters=5, scenes=2)
    finally:
        dist.destroy_process_group()
    print(out)
    assert out['launches_per_iteration'] == 2.0 and out['persons_per_rank'] == 4
    assert out['median_projection_difference_px'] < 0.05
    assert out['person_sharded_us_per_iteration'] > out['in_kernel_reduction_us_per_iteration'] > 0


def test_kept_tables_and_silent_gradient_launches_change_nothing(asset_root, monkeypatch):
    """Round 6: the gradient launches of a sharded stage share ONE workspace and, from the second on, skip the stage-constant part of the set-up
    (GLAMR_FLAG_KEEP_TABLES); every one but the stage's last skips the reporting part of its evaluation (GLAMR_FLAG_NO_REPORT); iterations 2 .. n - 2 are
    replays of one captured graph.  None of it is arithmetic: against the schedule with a fresh workspace, a full set-up and plain launches every
    iteration (GLAMR_SHARDED_KEEP_TABLES=0, GLAMR_SHARDED_GRAPH=0) parameters, projections and loss values must agree bit for bit -- both stages of
    glamr_static_multi, 4 persons with ragged existence ranges, one rank without a process group."""
    from glamr_amd import parallel
    from glamr_amd.global_recon import packing
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.utils import synth
    from oracle.port import build
    cfg = get_config('glamr_static_multi')
    T, P, K = 120, 4, 7
    in_dict = synth.trim_person(synth.make_in_dict(seed=21, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model()), 1, 9, 101)
    ora = build.load_optimizer(asset_root, cfg)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 21))
    jl = gc.j_local_from_oracle(ora.smpl, data)
    dev = torch.device('cuda:0')
    out = {}
    for name, env in (('kept', {}), ('plain', {'GLAMR_SHARDED_KEEP_TABLES': '0', 'GLAMR_SHARDED_GRAPH': '0'})):
        for k in ('GLAMR_SHARDED_KEEP_TABLES', 'GLAMR_SHARDED_GRAPH'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        packed = packing.PackedScenes([data, data], [jl, jl], dev)
        sched = parallel.PersonShardedSchedule(rank=0, world=1, use_dist=False)
        sched.run(packed, cfg['opt_stage_specs'], cfg['grecon_model_specs'], max_iters=K)
        torch.cuda.synchronize()
        out[name] = {k: packed.t[k].clone() for k in ('params', 'kp_2d_pred', 'orient_world', 'trans_world', 'cam_pose', 'losses')}
        if name == 'kept':
            assert getattr(sched, 'iteration_graphs', 0) == len(cfg['opt_stage_specs'])          # the captured iterations were used
    for k in out['kept']:
        assert torch.equal(out['kept'][k], out['plain'][k]), k
