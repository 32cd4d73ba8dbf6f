"""Build-container-only checks against the UNMODIFIED reference (/root/reference through oracle/ref_harness.py).  Skipped wherever the
reference tree is absent (the GPU box).

  * the committed fixtures are what the reference produces today: smpl / geom / nets / one grecon case regenerated into a temp dir and
    compared BIT FOR BIT with tests/golden/;
  * the CPU baseline bench.py reports is the port (oracle/port) because the reference cannot travel: its speed per Adam iteration is
    pinned against the reference's here (the port must not be the slower one)."""
import os
import time
import numpy as np
import pytest

from oracle import ref_harness as rh

pytestmark = [pytest.mark.reference, pytest.mark.skipif(not rh.available(), reason='/root/reference is not present')]


def test_committed_fixtures_are_what_the_reference_produces(tmp_path, monkeypatch):
    from oracle import make_golden as mg
    keep_cwd = os.getcwd()
    monkeypatch.setattr(mg, 'GOLD', str(tmp_path))
    try:
        os.chdir(rh.setup())                  # the reference globs its configs / assets relative to the cwd
        mg.gen_smpl()
        mg.gen_geom()
        mg.gen_nets()
        mg.gen_grecon([c for c in mg.GRECON_CASES if c[0] == 'glamr_static'])
    finally:
        os.chdir(keep_cwd)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    for name in ('smpl.npz', 'geom.npz', 'nets.npz', 'grecon_glamr_static_T90_P1.npz'):
        new, old = np.load(os.path.join(str(tmp_path), name)), np.load(os.path.join(gold, name))
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            assert np.array_equal(new[k], old[k], equal_nan=True), '%s[%s] is no longer what the reference produces' % (name, k)


def test_port_speed_is_the_reference_speed(asset_root):
    """ms per Adam iteration of oracle/port vs the reference classes on the 300-frame sequence of BASELINE configs[1], same thread count,
    best of 3 blocks of 6 iterations each.  Measured in round 2: port 130 ms, reference 162 ms per iteration (8 threads) -- the port is the FASTER
    of the two (ratio 0.80), so a GPU / CPU ratio quoted against it understates the gain over the reference; the bound keeps it that way."""
    import torch
    from oracle import make_golden as mg
    from oracle.port import build
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.utils import synth
    keep_cwd, keep_threads = os.getcwd(), torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    try:
        cfg = get_config('glamr_dynamic')
        spec = cfg['opt_stage_specs']['init_opt']
        in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
        port = build.load_optimizer(asset_root, cfg)
        pdata = port.init_data(in_dict)
        os.chdir(rh.setup())
        ref, rcfg = rh.reference_optimizer('glamr_dynamic', log=rh.QuietLog())
        rdata = ref.init_data(in_dict)

        def block(model, data, n=6):
            t0 = time.time()
            model.optimize_main(data, spec['opt_variables'], spec['opt_lr'], n, spec['loss_cfg'], {'stage': 'init_opt'})
            return (time.time() - t0) / n
        block(port, pdata, 2), block(ref, rdata, 2)                      # warm-up
        # blocks interleaved (a busy host slows both alike); a timing comparison says nothing when other jobs hold most of the cores
        if os.getloadavg()[0] > 0.75 * (os.cpu_count() or 1):
            pytest.skip('host busy (load %.1f on %d cores): no timing comparison' % (os.getloadavg()[0], os.cpu_count() or 1))
        tp, tr = [], []
        for _ in range(3):
            tp.append(block(port, pdata))
            tr.append(block(ref, rdata))
        t_port, t_ref = min(tp), min(tr)
    finally:
        os.chdir(keep_cwd)
        torch.set_num_threads(keep_threads)
    print('ms per Adam iteration: port %.1f, reference %.1f (ratio %.3f)' % (t_port * 1e3, t_ref * 1e3, t_port / t_ref))
    assert 0.6 < t_port / t_ref < 1.1
