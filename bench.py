"""Headline benchmark: sequences/sec of end-to-end global reconstruction (GlobalReconOptimizer.optimize_resident) on 300-frame,
1-person, dynamic-camera synthetic sequences (BASELINE.json configs[1], cfg `glamr_dynamic`, 500 Adam iterations).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode weak|strong --total 64]

A "step" = one pass of the hot path over one batch of B independent synthetic sequences per GPU: per-person preparation of the
HybrIK arrays, motion infilling + trajectory prediction, SMPL skinning, scene initialisation and the full optimisation schedule.
The HybrIK arrays are uploaded once before the timed region (stage_inputs), so `value` is the HBM-in / HBM-out rate the contract asks
for; the rate from HOST dictionaries to HOST dictionaries (optimize_batch / the pipelined optimize_stream) is measured after it and
reported as `host_inclusive_sequences_per_sec`.  Multi-GPU: one process per GPU (torch.distributed over RCCL), sequences are
independent, so ranks share nothing on the data path; the only collectives are the barrier and the max-reduction of the elapsed
time.  `--mode weak` (default): every rank works on its own B sequences; `--mode strong --total 64`: BASELINE configs[2], a fixed set
of 64 sequences split over the ranks.

After the warm-up the whole step is captured as HIP graphs, one set per stream, checked against a plain step bit for bit (same seed for the
sampled latents) and replayed in the timed region: the same kernels on the same buffers with a couple of launches on the host instead of ~25,
so a host with slow driver calls does not pace the GPU (`config.step_graph`; `--no-graph-step`, or any failure of capture / check: plain launches).

With two streams (the default) the batches are CO-SCHEDULED (`config.coscheduled_streams`, `pipeline`): a gate -- product code,
GlobalReconOptimizer.pipeline_gate, also what optimize_stream() uses -- starts a batch when the previous batch's priors are done, and its
motion infiller runs on kernels written to fit beside a resident workgroup of that batch's optimiser stage (no LDS, one wave per workgroup,
fragment-major activations: csrc/nn_free.hpp).  Same work, same results (the infiller's outputs agree with the LDS kernels' to 7e-7); the
stage launch takes longer beside them than alone, the step is shorter.  `--no-coschedule` / GLAMR_COSCHEDULE=0: the two streams left to
themselves on the LDS kernels.

Kernel durations come from the kernel's own clock (stamps in the workspace header = what rocprofv3 reports for the dispatch), and the
roofline launch is measured with the GPU to itself (see run()).

Prints ONE JSON line on rank 0:
  roofline      the dominant kernel (the fused optimiser stage).  It is LATENCY / ISSUE bound: `us_per_scene_iteration` against the
                1.45 us dependent-boundary floor is the figure that describes it; `achieved` / `frac` are the contract's NOTIONAL HBM line
                (SURVEY 8d's live-state bytes over the launch time -- bytes the kernel keeps on chip and does not move); `traffic` are the
                memory-side bytes of one launch from rocprofv3 PMC passes on the shipped instance (the newest profiles/rNN_pmc_stage_kernel.json).
                Measured with the GPU to itself; `pipeline` holds what the launch takes beside the other stream's priors
  kernels       stand-alone rooflines SURVEY 8(d) asks for: SMPL skinning (B = 300 and 19 200, with and without vertices), the priors'
                GEMMs, and the other BASELINE configs that fit one GPU (configs[0] 120-frame infiller + skinning, configs[3] 4-person scenes)
  cpu_baseline  the CPU oracle (a port of the reference, oracle/port) timed on a bounded sample on this box's host cores
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM-level traffic of one (person, iteration) of the fused optimiser, SURVEY.md 8(d) K5: parameters + Adam moments
# (read + write) 2 x 58 KB, cached joints 94 KB, keypoint targets + scores 94 KB, intrinsics 11 KB, trajectory prior 13 KB,
# HybrIK orientation/translation 7 KB, transforms 60 KB.  NOTIONAL: the kernel keeps this state on chip (LDS arena) and does not move it.
ALGO_BYTES_PER_PERSON_ITER = (2 * 58 + 94 + 94 + 11 + 13 + 7 + 60) * 1024
HBM_PEAK_GBS = 8000.0
F32_MFMA_PEAK_TFLOPS = 157.3
DEPENDENT_BOUNDARY_US = 1.45
SMPL_FLOP_PER_FRAME = 15.8e6                  # blend shapes + skinning + regression, with vertices
SMPL_BYTES_PER_FRAME_VERTS = 6890 * 3 * 4 + 26 * 3 * 4 + 82 * 4      # vertex + joint write-out, pose / shape read
NETS_FLOP_PER_SEQUENCE = 3.2e9                # infiller 272 MFLOP per window x 10 windows + trajectory predictor 0.93 GFLOP at 300 frames
PORT_OVER_REFERENCE = 0.80                    # time per iteration, port / unmodified reference, build container, 8 threads (130 vs 162 ms: tests/test_reference_container.py)
FP16X3_PEAK_TFLOPS = 2500.0 / 3.0             # fp32-grade products on the fp16 matrix cores cost three MFMAs per k step (hi*hi + hi*lo + lo*hi): dense fp16 peak / 3


def pmc_file():
    """The newest profiles/rNN_pmc_stage_kernel.json (written by tools/collect_pmc.py from rocprofv3 --pmc passes on the shipped instance)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_stage_kernel.json')))
    return files[-1] if files else None



CFG_ID, NUM_FRAMES = 'glamr_dynamic', 300


def build_model(asset_root, device, cfg_id=CFG_ID):
    import torch  # noqa: F401
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.models.prior_models import MotionTrajJointModel
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(device)
    mt = MotionTrajJointModel(None, device, None, smpl=smpl, results_root=os.path.join(asset_root, 'results'))
    return model_dict['global_recon_model'](get_config(cfg_id), device, None, smpl=smpl, mt_model=mt)


def ensure_assets():
    from glamr_amd.utils import synth
    from glamr_amd.models.layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT
    root = os.path.join(tempfile.gettempdir(), 'glamr_bench_assets_%d' % os.getuid())
    if not os.path.exists(os.path.join(root, 'data', 'J_regressor_extra.npy')):
        synth.write_smpl_assets(root)
    ck = os.path.join(root, 'results', 'traj_pred', 'traj_pred_demo', 'version_0', 'checkpoints', 'model-best-epoch=0000.ckpt')
    if not os.path.exists(ck):
        synth.write_checkpoints(root, INFILLER_LAYOUT, TRAJPRED_LAYOUT)
    return root


def cpu_baseline(asset_root, iters=40):
    """The CPU oracle (oracle/port: torch autograd + Adam, full SMPL skinning per iteration like the reference) on ONE 300-frame
    sequence: init_data in full, a scan over intra-op thread counts (4 iterations each), then `iters` (>= 40) of the 500
    iterations timed at the fastest count and extrapolated to the schedule (about 10-20 s of CPU work in total).  /root/reference does
    not exist on the GPU box, so this is the port; tests/test_reference_container.py pins port vs reference speed in the build container."""
    import torch
    from oracle.port import build
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.utils import synth
    cfg = get_config(CFG_ID)
    opt = build.load_optimizer(asset_root, cfg)
    in_dict = synth.make_in_dict(seed=0, num_frames=NUM_FRAMES, num_persons=1, smpl_model=synth.make_smpl_model())
    ncpu = os.cpu_count() or 1
    t0 = time.time()
    data = opt.init_data(in_dict)
    t_init = time.time() - t0
    spec = cfg['opt_stage_specs']['init_opt']
    run = lambda n: opt.optimize_main(data, spec['opt_variables'], spec['opt_lr'], n, spec['loss_cfg'], {'stage': 'init_opt'})
    scan = {}
    for nt in sorted({n for n in (4, 8, 16, 32) if n <= ncpu} | {min(ncpu, 64)}):
        torch.set_num_threads(nt)
        run(1)                                                      # warm-up
        t0 = time.time()
        run(4)
        scan[nt] = (time.time() - t0) / 4
    nt = min(scan, key=scan.get)
    torch.set_num_threads(nt)
    t0 = time.time()
    run(iters)
    per_iter = (time.time() - t0) / iters
    total = t_init + per_iter * spec['opt_niters']
    return {'value': 1.0 / total, 'unit': 'sequences/sec', 'cores': nt, 'kind': 'port', 'kind_note': 'the reference is absent on this box: oracle/port, pinned to it in the build container',
            'ms_per_iteration': per_iter * 1e3, 'iterations_timed': iters, 'port_over_reference': PORT_OVER_REFERENCE,
            'port_vs_reference': 'build container, 8 threads: port 130 ms, unmodified reference 162 ms per iteration (tests/test_reference_container.py): the port is the faster baseline',
            'sample': 'oracle/port on 1 sequence of %d frames: init_data (%.2f s) + thread scan %s ms/iter, then %d of %d Adam iterations at %d threads '
                      '(%.1f ms/iter), extrapolated to the full schedule (%.1f s/sequence)'
                      % (NUM_FRAMES, t_init, {k: round(v * 1e3, 1) for k, v in scan.items()}, iters, spec['opt_niters'], nt, per_iter * 1e3, total)}


def _timed(fn, reps=3, calls=1):
    """Best-of-`reps` duration of fn() in seconds, measured with HIP events on the current stream (the library launches there).  calls > 1: that many
    calls back to back between the two events, per call -- for calls of a millisecond or less, whose few launches the host enqueues more slowly than an
    idle GPU runs them (one call between two events then measures the interpreter: 1.15 against 0.95 ms for the skinning of 19 200 frames)."""
    import torch
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _c in range(calls):
            fn()
        b.record()
        torch.cuda.synchronize()
        dt = a.elapsed_time(b) * 1e-3 / calls
        best = dt if best is None else min(best, dt)
    return best


def kernel_lines(asset_root, model, dev):
    """Stand-alone measurements SURVEY 8(d) / BASELINE.md 5 ask for beside the headline (rank 0, after the timed region)."""
    import torch
    from glamr_amd.utils import synth
    from glamr_amd.models.priors import num_windows
    out = {}
    smpl = model.smpl
    g = torch.Generator(device='cpu').manual_seed(0)
    lines = []
    for B in (300, 19200):
        pose = (torch.randn(B, 72, generator=g) * 0.3).to(dev)
        betas, trans = torch.randn(B, 10, generator=g).to(dev), torch.randn(B, 3, generator=g).to(dev)
        for verts in (True, False):
            dt = _timed(lambda: smpl(global_orient=pose[:, :3], body_pose=pose[:, 3:], betas=betas, root_trans=trans, return_verts=verts), calls=20 if B > 1000 else 100)
            tf = SMPL_FLOP_PER_FRAME * B / dt / 1e12 if verts else None
            lines.append({'frames': B, 'vertices': verts, 'ms': round(dt * 1e3, 4),
                          'tflops_algorithmic': None if tf is None else round(tf, 2),
                          'frac_of_fp16x3_peak': None if tf is None else round(tf / FP16X3_PEAK_TFLOPS, 4),
                          'write_gbs': round(SMPL_BYTES_PER_FRAME_VERTS * B / dt / 1e9, 1) if verts else None,
                          'frac_of_hbm_peak': round(SMPL_BYTES_PER_FRAME_VERTS * B / dt / 1e9 / HBM_PEAK_GBS, 3) if verts else None})
    best = max((l['tflops_algorithmic'] or 0.0) for l in lines)
    out['smpl_lbs'] = {'timing': 'HIP events around 20 (19 200 frames) / 100 (300 frames) calls back to back, best of three, per call', 'flop_per_frame': SMPL_FLOP_PER_FRAME, 'fp16x3_peak_tflops': round(FP16X3_PEAK_TFLOPS, 1),
                       'frac_of_fp16x3_peak': round(best / FP16X3_PEAK_TFLOPS, 4), 'runs': lines}
    # the two priors on 1024 sequences of 300 frames (GEMM-dominated: the fp16-split MFMA kernels)
    md = synth.make_smpl_model()
    Bn, T = 1024, NUM_FRAMES
    pose = (torch.randn(Bn, T, 69, generator=g) * 0.2).to(dev)
    vis = torch.ones(Bn, T, device=dev)
    vis[:, 100:160] = 0
    meps, teps = torch.randn(Bn, num_windows(T), 128, generator=g).to(dev), torch.randn(Bn, 128, generator=g).to(dev)
    dt = _timed(lambda: model.mt_model.infer_padded(pose, vis, [T] * Bn, meps, teps), reps=2)
    out['priors'] = {'sequences': Bn, 'frames': T, 'ms': round(dt * 1e3, 2), 'sequences_per_sec': round(Bn / dt, 1),
                     'tflops_fp32_equivalent': round(NETS_FLOP_PER_SEQUENCE * Bn / dt / 1e12, 1), 'flop_per_sequence': NETS_FLOP_PER_SEQUENCE,
                     'frac_of_fp16x3_peak': round(NETS_FLOP_PER_SEQUENCE * Bn / dt / 1e12 / FP16X3_PEAK_TFLOPS, 4),
                     'note': 'infiller (10 autoregressive windows) + trajectory predictor; fp32-grade results on the fp16 matrix cores by 2-way operand splitting (3 MFMAs per k step)'}
    dtc = _timed(lambda: model.mt_model.infer_padded(pose, vis, [T] * Bn, meps, teps, coschedule=True), reps=2)
    out['priors']['coschedulable_kernels_ms'] = round(dtc * 1e3, 2)
    out['priors']['coschedulable_note'] = ('the same call with GLAMR_NETS_COSCHEDULE (what the two-stream pipeline uses): infiller on the LDS-free one-wave kernels -- '
                                           'slower ALONE (no on-chip fusion), but they run beside a resident optimiser stage, which the LDS kernels cannot')
    # BASELINE configs[0]: one 120-frame clip through the infiller, then full skinning with vertices -- batched over 1024 clips
    T0 = 120
    pose0, vis0 = pose[:, :T0].contiguous(), vis[:, :T0].contiguous()
    vis0[:, 40:64] = 0
    meps0 = meps[:, :num_windows(T0)].contiguous()
    betas0 = torch.randn(Bn * T0, 10, generator=g).to(dev)
    zero3 = torch.zeros(Bn * T0, 3, device=dev)

    def cfg0():
        o = model.mt_model.handle.infer(pose0, vis0, [T0] * Bn, motion_eps=meps0, traj=False)
        smpl(global_orient=zero3, body_pose=o['pose'].reshape(-1, 69), betas=betas0, root_trans=zero3, return_verts=True)
    dt = _timed(cfg0, reps=2)
    out['configs0_infiller_plus_lbs'] = {'workload': 'BASELINE configs[0]: 120-frame 1-person clips, motion infiller + SMPL LBS with vertices, batch of %d' % Bn,
                                         'sequences_per_sec': round(Bn / dt, 1), 'ms': round(dt * 1e3, 2)}
    # BASELINE configs[3]: 300-frame 4-person static-camera scenes, shared camera parameters, the whole schedule (200 + 500 iterations)
    m4 = build_model(asset_root, dev, 'glamr_static_multi')
    B4 = 64
    scenes = [synth.make_in_dict(seed=1000 + i, num_frames=NUM_FRAMES, num_persons=4, smpl_model=md) for i in range(B4)]
    rin4 = m4.stage_inputs(scenes)
    torch.cuda.synchronize()
    holder = {}

    def cfg3():
        holder['p'] = m4.optimize_resident(rin4)[1]
    dt = _timed(cfg3, reps=2)
    stage_ms = [m4.launch_ms(ws) for ws in holder['p'].stage_ws]
    iters = [s['opt_niters'] for s in m4.opt_stage_specs.values()]
    out['configs3_four_persons_shared_camera'] = {
        'workload': 'BASELINE configs[3]: 300-frame 4-person static-camera scenes (cfg glamr_static_multi: %s iterations), batch of %d scenes' % ('+'.join(map(str, iters)), B4),
        'scenes_per_sec': round(B4 / dt, 1), 'ms': round(dt * 1e3, 2), 'stage_launch_ms': [round(x, 2) for x in stage_ms],
        'us_per_scene_iteration': [round(x * 1e3 / n, 1) for x, n in zip(stage_ms, iters)],
        'shared_camera_reduction': 'in-kernel block reduction of the 9 shared camera gradients (one workgroup per scene); see `collective_alternative`',
        # the yardstick of `roofline` for this configuration: a workgroup walks its four persons one after the other (300 frames = 300 threads each),
        # so the dependent-boundary floor of SURVEY 8(d) K5 (1.45 us per person and iteration) counts four times
        'roofline': {'bound': 'latency/issue', 'unit': 'us per scene-iteration (lower is better; peak = 4 persons x the 1.45 us dependent-boundary floor)',
                     'peak': 4 * DEPENDENT_BOUNDARY_US, 'achieved': [round(x * 1e3 / n, 1) for x, n in zip(stage_ms, iters)],
                     'frac': [round(4 * DEPENDENT_BOUNDARY_US / (x * 1e3 / n), 4) for x, n in zip(stage_ms, iters)], 'traffic': None}}
    # latent-optimisation mode: the priors inside the Adam loop, one 300-frame sequence (what the mode is used on)
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.global_recon.models import model_dict
    cfg = get_config(CFG_ID)
    cfg['grecon_model_specs'].update(flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    ml = model_dict['global_recon_model'](cfg, dev, None, smpl=smpl, mt_model=model.mt_model)
    one = synth.make_in_dict(seed=0, num_frames=NUM_FRAMES, num_persons=1, smpl_model=md)
    ml.optimize(one, max_iters=3)                                   # allocations, attribute calls, the backward's transposed weights

    def run_k(k):
        torch.cuda.synchronize()
        t0 = time.time()
        ml.optimize(one, max_iters=k)
        torch.cuda.synchronize()
        return time.time() - t0
    K1, K2 = 12, 52
    t1, t2 = min(run_k(K1), run_k(K1)), min(run_k(K2), run_k(K2))      # (the first call with a new iteration count pays one-time set-up: the better of two)
    out['latent_optimisation_mode'] = {'workload': 'cfg %s with flag_opt_motion_latent / flag_opt_traj_latent, one %d-frame sequence' % (CFG_ID, NUM_FRAMES),
                                       'ms_per_iteration': round((t2 - t1) * 1e3 / (K2 - K1), 2), 'iterations': [K1, K2], 'seconds': [round(t1, 4), round(t2, 4)],
                                       'graph_replays_of_the_longer_run': int(getattr(ml, 'latent_graph_replays', 0)),
                                       'ms_per_iteration_round3_host_orchestrated': 20.3,
                                       'note': 'slope between a %d- and a %d-iteration run (init_data, the two plain iterations and the capture cancel): from the third iteration of '
                                               'a stage on the iteration -- taped infiller (10 windows), trajectory predictor, skinning, one gradient launch of the stage kernel, '
                                               'SMPL backward, infiller backward, two Adam steps with their step numbers on the device: 1 271 kernels (profiles/r06_latent_kernel_stats.csv) -- is ONE replayed HIP graph; '
                                               'round 6: the backward products of few rows on the one-wave split-fp16 kernel, the forward ones on gemm_small_kernel, the softmax of attention_bwd_kernel over four threads per row (12.8 -> 8.6 ms, profiles/r06_latent_ab.log)' % (K1, K2)}
    # ... and on a BATCH of sequences (round 5: the schedule takes S scenes; the reference runs the mode one sequence at a time)
    SB = 32
    many = [synth.make_in_dict(seed=s, num_frames=NUM_FRAMES, num_persons=1, smpl_model=md) for s in range(SB)]
    ml.optimize_batch(many, None, 3)

    def run_kb(k):
        torch.cuda.synchronize()
        t0 = time.time()
        ml.optimize_batch(many, None, k)
        torch.cuda.synchronize()
        return time.time() - t0
    KB1, KB2 = 6, 16
    tb1, tb2 = min(run_kb(KB1), run_kb(KB1)), min(run_kb(KB2), run_kb(KB2))
    msb = (tb2 - tb1) * 1e3 / (KB2 - KB1)
    out['latent_optimisation_mode']['batch_of_%d' % SB] = {'ms_per_iteration': round(msb, 2), 'ms_per_sequence_iteration': round(msb / SB, 3),
                                                          'iterations': [KB1, KB2], 'seconds': [round(tb1, 4), round(tb2, 4)]}
    return out


def one_sequence_latency(model, in_dict, reps=5):
    """BASELINE configs[1] is literally ONE 300-frame sequence: the latency of the reference's own call pattern, nothing to batch.  Median of
    `reps` runs of optimize(in_dict) (host dictionary in -> host dictionary out) and of optimize_resident on the staged inputs (HBM -> HBM)."""
    import statistics
    import torch
    model.optimize(in_dict)                                        # warm-up: allocations, attribute calls, the priors' graph for this geometry
    model.optimize(in_dict)
    host, hbm = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        model.optimize(in_dict)
        torch.cuda.synchronize()
        host.append((time.time() - t0) * 1e3)
    rin = model.stage_inputs([in_dict])
    for i in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.time()
        _, pk = model.optimize_resident(rin)
        torch.cuda.synchronize()
        if i:
            hbm.append((time.time() - t0) * 1e3)
    stage = [model.launch_ms(ws) for ws in pk.stage_ws]
    return {'host_dict_to_host_dict': round(statistics.median(host), 3), 'hbm_to_hbm': round(statistics.median(hbm), 3), 'stage_kernel': [round(x, 3) for x in stage],
            'sequences_per_sec_one_in_flight': round(1e3 / statistics.median(host), 1), 'reps': reps,
            'note': 'one 300-frame 1-person sequence alone on the GPU (the reference calls optimize() per sequence); `value` is the batched rate'}


def small_collective_latency(dev, world):
    """What the person-sharded variant of configs[3] would pay per iteration: an all-reduce of 9 floats + an all-gather of 4 x 300 x 12
    floats over RCCL, measured on this job's ranks (with one rank: the launch + completion cost of the two collectives, no link traffic)."""
    import torch
    import torch.distributed as dist
    g9 = torch.zeros(9, device=dev)
    own = torch.zeros(300 * 12, device=dev)
    allp = [torch.zeros(300 * 12, device=dev) for _ in range(world)]
    for _ in range(5):
        dist.all_reduce(g9)
        dist.all_gather(allp, own)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 200
    for _ in range(n):
        dist.all_reduce(g9)
        dist.all_gather(allp, own)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6


def person_sharded_line(asset_root, dev, rank, world, iters=20, scenes=8, group=None):
    """BASELINE configs[3] both ways on this job's ranks (SURVEY 8e "measure both and report"): `iters` iterations of every stage of cfg
    glamr_static_multi on `scenes` 4-person 300-frame scenes, (a) the default schedule -- one workgroup per scene, the shared camera's gradient
    reduced on chip, every rank its own scenes -- and (b) the person-sharded schedule (glamr_amd/parallel.py PersonShardedSchedule): the persons
    of each scene split over the ranks, per iteration 2 launches + an all-gather of the world poses + an all-reduce of the camera gradient over
    RCCL + the Adam launch.  Needs a process group (--gpus > 1, or --force-dist for a world of one rank)."""
    import torch
    from glamr_amd import parallel
    from glamr_amd.utils import synth
    m4 = build_model(asset_root, dev, 'glamr_static_multi')
    md = synth.make_smpl_model()
    in_dicts = [synth.make_in_dict(seed=2000 + i, num_frames=NUM_FRAMES, num_persons=4, smpl_model=md) for i in range(scenes)]
    rin = m4.stage_inputs(in_dicts)
    n_stage_iters = sum(min(iters, s['opt_niters']) for s in m4.opt_stage_specs.values())

    def fused():
        torch.manual_seed(4)                                           # the same latent draws for both variants
        _, packed = m4.init_resident(rin, init_forward=False)
        torch.cuda.synchronize()
        t0 = time.time()
        m4.run_schedule(packed, max_iters=iters)
        torch.cuda.synchronize()
        return time.time() - t0, packed

    def sharded():
        torch.manual_seed(4)
        _, packed = m4.init_resident(rin, init_forward=False)
        sched = parallel.PersonShardedSchedule(group=group)
        torch.cuda.synchronize()
        t0 = time.time()
        sched.run(packed, m4.opt_stage_specs, m4.specs, max_iters=iters)
        torch.cuda.synchronize()
        return time.time() - t0, packed, sched
    fused()
    t_f, pk_f = fused()
    sharded()
    t_s, pk_s, sched = sharded()
    diff = float((pk_f.t['kp_2d_pred'] - pk_s.t['kp_2d_pred']).abs().median())
    return {'workload': 'cfg glamr_static_multi, %d scenes of 4 persons x %d frames, %d iterations per stage (%d in all)' % (scenes, NUM_FRAMES, iters, n_stage_iters),
            'ranks': world, 'persons_per_rank': len(sched.owned(4)),
            'in_kernel_reduction_us_per_iteration': t_f / n_stage_iters * 1e6,
            'person_sharded_us_per_iteration': t_s / n_stage_iters * 1e6,
            'person_sharded_collective_host_us_per_iteration': sched.collective_seconds / n_stage_iters * 1e6,
            'launches_per_iteration': sched.launches / float(n_stage_iters),
            # iterations 2 .. n - 1 of a stage are replays of ONE captured graph (forward-only launch, all-gather, gradient launch, all-reduce, Adam):
            # their device time per iteration, without the two plain iterations and the capture a stage starts with (a real stage has 200 - 500)
            'person_sharded_replayed_us_per_iteration': (sum(a.elapsed_time(b) for a, b, _ in sched.replay_events) * 1e3 / max(1, sum(k for _, _, k in sched.replay_events))
                                                         if getattr(sched, 'replay_events', None) else None),
            'iteration_graphs': getattr(sched, 'iteration_graphs', 0),
            'median_projection_difference_px': diff,
            'note': 'per iteration the sharded form pays 2 stage launches (forward-only + gradient), an all-gather of 4 x 300 x 6 floats, an all-reduce of the '
                    'shared camera gradient and the Adam launch; the default keeps the scene in one workgroup for all iterations of a stage'}


HARD_EXIT = []      # reasons why this rank must leave without tearing its process group down (see `guarded`)


def guarded(fn, dev, seconds):
    """fn() on a worker thread with a wall-clock limit: (result, None) or (None, reason).  After a time-out the thread may still sit in a collective:
    the caller must not touch the process group again and leaves through os._exit."""
    import threading
    import torch
    box = {}

    def work():
        try:
            if dev.type == 'cuda':
                torch.cuda.set_device(dev)
            box['result'] = fn()
        except BaseException as e:      # noqa: BLE001 -- reported by the caller; the headline must not depend on this line
            box['error'] = '%s: %s' % (type(e).__name__, e)
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return None, 'no result after %.0f s' % seconds
    if 'error' in box:
        return None, box['error']
    return box['result'], None


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: this file re-executed under torch.distributed.run (glamr_amd.parallel.self_launch); rank 0's
    JSON line is the only line on stdout."""
    from glamr_amd import parallel
    return parallel.self_launch(n, [os.path.abspath(__file__)], argv, keep_on_stdout=lambda line: line.lstrip().startswith('{"metric"'))


def strong_line(model, md, rank, world, dev, total, steps, sync, use_dist):
    """BASELINE configs[2] in the SAME run as the weak line: a fixed job of `total` (64) independent 300-frame sequences split over the ranks in
    balanced contiguous blocks (parallel.shard_range), no data-path collective; `steps` passes, barrier + synchronize on both sides, max over
    ranks.  It cannot scale and is not meant to: 64 scenes occupy 64 of one GPU's 256 CUs for about ONE sequence latency, so N GPUs finish
    64 / N scenes each in the same time -- the figure exists to show exactly that next to the weak line."""
    import torch.distributed as dist
    from glamr_amd import parallel
    from glamr_amd.utils import synth
    lo, hi = parallel.shard_range(total, rank, world)
    in_dicts = [synth.make_in_dict(seed=sd, num_frames=NUM_FRAMES, num_persons=1, smpl_model=md) for sd in range(lo, hi)]
    rin = model.stage_inputs(in_dicts) if in_dicts else None
    run_once = (lambda: model.optimize_resident(rin)) if in_dicts else (lambda: None)
    run_once()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.time()
    for _ in range(steps):
        run_once()
    sync()
    if use_dist:
        dist.barrier()
    elapsed = parallel.max_over_ranks(time.time() - t0, dev)
    n = parallel.sum_over_ranks(hi - lo, dev)
    return {'workload': 'BASELINE configs[2]: %d independent %d-frame 1-person sequences (cfg %s) split over %d rank(s), %d on rank 0' % (total, NUM_FRAMES, CFG_ID, world, hi - lo),
            'scaling': 'strong', 'sequences_total': n, 'steps': steps, 'ms_per_step': elapsed / steps * 1e3, 'sequences_per_sec': n * steps / elapsed,
            'note': 'one plain (ungraphed, single-stream) optimize_resident per step; a GPU holds one 300-frame scene per CU, so %d scenes use %d of its CUs '
                    'for about one sequence latency -- more GPUs cannot shorten that: the weak line is the scaling figure' % (hi - lo, hi - lo)}


class _StubModel:
    """CPU stand-in used by tests/test_parallel_gloo.py to run this file's distributed skeleton (init, asset barrier, seed partition,
    timing protocol, max-reduction, rank-0 JSON) over gloo without a GPU.  Never used by a real measurement."""
    opt_stage_specs = {'init_opt': {'opt_niters': 500}}
    timings = {}

    def stage_inputs(self, in_dicts):
        return list(in_dicts)

    def optimize_resident(self, rin):
        time.sleep(0.002 * len(rin))

        class P:
            stage_ws = []
        return None, P()


def run(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='independent sequences per GPU per step (weak scaling)')
    ap.add_argument('--mode', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--total', type=int, default=64, help='--mode strong: sequences in the whole job (BASELINE configs[2]: 64)')
    ap.add_argument('--streams', type=int, default=2, help='HIP streams the steps alternate over')
    ap.add_argument('--no-host-stream', action='store_true', help='skip the optimize_stream() measurement (profiler runs of ONE stream: its batches are co-scheduled and would mix into the kernel statistics)')
    ap.add_argument('--no-coschedule', action='store_true', help='two streams left to themselves and the LDS kernels for the priors (the round-2 / early round-3 pipeline)')
    ap.add_argument('--no-graph-step', action='store_true', help='do NOT capture a whole step per stream as one HIP graph after the warm-up (default: capture, check the replay '
                    'against a plain step bit for bit, replay it in the timed region -- a host whose driver calls are slow then no longer paces the ~25 launches of '
                    'a step; any failure of capture or check falls back to plain launches)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-lines', action='store_true')
    ap.add_argument('--no-strong-line', action='store_true', help='skip the BASELINE configs[2] line (64 sequences split over the ranks) measured after the weak one')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo with --stub-model for the CPU test of the skeleton)')
    ap.add_argument('--stub-model', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-early-prep', action='store_true', help='capture the pipelined step with the library\'s default cut (the whole batch behind the gate: two graphs per step)')
    ap.add_argument('--force-dist', action='store_true', help='create the process group also at --gpus 1 (a world of one rank): runs the init / barrier / reduction / '
                    'collective-latency code of the multi-GPU path on a single GPU (tests/test_e2e_gpu.py)')
    args = ap.parse_args(argv)

    if args.gpus > 1 and 'RANK' not in os.environ:
        # launched plainly (`python bench.py --gpus N`): spawn the N ranks ourselves, one process per GPU, exactly as the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` command line would; rank 0's JSON line is this process's
        raise SystemExit(self_launch(args.gpus, list(sys.argv[1:] if argv is None else argv)))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d inside a job of WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or plainly (no RANK in the '
                         'environment) and bench.py spawns its own ranks' % (args.gpus, world, args.gpus))
    on_gpu = not args.stub_model
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank) if on_gpu else torch.device('cpu')
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from glamr_amd import parallel
    from glamr_amd.utils import synth
    # synthetic model files are written ONCE: rank 0 creates them, everybody else waits at the barrier and then only reads
    if rank == 0:
        ensure_assets()
    if use_dist:
        dist.barrier()
    asset_root = ensure_assets()
    model = _StubModel() if args.stub_model else build_model(asset_root, dev)
    md = synth.make_smpl_model()
    if args.mode == 'weak':
        seeds = parallel.weak_scaling_seeds(args.batch, rank)      # every rank its own B sequences: seeds rank*B .. rank*B + B - 1
    else:
        seeds = list(range(*parallel.shard_range(args.total, rank, world)))       # a fixed job split over the ranks
    B = len(seeds)
    in_dicts = [synth.make_in_dict(seed=sd, num_frames=NUM_FRAMES, num_persons=1, smpl_model=md) for sd in seeds]

    rin = model.stage_inputs(in_dicts)                             # HybrIK arrays resident in HBM before the clock starts
    stage_events = []
    # consecutive steps are independent batches: they are enqueued on alternating HIP streams, so the launch seams and tails of one
    # batch are covered by the next one (--streams 1 serialises them)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))] if on_gpu else [None]
    sync()                                                         # the upload ran on the default stream; the side streams do not wait for it
    keep = []
    # two streams: batches are STAGGERED (a batch starts when the previous one's priors have finished) and the infiller runs on the kernels
    # that fit beside a resident stage workgroup -- product code: GlobalReconOptimizer.pipeline_gate, also what optimize_stream() does
    coschedule = on_gpu and len(streams) >= 2 and not args.no_coschedule and not args.stub_model
    if coschedule:
        from glamr_amd.global_recon.models.global_recon_model import PipelineGate, coschedule_enabled
        coschedule = coschedule_enabled()
        if coschedule:
            model.pipeline_gate = PipelineGate()

    def step(i=0):
        if on_gpu:
            with torch.cuda.stream(streams[i % len(streams)]):
                _, packed = model.optimize_resident(rin)
        else:
            _, packed = model.optimize_resident(rin)
        stage_events.append(packed.stage_ws)
        keep.append(packed)                                        # results of the timed steps stay resident until the clock stops
        if len(keep) > 2 * len(streams):
            keep.pop(0)

    def replay_check_of(graphs, pairs=4):
        """The streams' step graphs replayed alternately as in the timed loop, under TWO generator states that swap between the graphs from pair to
        pair (torch's graphs take seed and offset at replay time), with the output arrays poisoned before every pair: each replay must leave in its
        graph's arrays exactly what ONE plain step with that state leaves -- projections, every optimised parameter, cached joints, camera -- bit
        for bit.  A value read from the other stream, left over from the previous replay, or computed wrong beside the other stream's kernels is a
        difference here and nowhere in a timing (round 6: profiles/r06_pipeline_corruption.log)."""
        keys = ('kp_2d_pred', 'params', 'j_local', 'cam_pose')
        with torch.random.fork_rng(devices=[dev]):
            seeds = (20260927, 20260928)
            want = {}
            for seed in seeds:
                torch.manual_seed(seed)
                with torch.cuda.stream(streams[0]):
                    _, ref = model.optimize_resident(rin)
                sync()
                want[seed] = {k: ref.t[k].clone() for k in keys}
            gate_now = getattr(model, 'pipeline_gate', None)
            if gate_now is not None:
                gate_now.last = None
            worst, equal, n_rep = 0.0, True, 0
            for pair in range(pairs):
                for g in graphs:
                    for k in keys:
                        g.packed.t[k].fill_(float('nan'))
                sync()
                used = {}
                for i, g in enumerate(graphs if pair % 2 == 0 else graphs[::-1]):
                    used[id(g)] = seeds[(pair + i) % 2]
                    torch.manual_seed(used[id(g)])
                    g.replay()
                    n_rep += 1
                sync()
                for g in graphs:
                    for k in keys:
                        got, w = g.packed.t[k], want[used[id(g)]][k]
                        equal = equal and bool(torch.isfinite(got).all()) and torch.equal(got, w)
                    d = (g.packed.t['kp_2d_pred'] - want[used[id(g)]]['kp_2d_pred']).abs().max()
                    worst = max(worst, float(d) if bool(torch.isfinite(d)) else float('inf'))
        return {'replays': n_rep, 'streams': len(graphs), 'seeds': len(seeds), 'arrays': list(keys), 'bit_identical_to_plain_steps': equal, 'max_projection_difference_px': worst}

    gate_cut = None
    graph_step = on_gpu and not args.no_graph_step
    for i in range(max(args.warmup, len(streams) if graph_step else 0)):
        step(i)
    sync()
    step_graphs = None
    if graph_step:
        # every stream has run the step once (allocations, attribute calls, the priors' own graph): capture it, one graph per stream, through
        # the product's own entry point (GlobalReconOptimizer.capture_resident: capture + bit-for-bit check of a replay against a plain step)
        try:
            # The library's default cut under a gate is preparation | priors + skinning | rest (three graphs: the preparation does not wait for the
            # gate); --no-early-prep times the two-graph cut.  The streams' replays must reproduce plain steps bit for bit right here, before the
            # clock starts, and again after it: a failed check is an ERROR (exit code 1), not a warning.
            if coschedule and args.no_early_prep:
                os.environ['GLAMR_GATE_PREP'] = 'late'
            gate_cut = None if not coschedule else ('two graphs (--no-early-prep)' if os.environ.get('GLAMR_GATE_PREP', 'early') == 'late'
                                                    else 'preparation ahead of the gate (three graphs, the library default)')
            step_graphs = [model.capture_resident(rin, stream=st, check=True) for st in streams]
            sync()
            pre = replay_check_of(step_graphs, pairs=2)
            if not pre['bit_identical_to_plain_steps']:
                raise SystemExit('bench: the step graphs do not reproduce plain steps BEFORE the timed region (max %.3g px)' % pre['max_projection_difference_px'])
            sync()
            eager_step = step

            def step(i=0):
                step_graphs[i % len(step_graphs)].replay()
                # (no per-launch stamps here: a replay rewrites the ONE workspace its graph was captured with, so the stamps of all but
                # the last replay are gone by the time the clock stops -- the roofline launches are measured right after, see below)
        except Exception as e:      # noqa: BLE001 -- anything: the plain launches are always available
            sys.stderr.write('bench: step graph not used (%s); plain launches\n' % e)
            step_graphs = None
            try:
                sync()
            except Exception:      # noqa: BLE001
                pass
    del stage_events[:]
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.time()
    for i in range(args.steps):
        step(i)
    enqueue = time.time() - t0                                       # host time to ENQUEUE the timed steps (launch-bound if close to `elapsed`)
    sync()
    if use_dist:
        dist.barrier()
    elapsed = time.time() - t0
    # What the timed region computed, checked after the clock has stopped (replay_check_of above): a difference is an error, no number is printed
    replay_check = None
    if step_graphs and on_gpu:
        replay_check = replay_check_of(step_graphs)
        replay_check['gate_cut'] = gate_cut
        if not replay_check['bit_identical_to_plain_steps']:
            raise SystemExit('bench: the pipelined replays do NOT reproduce plain steps (max %.3g px): no number is reported for a timing of wrong results'
                             % replay_check['max_projection_difference_px'])
    beside = None
    if coschedule:
        # what a stage launch takes in THIS pipeline (beside the other stream's priors): four more gated steps as plain launches, whose
        # workspaces carry the kernel's own clock stamps (graph replays share one workspace per graph) -- after the clock has stopped
        try:
            del stage_events[:]
            for i in range(6):
                (eager_step if step_graphs else step)(i)
            sync()
            beside = [round(model.launch_ms(ws), 2) for wss in stage_events[2:5] for ws in wss]      # (the last one has no neighbour)
        except Exception as e:      # noqa: BLE001
            sys.stderr.write('bench: co-scheduled stage launches not measured (%s)\n' % e)
        del stage_events[:]
        model.pipeline_gate = None                                   # the single-stream measurements below run the plain step
    if coschedule and args.no_early_prep:
        os.environ.pop('GLAMR_GATE_PREP', None)                      # (set above for this pipeline only: a caller in the same process keeps its own setting)
    elapsed = parallel.max_over_ranks(elapsed, dev)
    n_total = parallel.sum_over_ranks(B, dev)                      # units all ranks processed per step
    rccl_ranks = dist.get_world_size() if (use_dist and dist.get_backend() == 'nccl') else None
    strong = None
    if args.mode == 'weak' and not args.no_strong_line:
        gate_keep, model.pipeline_gate = getattr(model, 'pipeline_gate', None), None
        strong = strong_line(model, md, rank, world, dev, args.total, max(2, min(5, args.steps)), sync, use_dist)
        model.pipeline_gate = gate_keep
    if args.stub_model:
        out = {'metric': 'sequences/sec (300-frame, 1-person) end-to-end global_recon', 'value': n_total * args.steps / elapsed, 'unit': 'sequences/sec',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
               'scaling': args.mode, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'config': {'workload': 'STUB (CPU test of the skeleton)', 'sequences_total': n_total,
                                                                                                          'seeds_first_last': [int(seeds[0]), int(seeds[-1])]},
               'rccl_ranks': rccl_ranks, 'process_group_ranks': dist.get_world_size() if use_dist else None, 'configs2_strong_64': strong}
        if rank == 0:
            print(json.dumps(out))
        if use_dist:
            dist.destroy_process_group()
        return out if rank == 0 else None

    # one entry per optimiser-stage launch in the timed region: the kernel's own clock (earliest workgroup start to latest workgroup
    # end, what rocprofv3 reports for the dispatch) -- HIP events around the launch would also count the time it waits behind the
    # other stream
    kms_timed = [model.launch_ms(ws) for wss in stage_events for ws in wss] if not step_graphs else None
    # With more than one stream the stage launches of consecutive batches share the CUs, so a launch's span says nothing about the
    # kernel: the roofline figure comes from launches that have the GPU to themselves -- the timed ones when --streams 1, otherwise
    # two extra single-stream steps right after the timed region.
    if len(streams) == 1 and kms_timed:
        kms = kms_timed
    else:
        kms = []
        for _ in range(2):
            with torch.cuda.stream(streams[0]):
                _, packed = model.optimize_resident(rin)
            torch.cuda.synchronize()
            kms.extend(model.launch_ms(ws) for ws in packed.stage_ws)
    # host dictionaries in -> host dictionaries out, for the record (never `value`): one call, and the pipelined stream of batches
    # (the FIRST call on a stream nothing ran on before pays hipMalloc for the whole working set -- the caching allocator's pools are per stream
    # -- and page-locks a 353 MB staging set: reported apart; `single_call` is the call after it, what a caller looping over batches sees)
    t0 = time.time()
    model.optimize_batch(in_dicts)
    host_first = time.time() - t0
    model.optimize_batch(in_dicts)
    t0 = time.time()
    model.optimize_batch(in_dicts)
    host_single = time.time() - t0
    tm = dict(model.timings)
    host_stream = None
    if hasattr(model, 'optimize_stream') and not args.no_host_stream:
        # the pipelined stream alternates batches over two compute streams: the two this run already has (the runtime multiplexes HIP streams
        # onto 4 hardware queues; with two more of its own the stream's batches shared queues with idle ones: 15 700 instead of 22 000 seq/s)
        if len(streams) == 2:
            model.__dict__.setdefault('_compute_streams', list(streams))
        # steady state of the pipelined stream: the first pass creates the three pinned output sets a depth-3 pipeline holds (page-locking
        # 400 MB each: a one-off ~100 ms the 6-batch figure of rounds 1-2 carried); the timed pass re-uses them
        for _ in model.optimize_stream([in_dicts] * 4):
            pass
        nb = 24                                                          # (fill and drain of the three-deep pipeline are inside the figure)
        t0 = time.time()
        n_out = sum(len(r) for r in model.optimize_stream([in_dicts] * nb))
        host_stream = n_out / (time.time() - t0)
    coll_us = small_collective_latency(dev, world) if use_dist else None
    latency = one_sequence_latency(model, in_dicts[0]) if (rank == 0 and not args.stub_model and not args.no_kernel_lines and hasattr(model, 'optimize')) else None      # (--no-kernel-lines: profiler runs see the batched launches only)
    sharded4 = None
    if not args.stub_model and not args.no_kernel_lines:
        if use_dist:
            # 4 persons: at most 4 ranks take part (a sub-group of the first four on a larger job; every rank creates the group)
            n_sh = min(world, 4)
            grp = dist.new_group(list(range(n_sh))) if world > n_sh else None
            if rank < n_sh:
                # This is the ONE part of the run whose collectives sit on the data path, and no multi-GPU box was ever available to rehearse it:
                # it runs under a wall-clock guard on every rank.  A rank whose guard expires (a peer failed and the collectives wait for it) or
                # whose call raised gives up the line; rank 0 then prints the headline without it and every rank leaves through os._exit
                # (a process group with a collective in flight cannot be torn down).
                sharded4, abandoned = guarded(lambda: person_sharded_line(asset_root, dev, rank, n_sh, group=grp), dev,
                                              float(os.environ.get('GLAMR_BENCH_SHARDED_TIMEOUT', '240')))
                if abandoned:
                    sys.stderr.write('bench: person-sharded line abandoned on rank %d (%s)\n' % (rank, abandoned))
                    HARD_EXIT.append(abandoned)
        elif world == 1 and on_gpu:
            # a single-GPU run has no process group: one of ONE rank is created for this line alone, so that the RCCL calls of the
            # person-sharded schedule execute here too (nothing crosses a link: the figure is the launch + completion cost per iteration)
            try:
                import socket
                sk = socket.socket()
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
                sk.close()
                os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
                dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
                try:
                    sharded4 = person_sharded_line(asset_root, dev, 0, 1)
                finally:
                    dist.destroy_process_group()
            except Exception as e:      # noqa: BLE001 -- the headline must not depend on this line
                sys.stderr.write('bench: person-sharded line skipped (%s)\n' % e)
                sharded4 = None

    out = None
    if rank == 0:
        iters = sum(s['opt_niters'] for s in model.opt_stage_specs.values())
        k_avg = sum(kms) / max(1, len(kms))
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        rounds = -(-B // n_cus)                                      # one 300-frame scene per CU at a time
        us_scene_iter = k_avg * 1e3 / iters / rounds
        algo_bytes = B * iters * ALGO_BYTES_PER_PERSON_ITER
        achieved = algo_bytes / (k_avg * 1e-3) / 1e9 if k_avg > 0 else 0.0
        traffic, traffic_src, issue_util, over_compulsory, bytes_si, valu_issue = None, 'no profiles/rNN_pmc_stage_kernel.json', None, None, None, None
        pf = pmc_file()
        if pf:
            pmc = json.load(open(pf))
            bytes_si = pmc['bytes_per_scene_iteration']
            traffic = bytes_si * B * iters
            traffic_src = os.path.relpath(pf, ROOT) + ': ' + pmc['source']
            # compulsory = the live state of a scene read or written ONCE per launch, per iteration
            over_compulsory = bytes_si / (ALGO_BYTES_PER_PERSON_ITER / float(iters))
            sq = pmc.get('counters', {}).get('sq', {})
            if 'SQ_ACTIVE_INST_ANY' in sq and 'SQ_WAVE_CYCLES' in sq:
                # five waves of a 300-frame scene on the four SIMDs of its CU: issue slots available = 4/5 of the summed wave cycles
                issue_util = sq['SQ_ACTIVE_INST_ANY']['mean'] / (0.8 * sq['SQ_WAVE_CYCLES']['mean'])
            inst = pmc.get('counters', {}).get('inst', {})
            if 'SQ_INSTS_VALU' in inst:
                # second yardstick: pure VALU issue time -- a wave64 VALU instruction occupies its SIMD for 2 cycles
                # (MI355X_MICROARCH.md); one wave alone, and the SIMD that carries two of a 300-frame scene's five waves
                waves = 5 * pmc.get('scenes', B)
                valu_wave_iter = inst['SQ_INSTS_VALU']['mean'] / (waves * pmc.get('iterations', iters))
                clock_hz = 2.4e9
                valu_issue = {'valu_per_wave_iteration': valu_wave_iter, 'cycles_per_valu': 2, 'clock_ghz': clock_hz / 1e9,
                              'one_wave_us': valu_wave_iter * 2 / clock_hz * 1e6, 'busiest_simd_us': 2 * valu_wave_iter * 2 / clock_hz * 1e6}
                valu_issue['frac_of_one_wave'] = valu_issue['one_wave_us'] / us_scene_iter if us_scene_iter > 0 else None
                valu_issue['frac_of_busiest_simd'] = valu_issue['busiest_simd_us'] / us_scene_iter if us_scene_iter > 0 else None
        out = {
            'metric': 'sequences/sec (300-frame, 1-person) end-to-end global_recon', 'value': n_total * args.steps / elapsed,
            'unit': 'sequences/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.mode, 'vs_baseline': None,
            'dtype': 'f32 (optimiser, init_data) / fp16x2-split MFMA with f32 accumulation (priors, LBS: every f32 operand as hi + lo fp16 planes, hi*hi + hi*lo + lo*hi)',
            'data': 'synthetic', 'host_enqueue_ms_per_step': enqueue / args.steps * 1e3,
            'rccl_ranks': rccl_ranks, 'process_group_ranks': dist.get_world_size() if use_dist else None,
            'config': {'workload': ('BASELINE configs[1]: %d-frame 1-person dynamic-camera sequences, cfg %s (%d Adam iterations), batch of %d independent '
                                    'sequences per GPU, HybrIK arrays resident in HBM%s' % (NUM_FRAMES, CFG_ID, iters, B, '; the step is replayed as one captured HIP graph per stream (bit-checked against plain launches)' if step_graphs else '')) if args.mode == 'weak' else
                                   ('BASELINE configs[2]: %d independent %d-frame 1-person sequences (cfg %s, %d iterations) split over %d GPU(s): %d per GPU '
                                    '= %d of %d CUs busy in the optimiser stage' % (args.total, NUM_FRAMES, CFG_ID, iters, world, B, min(B, n_cus), n_cus)),
                       'sequences_per_gpu': B, 'frames': NUM_FRAMES, 'persons': 1, 'parallelism': 'sequence-sharded x%d' % world,
                       'streams_per_gpu': len(streams), 'step_graph': bool(step_graphs), 'coscheduled_streams': bool(coschedule)},
            'roofline': {'kernel': 'grecon_stage_kernel<1,true,1,304>', 'bound': 'latency/issue',
                         'bound_note': 'one workgroup per scene, state on chip, 7 workgroup barriers per iteration; neither HBM nor the matrix pipes limit it.  Round 5 removed the exposed '
                                       'memory waits of the dependent chain (13.3 -> 10.6 us per scene-iteration, profiles/r05_stage_ab.log); what bounds it now is ISSUE on the SIMD that '
                                       'carries two of a 300-frame scene\'s five waves: the fifth wave leaves the keypoint phase 1.1 us after the other four, which wait for it at the '
                                       'next barrier (profiles/r05_phase_times.log).  Round 6: 43 instead of 52 operations per scored joint (10.83 -> 10.5 us, profiles/r06_stage_ab.log); interleaved '
                                       'joint chains (max-ILP scheduling), staged Adam chains and two scenes per CU measured without gain (profiles/r06_session2_probes.log). '
                                       'achieved / peak / frac are SURVEY 8(d) K5\'s yardstick: microseconds per scene-iteration against the 1.45 us dependent-boundary floor '
                                       "(frac = floor / achieved).  The contract's hbm line is kept under `contract_notional` and is NOTIONAL",
                         'achieved': us_scene_iter, 'peak': DEPENDENT_BOUNDARY_US, 'unit': 'us per scene-iteration (lower is better; peak = dependent-boundary floor)',
                         'frac': DEPENDENT_BOUNDARY_US / us_scene_iter if us_scene_iter > 0 else None,
                         'us_per_scene_iteration': us_scene_iter, 'dependent_boundary_floor_us': DEPENDENT_BOUNDARY_US,
                         'times_above_floor': us_scene_iter / DEPENDENT_BOUNDARY_US,
                         'issue_slot_utilisation': issue_util, 'valu_issue_yardstick': valu_issue, 'traffic_over_compulsory': over_compulsory, 'traffic_bytes_per_scene_iteration': bytes_si,
                         'compulsory_bytes_per_scene_iteration': ALGO_BYTES_PER_PERSON_ITER / float(iters),
                         'contract_notional': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'contract_notional_frac': achieved / HBM_PEAK_GBS,
                                               'note': '%d B per person-iteration (SURVEY.md 8d K5 live state) x %d scenes x %d iterations / launch time: bytes the '
                                                       'kernel keeps in LDS and does not move' % (ALGO_BYTES_PER_PERSON_ITER, B, iters)},
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'traffic_gbs': None if traffic is None or k_avg <= 0 else traffic / (k_avg * 1e-3) / 1e9,
                         'traffic_frac_of_hbm_peak': None if traffic is None or k_avg <= 0 else traffic / (k_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'avg_launch_ms': k_avg, 'scenes_per_launch': B, 'iterations_per_launch': iters, 'rounds_per_launch': rounds,
                         'launch_ms_each': [round(x, 2) for x in kms],
                         'launch_ms_in_timed_region': None if kms_timed is None else [round(x, 2) for x in kms_timed],
                         'measured': ('the launches of the timed region (one stream)' if kms is kms_timed else
                                      '2 single-stream plain steps right after the timed region: in the timed region the launches of the %d stream(s) '
                                      'share the CUs%s' % (len(streams), ' and are graph replays (one workspace per graph: no per-replay stamps)' if step_graphs else '')),
                         'note': 'traffic = memory-side bytes of one launch (rocprofv3 PMC passes on the shipped instance)'},
            'replay_check': replay_check,
            'pipeline': None if not coschedule else {
                'coscheduled_streams': True, 'stage_launch_ms_alone': k_avg, 'stage_launch_ms_beside_the_priors': beside,
                'critical_cycle': 'per batch and stream -- preparation (ahead of the gate: three-graph cut), infiller I beside the other batch\'s stage (~27 ms), then, when '
                                  'that stage has retired, skinning (1.4 ms) and trajectory predictor P (4.4 ms, LDS kernels), gate, scene assembly + forward-only launch '
                                  '(0.3 ms), stage S (~27.7 ms beside the next infiller; 21.7 alone).  I and S end together; P + skinning + assembly (6.5 ms) are serial: '
                                  'profiles/r06_gap_trace.log.  The skinning sits before the gate because it is faster alone (34.7 against 35.4 ms per step), not to avoid '
                                  'anything: the corruption rounds 4 - 5 saw there was packed-fp32 arithmetic going wrong beside the other stream\'s MFMA kernels, removed at '
                                  'its root in round 6 (profiles/r06_pipeline_corruption.log, glamr_amd/build.py)',
                'note': 'two streams, batches staggered by GlobalReconOptimizer.pipeline_gate: a batch starts when the previous one\'s priors are done, so its '
                        'motion infiller -- LDS-free one-wave kernels on fragment-major activations (csrc/nn_free.hpp) -- runs in the SIMD issue slots and '
                        'matrix pipes a resident stage workgroup leaves idle.  The stage launch is slower beside them than alone; the step is shorter '
                        '(--no-coschedule: the LDS kernels, streams left to themselves).  `roofline` describes the stage kernel ALONE'},
            'latency_one_sequence_ms': latency,
            'host_inclusive_sequences_per_sec': host_stream if host_stream is not None else B / host_single,
            'host_inclusive_single_call_sequences_per_sec': B / host_single,
            'host_inclusive_first_call_on_a_new_stream_sequences_per_sec': B / host_first,
            'host_inclusive_stage_seconds': {k: round(v, 4) for k, v in tm.items()},
        }
        if strong is not None:
            out['configs2_strong_64'] = strong
        if sharded4 is not None:
            out['configs3_person_sharded'] = sharded4
        if coll_us is not None:
            out['collective_alternative'] = {'us_per_iteration_allreduce9_plus_allgather_4x300x12': coll_us, 'ranks': world,
                                             'note': 'what a person-sharded 4-person scene would add to EVERY iteration (loss_func.py:255-268, '
                                                     'global_recon_model.py:597-601) against the in-kernel reduction of `kernels.configs3`'}
        if world == 1 and not args.no_kernel_lines and args.mode == 'weak':
            out['kernels'] = kernel_lines(asset_root, model, dev)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(asset_root)
        print(json.dumps(out))
    if HARD_EXIT:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if use_dist:
        dist.destroy_process_group()
    return out


def main():
    run()
    # ONE JSON line on stdout: RCCL prints a version banner to stdout when the process exits (after the line above) -- whatever is
    # written from here on goes to stderr
    sys.stdout.flush()
    os.dup2(2, 1)


if __name__ == '__main__':
    main()
