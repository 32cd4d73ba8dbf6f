"""Headline benchmark: sequences/sec of end-to-end global reconstruction (GlobalReconOptimizer.optimize_batch) on 300-frame,
1-person, dynamic-camera synthetic sequences (BASELINE.json configs[1], cfg `glamr_dynamic`, 500 Adam iterations).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" = one pass of the hot path over one batch of B independent synthetic sequences per GPU: per-person preparation of the
HybrIK arrays, motion infilling + trajectory prediction, SMPL skinning, scene initialisation and the full optimisation schedule
(GlobalReconOptimizer.optimize_resident).  The HybrIK arrays are uploaded once before the timed region (stage_inputs), so `value` is
the HBM-in / HBM-out rate the contract asks for; the rate from HOST dictionaries to HOST dictionaries (optimize_batch: numpy
scatter + PCIe both ways + building the reference's output dictionaries) is measured after it and reported as
`host_inclusive_sequences_per_sec`.  Multi-GPU: one process per GPU (torch.distributed over RCCL), sequences are independent, so
ranks share nothing on the data path (weak scaling); the only collectives are the barrier and the max-reduction of the elapsed time.

Consecutive steps alternate over `--streams` HIP streams (default 2).  Kernel durations come from the kernel's own clock (stamps in
the workspace header), and the roofline launch is measured with the GPU to itself (see main()).

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (the fused optimiser stage), `cpu_baseline` the CPU
oracle (a port of the reference, oracle/port) timed on a bounded sample on this box's host cores.
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
# HybrIK orientation/translation 7 KB, transforms 60 KB
ALGO_BYTES_PER_PERSON_ITER = (2 * 58 + 94 + 94 + 11 + 13 + 7 + 60) * 1024
HBM_PEAK_GBS = 8000.0
# memory-side traffic of the stage kernel, measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on this workload
# (profiles/r01_pmc_stage_kernel_b1024.csv: 21.1e6 KB fetched, 31.1e6 KB written by the launch of 1024 scenes x 500 iterations) and
# corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request: doubled); per scene-iteration
TRAFFIC_BYTES_PER_SCENE_ITER = (2 * 21.1e6 + 31.1e6) * 1024 / (1024 * 500)
CFG_ID, NUM_FRAMES = 'glamr_dynamic', 300


def build_model(asset_root, device):
    import torch
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.models.prior_models import MotionTrajJointModel
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(device)
    mt = MotionTrajJointModel(None, device, None, smpl=smpl, results_root=os.path.join(asset_root, 'results'))
    return model_dict['global_recon_model'](get_config(CFG_ID), device, None, smpl=smpl, mt_model=mt)


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


def cpu_baseline(asset_root, iters=12):
    """The CPU oracle (oracle/port: torch autograd + Adam, full SMPL skinning per iteration like the reference) on ONE 300-frame
    sequence: init_data in full, then `iters` of the 500 iterations timed at several intra-op thread counts; the fastest is
    extrapolated to the schedule (about 20 s of CPU work in total)."""
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
    best = None
    for nt in sorted({n for n in (4, 8, 16, 32) if n <= ncpu} | {min(ncpu, 64)}):
        torch.set_num_threads(nt)
        opt.optimize_main(data, spec['opt_variables'], spec['opt_lr'], 2, spec['loss_cfg'], {'stage': 'init_opt'})          # warm-up
        t0 = time.time()
        opt.optimize_main(data, spec['opt_variables'], spec['opt_lr'], iters, spec['loss_cfg'], {'stage': 'init_opt'})
        per_iter = (time.time() - t0) / iters
        if best is None or per_iter < best[1]:
            best = (nt, per_iter)
    nt, per_iter = best
    total = t_init + per_iter * spec['opt_niters']
    return {'value': 1.0 / total, 'unit': 'sequences/sec', 'cores': nt, 'kind': 'port',
            'sample': 'oracle/port on 1 sequence of %d frames: init_data (%.2f s) + %d of %d Adam iterations timed at 4..64 threads, best = %d threads '
                      '(%.1f ms/iter), extrapolated to the full schedule (%.1f s/sequence)' % (NUM_FRAMES, t_init, iters, spec['opt_niters'], nt, per_iter * 1e3, total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=1024, help='independent sequences per GPU per step')
    ap.add_argument('--streams', type=int, default=2, help='HIP streams the steps alternate over')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    from glamr_amd import parallel
    from glamr_amd.utils import synth
    if rank == 0:
        asset_root = ensure_assets()
    if world > 1:
        dist.barrier()
    asset_root = ensure_assets()
    model = build_model(asset_root, dev)
    md = synth.make_smpl_model()
    B = args.batch
    # every rank works on its own sequences: seeds rank*B .. rank*B + B - 1 (independent units, no data-path collective)
    in_dicts = [synth.make_in_dict(seed=sd, num_frames=NUM_FRAMES, num_persons=1, smpl_model=md) for sd in parallel.weak_scaling_seeds(B, rank)]

    rin = model.stage_inputs(in_dicts)                             # HybrIK arrays resident in HBM before the clock starts
    stage_events = []

    # consecutive steps are independent batches: they are enqueued on alternating HIP streams, so the matrix-core-bound prior
    # networks of one batch can run under the latency-bound optimiser stage of the previous one (--streams 1 serialises them)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    torch.cuda.synchronize()                                       # the upload ran on the default stream; the side streams do not wait for it
    keep = []

    def step(i=0):
        with torch.cuda.stream(streams[i % len(streams)]):
            _, packed = model.optimize_resident(rin)
        stage_events.append(packed.stage_ws)
        keep.append(packed)                                        # results of the timed steps stay resident until the clock stops
        if len(keep) > 2 * len(streams):
            keep.pop(0)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    del stage_events[:]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.time() - t0
    elapsed = parallel.max_over_ranks(elapsed, dev)
    # one entry per optimiser-stage launch in the timed region: the kernel's own clock (earliest workgroup start to latest workgroup
    # end, what rocprofv3 reports for the dispatch) -- HIP events around the launch would also count the time it waits behind the
    # other stream
    kms_timed = [model.launch_ms(ws) for wss in stage_events for ws in wss]
    # With more than one stream the stage launches of consecutive batches share the CUs, so a launch's span says nothing about the
    # kernel: the roofline figure comes from launches that have the GPU to themselves -- the timed ones when --streams 1, otherwise
    # two extra single-stream steps right after the timed region.
    if len(streams) == 1:
        kms = kms_timed
    else:
        kms = []
        for _ in range(2):
            with torch.cuda.stream(streams[0]):
                _, packed = model.optimize_resident(rin)
            torch.cuda.synchronize()
            kms.extend(model.launch_ms(ws) for ws in packed.stage_ws)
    # host dictionaries in -> host dictionaries out, for the record (never `value`)
    n_host = min(2, args.steps)
    t0 = time.time()
    for _ in range(n_host):
        model.optimize_batch(in_dicts)
    host_elapsed = (time.time() - t0) / n_host
    tm = dict(model.timings)

    if rank == 0:
        iters = sum(s['opt_niters'] for s in model.opt_stage_specs.values())
        k_avg = sum(kms) / max(1, len(kms))
        algo_bytes = B * iters * ALGO_BYTES_PER_PERSON_ITER
        achieved = algo_bytes / (k_avg * 1e-3) / 1e9 if k_avg > 0 else 0.0
        out = {
            'metric': 'sequences/sec (300-frame, 1-person) end-to-end global_recon', 'value': B * world * args.steps / elapsed,
            'unit': 'sequences/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: %d-frame 1-person dynamic-camera sequences, cfg %s (%d Adam iterations), '
                                   'batch of %d independent sequences per GPU, HybrIK arrays resident in HBM' % (NUM_FRAMES, CFG_ID, iters, B),
                       'sequences_per_gpu': B, 'frames': NUM_FRAMES, 'persons': 1, 'parallelism': 'sequence-sharded x%d' % world,
                       'streams_per_gpu': len(streams)},
            'roofline': {'kernel': 'grecon_stage_kernel', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': TRAFFIC_BYTES_PER_SCENE_ITER * B * iters, 'avg_launch_ms': k_avg,
                         'us_per_iteration': k_avg * 1e3 / iters, 'dependent_boundary_floor_us': 1.45,
                         'launch_ms_each': [round(x, 2) for x in kms], 'launch_ms_in_timed_region': [round(x, 2) for x in kms_timed],
                         'measured': ('the launches of the timed region (one stream)' if len(streams) == 1 else
                                      '2 single-stream steps right after the timed region: in the timed region the launches of the %d streams '
                                      'share the CUs and each spans about twice its own duration' % len(streams)),
                         'note': 'latency-bound: one workgroup per scene, state on chip; traffic = PMC FETCH_SIZE x 2 + WRITE_SIZE per launch (profiles/r01_pmc_stage_kernel_b1024.csv) scaled to this batch, '
                                 'about a third of the algorithmic bytes because parameters are the only per-iteration stream; algorithmic bytes = %d B per person-iteration '
                                 '(SURVEY.md 8d K5) x %d scenes x %d iterations' % (ALGO_BYTES_PER_PERSON_ITER, B, iters)},
            'host_inclusive_sequences_per_sec': B / host_elapsed,
            'host_inclusive_stage_seconds': {k: round(v, 4) for k, v in tm.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(asset_root)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
