"""Multi-GPU partitioning of the hot path: sequences (and the persons inside them) are independent units
(global_recon/run_dataset.py:67-105 loops them serially), so a node is used by giving every rank its own contiguous block of
sequences -- no data-path collective.  torch.distributed (RCCL on MI355X, gloo in the CPU tests) is used only to rendezvous,
to agree on the elapsed time (max over ranks) and to gather small per-sequence results on rank 0."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def shard_range(n_items, rank, world):
    """Contiguous, balanced block [lo, hi) of rank `rank`; the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def weak_scaling_seeds(per_rank, rank):
    """bench.py: every rank works on `per_rank` sequences of its own (weak scaling)."""
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def self_launch(n, target, argv, keep_on_stdout=None):
    """A command that was started plainly with `--gpus N` (no RANK in the environment) becomes N ranks: `target` (['script.py'] or
    ['-m', 'package.module']) is re-executed under torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 at a free port.
    The ranks' stdout passes through a filter: lines `keep_on_stdout(line)` accepts (default: all) stay on stdout, whatever else a backend
    prints there (gloo's connection notes, RCCL's banner) goes to stderr.  Returns the launcher's exit code."""
    import socket
    import subprocess
    import sys
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # this pool's driver only supports dmabuf IPC (RCCL fails without it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port)] + list(target) + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        (sys.stdout if (keep_on_stdout is None or keep_on_stdout(line)) else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def init_from_env(backend=None, device=None):
    """Process group of a rank started by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment): `nccl` (= RCCL) on a HIP
    device, gloo otherwise.  Returns (rank, world, local_rank)."""
    rank, world, local_rank = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = backend or ('nccl' if (device is not None and device.type == 'cuda') else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """Elapsed time of the slowest rank (what the driver's clock sees)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    """Units all ranks processed (the numerator of the whole-job throughput)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_results(local_results, dst=0):
    """Per-sequence result summaries (small python objects) collected on rank `dst` in global sequence order."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(local_results)
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(list(local_results), out, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [x for part in out for x in part]


# ------------------------------------------------------------------------------------------------------------------------------------
# Person-sharded scenes: the collective variant of BASELINE configs[3]
# ------------------------------------------------------------------------------------------------------------------------------------

def _device_run_stage(packed, sd, want_grads, ws=None):
    """glamr_grecon_run_stage on the packed batch (current stream); returns the gradient record (n_scenes, scene_stride) or None.  `ws`: a
    workspace the caller keeps across launches (GLAMR_FLAG_KEEP_TABLES needs the previous launch's); a fresh one otherwise."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    sb = packed.struct()
    grads = torch.zeros_like(packed.t['params']) if want_grads else None
    if ws is None:
        ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=packed.device)
    _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), _lib.ptr(grads), _lib.ptr(ws), _lib.current_stream()))
    packed.last_ws = ws
    return grads


def _device_adam_step(p, m, v, g, lr, step):
    from . import _lib
    _lib.check(_lib.lib().glamr_adam_step(p.numel(), _lib.ptr(p), _lib.ptr(m), _lib.ptr(v), _lib.ptr(g), float(lr), int(step), _lib.current_stream()))


class PersonShardedSchedule:
    """The staged optimisation of multi-person scenes with the PERSONS of a scene sharded over the ranks of a process group -- the variant
    north_star names for BASELINE configs[3] ("RCCL over xGMI only for the shared-camera reduction"), built next to the default one (a
    4-person scene fits one workgroup, whose block reduction sums the shared camera's gradient on chip: GlobalReconOptimizer.run_schedule)
    so that both can be measured (SURVEY.md 8e "measure both and report").

    What couples the persons of a scene, per Adam iteration (global_recon/models/global_recon_model.py:475-477,597-601, loss_func.py:248-271):
      * the shared camera: every person's reprojection / camera-orientation residuals pull on the same `cam_rot_6d(_fix)`, `cam_trans(_fix)`
        -> ALL-REDUCE (sum) of the camera block of the gradient (9 floats with `flag_fixed_cam`, 9 per frame otherwise);
      * `rel_transform`: each ordered pair (i, j) compares inv(T_i) T_j with its camera-frame value, T = person_transform_world
        -> ALL-GATHER of every person's world pose (smpl_orient_world, root_trans_world: 6 floats per frame) before the gradients are formed.
    Normalisers (visible frames, existing frames, persons) are global; every rank knows all persons' visibility masks and existence ranges.

    One iteration on every rank, host-orchestrated over the existing ABI (no new kernel: glamr_grecon_run_stage with `frozen` person slots):
      1. forward-only launch (niters 0): world poses of the rank's OWN persons at the current parameters;
      2. all-gather -> the other ranks' persons are FROZEN slots of this rank's scene (pose given, no residuals of their own, no gradient);
      3. gradient launch (niters 1, lr 0, grads_out): gradients of the own persons' variables and this rank's PART of the camera gradient
         (the camera-only regularisers are counted on rank 0 only: GLAMR_FLAG_NO_CAMERA_TERMS elsewhere);
      4. all-reduce of the camera block;  5. glamr_adam_step (the optimiser's own update function, torch.optim.Adam to the bit) on the
         replicated camera parameters and the own persons' blocks, moments kept by this object across the launches of a stage.
    The INITIALISATION is replicated (every rank packs the whole scene; init_data is ~1 % of the schedule): what is sharded is the
    700-iteration loop.  Stages whose camera is DERIVED from the persons (flag_opt_cam_from_person_pose without 'cam') are refused: there
    the camera average carries gradients from every person's residuals to every other person, which this exchange does not cover.

    `run_stage(packed, stage_desc, want_grads)` / `adam_step(p, m, v, g, lr, step)`: the device entry points by default; the CPU tests
    inject the host runtime of tests/hostsim and run two ranks over gloo."""

    def __init__(self, rank=None, world=None, group=None, run_stage=None, adam_step=None, grad_hook=None, use_dist=None):
        initialised = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if initialised else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if initialised else 1)
        self.group = group
        self.run_stage = run_stage or _device_run_stage
        self.adam_step = adam_step or _device_adam_step
        # collectives are issued whenever a process group exists -- also in a world of ONE rank (bench.py --force-dist, the single-GPU test of
        # the RCCL path): nothing is exchanged then, but the calls and their stream ordering are the ones of the N-rank run
        self.use_dist = initialised if use_dist is None else bool(use_dist)
        # grad_hook(packed, stage, spec, grads): edits the gradient record of an iteration in place before the Adam step -- how the launch-by-launch
        # schedule also serves model flags that only change WHICH variables move (GlobalReconOptimizer: flag_opt_vis_local_rot)
        self.grad_hook = grad_hook
        self.collective_seconds = 0.0
        self.launches = 0

    def owned(self, n_persons):
        lo, hi = shard_range(n_persons, self.rank, self.world)
        return list(range(lo, hi))

    def _all_gather_poses(self, packed, own, block):
        """Own persons' (orient_world, trans_world) of every scene -> everybody; the others' land in the base arrays of their frozen slots."""
        import time
        if not self.use_dist:
            return
        S, P, T = packed.S, packed.P, packed.T
        ow, tw = packed.t['orient_world'].view(S, P, T, 3), packed.t['trans_world'].view(S, P, T, 3)
        mine = torch.zeros((block, S, T, 6), dtype=torch.float32, device=ow.device)
        for k, pi in enumerate(own):
            mine[k, :, :, :3], mine[k, :, :, 3:] = ow[:, pi], tw[:, pi]
        t0 = time.time()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        self.collective_seconds += time.time() - t0
        bo, bt = packed.t['base_orient'].view(S, P, T, 3), packed.t['base_trans'].view(S, P, T, 3)
        for r, part in enumerate(parts):
            if r == self.rank:
                continue
            lo, hi = shard_range(P, r, self.world)
            for k, pi in enumerate(range(lo, hi)):
                bo[:, pi], bt[:, pi] = part[k, :, :, :3], part[k, :, :, 3:]

    def _all_reduce(self, t):
        import time
        if self.use_dist:
            t0 = time.time()
            dist.all_reduce(t, group=self.group)
            self.collective_seconds += time.time() - t0

    def run(self, packed, opt_stage_specs, model_specs, max_iters=None, has_wd=False):
        """Runs the whole schedule on `packed` (the FULL scene batch, identical on every rank when the call starts).  On return every rank
        holds the full result: parameters, world poses and projections of all persons, the camera."""
        from .global_recon import packing
        S, P, T, l = packed.S, packed.P, packed.T, packed.layout
        if P < self.world:
            raise ValueError('%d persons cannot be sharded over %d ranks' % (P, self.world))
        own = self.owned(P)
        block = -(-P // self.world)
        frozen = torch.ones((S, P), dtype=torch.int32)
        frozen[:, own] = 0
        packed.t['frozen'] = frozen.reshape(-1).to(packed.device)
        params = packed.t['params']
        person_cols = [slice(l['person0'] + pi * l['person_stride'], l['person0'] + (pi + 1) * l['person_stride']) for pi in range(P)]
        for stage, spec in opt_stage_specs.items():
            if model_specs.get('flag_opt_cam_from_person_pose', False) and 'cam' not in spec['opt_variables']:
                raise NotImplementedError('stage %r derives the camera from the persons: not covered by the person-sharded exchange' % stage)
            n = spec['opt_niters'] if max_iters is None else min(max_iters, spec['opt_niters'])
            m, v = torch.zeros_like(params), torch.zeros_like(params)
            not_own = [pi for pi in range(P) if pi not in own]
            grad_ws = None
            if self.run_stage is _device_run_stage and params.is_cuda and os.environ.get('GLAMR_SHARDED_KEEP_TABLES', '1') != '0':
                from . import _lib
                grad_ws = torch.empty(_lib.lib().glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=packed.device)

            def iteration(it, adam, report=True):
                # report=False (every iteration but the stage's last): the gradient launch skips the reporting part of its evaluation --
                # all 26 joints, the outputs, the 13 loss reductions (GLAMR_FLAG_NO_REPORT); the stage's outputs are the last iteration's
                keep = packing.FLAG_KEEP_CAM_PARAMS if it > 0 else 0
                fwd = packing.stage_desc(spec, model_specs, has_wd, niters=0)
                fwd.flags |= keep | packing.FLAG_POSES_ONLY                         # (only the world poses are wanted: no residuals, no projections)
                self.run_stage(packed, fwd, False)                                   # 1. own poses at the current parameters (+ camera parameters at it 0)
                self._all_gather_poses(packed, own, block)                           # 2.
                gd = packing.stage_desc(spec, model_specs, has_wd, niters=1)
                gd.lr = 0.0                                                          # the update is made below, after the reduction
                gd.flags |= packing.FLAG_KEEP_CAM_PARAMS | (packing.FLAG_NO_CAMERA_TERMS if self.rank != 0 else 0) | (0 if report else packing.FLAG_NO_REPORT)
                if grad_ws is not None:
                    # the gradient launches of a stage share ONE workspace nothing else writes to: from the second on, the stage-constant tables of
                    # the set-up are still in it (GLAMR_FLAG_KEEP_TABLES: ~100 us of a 250 us launch)
                    gd.flags |= packing.FLAG_KEEP_TABLES if it > 0 else 0
                    grads = self.run_stage(packed, gd, True, ws=grad_ws)                 # 3.
                else:
                    grads = self.run_stage(packed, gd, True)                         # 3.
                self.launches += 2
                if self.use_dist:
                    cam = grads[:, :l['person0']].contiguous()
                    self._all_reduce(cam)                                            # 4.
                    grads[:, :l['person0']] = cam
                for pi in not_own:
                    grads[:, person_cols[pi]] = 0.0
                if self.grad_hook is not None:
                    self.grad_hook(packed, stage, spec, grads)
                adam(grads)                                                          # 5.

            graph = self._iteration_graph(packed, params, m, v, spec['opt_lr'], n, iteration) if n > 3 else None
            if graph is None:
                for it in range(n):
                    iteration(it, lambda g, it=it: self.adam_step(params.view(-1), m.view(-1), v.view(-1), g.view(-1), spec['opt_lr'], it + 1), report=(it == n - 1))
            has_wd = has_wd or 'world_dheading' in spec['opt_variables']
            if spec.get('reinitialize_cam', False):
                packed.t['cam_pose'][:] = packed.t['cam_pose'][:, :1]
        # every rank ends with the whole scene: own persons' variables and outputs to everybody -- including the BASE poses of the persons it
        # did not own (during the loop those slots carried the peers' world poses: _all_gather_poses) and the loss values of the whole scene
        self._share_results(packed, own, block, person_cols)
        self._reduce_losses(packed, own)
        packed.t['frozen'] = None
        packed.has_world_dheading = has_wd
        return packed

    def _iteration_graph(self, packed, params, m, v, lr, n, iteration):
        """Iterations 2 .. n - 2 of a stage as replays of ONE captured HIP graph, iterations 0, 1 and n - 1 plainly (the last one reports: outputs and loss
        values; the others carry GLAMR_FLAG_NO_REPORT) (device entry points only): forward-only launch, all-gather, gradient
        launch, all-reduce, Adam -- five launches and two collectives enqueued by a single graph launch, the Adam step number read from a device
        counter (glamr_adam_step_indexed / glamr_counter_add).  Host-orchestrated the same sequence costs ~420 us per iteration against ~100 us of
        kernel time (bench.py `configs3_person_sharded`): on an 8-GPU node that would measure launches, not links.  Iteration 0 runs plainly (it
        initialises the camera parameters and every lazily built table; RCCL communicators must exist before a capture).  Returns True when the
        whole stage has run; None = not applicable (injected run_stage / adam_step, CPU tensors, GLAMR_SHARDED_GRAPH=0) or the capture failed, in
        which case nothing of the stage has run and the caller's plain loop takes over."""
        import os
        # With more than one rank the captured iteration (collectives inside a HIP graph) is OPT-IN (GLAMR_SHARDED_GRAPH=1): it has only ever run
        # on a process group of one rank -- no multi-GPU box was available to this project -- and a capture that fails on one rank alone would
        # leave the ranks in different collective sequences (a hang, not an error).  The plain loop is what the world_size-2 tests cover.
        knob = os.environ.get('GLAMR_SHARDED_GRAPH', '1' if self.world <= 1 else '0')
        if self.run_stage is not _device_run_stage or self.adam_step is not _device_adam_step or not params.is_cuda or knob == '0':
            return None
        import ctypes
        import numpy as np
        from . import _lib
        L = _lib.lib()
        tab = np.zeros((n, 2), np.float32)
        _lib.check(L.glamr_adam_coef_table(ctypes.c_double(float(lr)), int(n), _lib.ptr(tab)))
        tab_d = torch.from_numpy(tab).to(params.device)
        step = torch.zeros(1, dtype=torch.int32, device=params.device)

        def adam_indexed(g):
            st = _lib.current_stream()
            _lib.check(L.glamr_adam_step_indexed(params.numel(), _lib.ptr(params), _lib.ptr(m), _lib.ptr(v), _lib.ptr(g), _lib.ptr(tab_d), _lib.ptr(step), st))
            _lib.check(L.glamr_counter_add(_lib.ptr(step), 1, st))
        snapshot = (params.clone(), m.clone(), v.clone(), self.launches)
        try:
            iteration(0, adam_indexed, report=False)
            iteration(1, adam_indexed, report=False)                                # (a plain iteration of the captured kind first: allocations, attribute calls)
            torch.cuda.synchronize(params.device)
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=params.device)
            side.wait_stream(torch.cuda.current_stream(params.device))
            with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                iteration(2, adam_indexed, report=False)                            # (iteration number only selects KEEP_CAM_PARAMS: any it > 0 is the same graph)
            self.launches -= 2                                                      # (capturing launched nothing)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(2, n - 1):
                g.replay()
                self.launches += 2
            ev1.record()
            self.__dict__.setdefault('replay_events', []).append((ev0, ev1, n - 3))      # (device time of the replayed iterations: bench.py)
            iteration(n - 1, adam_indexed, report=True)                             # the stage's last iteration, plainly: its evaluation reports
            self.iteration_graphs = getattr(self, 'iteration_graphs', 0) + 1
            return True
        except Exception as e:      # noqa: BLE001 -- anything (a runtime without capturable collectives, ...): restore and let the plain loop run
            import sys
            sys.stderr.write('PersonShardedSchedule: iteration graph not used (%s)\n' % e)
            torch.cuda.synchronize(params.device)
            params.copy_(snapshot[0]); m.copy_(snapshot[1]); v.copy_(snapshot[2])
            self.launches = snapshot[3]
            return None

    def _reduce_losses(self, packed, own):
        """packed.t['losses'] after a sharded run: every rank's last evaluation reported ITS part -- the residuals of its own persons over the
        GLOBAL normalisers -- so the scene's values are the sum over ranks.  Two exceptions: the camera-only terms (their VALUES are
        reported by every rank, only their gradients are rank 0's: GLAMR_FLAG_NO_CAMERA_TERMS) count once, and the monitor-only keypoint
        distance is a mean over the scored keypoints the rank saw: ranks are weighted by their own persons' visible frames."""
        if not self.use_dist or self.world == 1:
            return
        from . import _lib
        S, P, T = packed.S, packed.P, packed.T
        losses = packed.t['losses']
        vis = (packed.t['vis'].view(S, P, T) > 0).sum(dim=2).to(losses.dtype)              # visible frames per (scene, person)
        w = vis[:, own].sum(dim=1)
        part = losses.clone()
        if self.rank != 0:
            part[:, _lib.LOSS_CAMERA_ONLY[0]:_lib.LOSS_CAMERA_ONLY[1]] = 0.0
        buf = torch.cat([part, (losses[:, _lib.LOSS_KP_2D_DIST] * w)[:, None], w[:, None]], dim=1).contiguous()
        self._all_reduce(buf)
        losses.copy_(buf[:, :losses.shape[1]])
        losses[:, _lib.LOSS_KP_2D_DIST] = buf[:, -2] / buf[:, -1].clamp_min(1.0)

    def _share_results(self, packed, own, block, person_cols):
        if not self.use_dist:
            return
        S, P, T = packed.S, packed.P, packed.T
        names = (('orient_world', 3), ('trans_world', 3), ('orient_cam_in_world', 3), ('kp_2d_pred', 52), ('base_orient', 3), ('base_trans', 3))
        width = sum(w for _, w in names)
        stride = person_cols[0].stop - person_cols[0].start
        mine = torch.zeros((block, S, T * width + stride), dtype=torch.float32, device=packed.device)
        for k, pi in enumerate(own):
            o = 0
            for name, w in names:
                mine[k, :, o:o + T * w] = packed.t[name].view(S, P, T * w)[:, pi]
                o += T * w
            mine[k, :, o:] = packed.t['params'][:, person_cols[pi]]
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        for r, part in enumerate(parts):
            if r == self.rank:
                continue
            lo, hi = shard_range(P, r, self.world)
            for k, pi in enumerate(range(lo, hi)):
                o = 0
                for name, w in names:
                    packed.t[name].view(S, P, T * w)[:, pi] = part[k, :, o:o + T * w]
                    o += T * w
                packed.t['params'][:, person_cols[pi]] = part[k, :, o:]
