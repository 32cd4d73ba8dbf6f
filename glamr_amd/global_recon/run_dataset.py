"""Reconstruct the test sequences of a dataset: the reference's `global_recon/run_dataset.py` on the MI355X path, with its command
line (run_dataset.py:42-47; README "Evaluation"):

    python -m glamr_amd.global_recon.run_dataset --dataset 3dpw --cfg glamr_3dpw --out_dir out/3dpw [--seeds 1,2,3] [--cached 1]

Per sequence it expects `<out_dir>/<seq>/pose_est/pose.pkl` (HybrIK, pose_est/hybrik_demo/demo.py; when the file is missing the
reference's own `run_pose_est_on_video` is called if it can be imported, as run_dataset.py:79-81 does) and the ground-truth pickle
`<gt_dir>/<seq>.pkl` (preprocess/preprocess_3dpw.py; `--gt_dir` defaults to the dataset's `processed_v1/pose` directory, :27-38), and
writes `<out_dir>/<seq>/grecon/<seq>_seed<k>.pkl` (:91-101, same dictionary).  Extensions: `--seqs` restricts / replaces the sequence
list, `--seeds` also takes space-separated integers, and with ground truth present the metric line of `eval_dataset.py` is printed
(returned by main()).  All sequences of one seed are optimised as ONE batch on the device.

`--gpus N` (extension; BASELINE configs[4]: the 3DPW test set on the 8 GPUs of a node): the sequence list is split into N contiguous
balanced blocks (parallel.shard_range), one process per GPU reconstructs and evaluates its block -- sequences are independent, no
data-path collective -- and the per-sequence metrics are gathered on rank 0 (parallel.gather_results over RCCL), which folds them in
sequence order and prints the metric line.  Started plainly the command spawns its own ranks (parallel.self_launch); started under
`torch.distributed.run --nproc-per-node N` it takes RANK / WORLD_SIZE from the environment.  The reference loops sequences serially on
one GPU (run_dataset.py:67-105)."""
import argparse
import copy
import glob
import os
import pickle

import numpy as np
import torch

_3DPW_TEST = ('downtown', 'arguing_00 bar_00 bus_00 cafe_00 car_00 crossStreets_00 downstairs_00 enterShop_00 rampAndStairs_00 runForBus_00 runForBus_01 '
              'sitOnStairs_00 stairs_00 upstairs_00 walkBridge_01 walkUphill_00 walking_00 warmWelcome_00 weeklyMarket_00 windowShopping_00')
DATASET_PATHS = {          # run_dataset.py:27-38
    '3dpw': dict(image='datasets/3DPW/imageFiles', bbox='datasets/3DPW/processed_v1/bbox', gt_pose='datasets/3DPW/processed_v1/pose'),
    'h36m': dict(image='datasets/H36M/occluded_v2/images', bbox='datasets/H36M/occluded_v2/bbox', gt_pose='datasets/H36M/occluded_v2/pose'),
}


def test_sequences(dataset):
    """The sequence lists of run_dataset.py:18-24 (the 3DPW test split; subjects 9 and 11 of H36M as found on disk)."""
    if dataset == '3dpw':
        return ['%s_%s' % (_3DPW_TEST[0], s) for s in _3DPW_TEST[1].split()] + ['flat_guitar_01', 'flat_packBags_00', 'office_phoneCall_00', 'outdoors_fencing_01']
    if dataset == 'h36m':
        return sorted(glob.glob('datasets/H36M/processed_v1/pose/s_09*.pkl')) + sorted(glob.glob('datasets/H36M/processed_v1/pose/s_11*.pkl'))
    return []


def parse_seeds(values):
    """'1,2,3' (the reference, :56) or separate integers."""
    if isinstance(values, str):
        values = [values]
    return [int(x) for v in values for x in str(v).split(',') if x != '']


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--dataset', default='3dpw')
    ap.add_argument('--cfg', default='glamr_3dpw')
    ap.add_argument('--out_dir', default='out/3dpw')
    ap.add_argument('--seeds', nargs='+', default=['1'])
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--cached', type=int, default=1)
    ap.add_argument('--seqs', nargs='+', default=None, help='extension: sequences to run (default: the test sequences of --dataset)')
    ap.add_argument('--gt_dir', default=None, help="extension: ground-truth directory (default: the dataset's gt_pose path when it exists)")
    ap.add_argument('--gpus', type=int, default=1, help='extension: shard the sequences over N GPUs of this node, one process per GPU (rank r uses GPU r; --gpu is the single-process device)')
    ap.add_argument('--backend', default=None, help='extension: torch.distributed backend of --gpus N (default nccl = RCCL)')
    ap.add_argument('--stub-model', action='store_true', help=argparse.SUPPRESS)      # CPU stand-ins for the optimiser and the evaluator: tests of the sharding protocol over gloo
    return ap


class _StubModel:
    """CPU stand-in (tests/test_parallel_gloo.py): the result of a sequence is a function of its input alone, so the gathered metric line must
    not depend on how the sequences were split."""
    smpl = None

    def optimize_batch(self, in_dicts):
        return [dict(d, seq_len=len(next(iter(d['est'].values()))['bboxes_dict']['exist']), person_data={}) for d in in_dicts]


class _StubEvaluator:
    def __init__(self):
        from glamr_amd.global_recon.utils.evaluator import AverageMeter
        from collections import defaultdict
        self.AverageMeter, self.totals, self.order = AverageMeter, defaultdict(AverageMeter), []

    def compute_sequence_metrics(self, data, name=None, accumulate=False):
        est = next(iter(data['est'].values()))
        return {'seq_len': data['seq_len'], 'metrics': {'G-MPJPE': self.AverageMeter(float(np.abs(est['root_trans']).sum()) + sum(map(ord, name or "")) % 7, data['seq_len'])}}

    def metrics_from_multiple_seeds(self, arr):
        return {'seq_len': arr[0]['seq_len'], 'metrics': {'G-MPJPE': self.AverageMeter(float(np.mean([m['metrics']['G-MPJPE'].avg for m in arr])), arr[0]['metrics']['G-MPJPE'].count)}}

    def update_accumulated_metrics(self, m, name=None):
        self.order.append(name)
        self.totals['G-MPJPE'].update(m['metrics']['G-MPJPE'].avg, m['metrics']['G-MPJPE'].count)

    def print_metrics(self, prefix='', print_accum=False):
        line = '%sstub --- G-MPJPE: %.6f sequences: %s' % (prefix, self.totals['G-MPJPE'].avg, ','.join(self.order))
        print(line)
        return line


def main(argv=None):
    args = build_parser().parse_args(argv)
    from glamr_amd import parallel
    if args.gpus > 1 and 'RANK' not in os.environ:
        import sys
        rc = parallel.self_launch(args.gpus, ['-m', 'glamr_amd.global_recon.run_dataset'], list(sys.argv[1:] if argv is None else argv))
        if rc:
            raise SystemExit(rc)
        return None
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.utils.evaluator import Evaluator
    from glamr_amd.global_recon.run_demo import _need
    from glamr_amd.utils import wire

    seeds = parse_seeds(args.seeds)
    dataset = args.dataset
    paths = DATASET_PATHS.get(dataset.lower(), {})
    seqs = args.seqs if args.seqs is not None else [os.path.splitext(os.path.basename(s))[0] for s in test_sequences(dataset.lower())]
    if not seqs:
        raise SystemExit('no sequences: dataset %r has no built-in test list here, pass --seqs' % dataset)
    gt_dir = args.gt_dir if args.gt_dir is not None else (paths.get('gt_pose') if paths and os.path.isdir(paths['gt_pose']) else None)
    rank, world, local_rank = parallel.env_rank_world()
    if world != max(1, args.gpus):
        raise SystemExit('--gpus %d inside a job of WORLD_SIZE=%d' % (args.gpus, world))
    cfg = get_config(args.cfg)
    if args.stub_model:
        device = torch.device('cpu')
        parallel.init_from_env(args.backend or 'gloo', device)
        model, evaluator = _StubModel(), (_StubEvaluator() if gt_dir else None)
    else:
        gpu = local_rank if world > 1 else args.gpu
        device = torch.device('cuda', gpu)
        torch.cuda.set_device(gpu)
        parallel.init_from_env(args.backend, device)
        model = model_dict[cfg.get('grecon_model_name', 'global_recon_model')](cfg, device, None)
        evaluator = Evaluator(algo=args.cfg, dataset=dataset, device=device, smpl=model.smpl) if gt_dir else None
    all_seqs = list(seqs)
    lo, hi = parallel.shard_range(len(all_seqs), rank, world)            # this rank's contiguous block of sequences
    seqs = all_seqs[lo:hi]
    # (the name is handed on as typed, like eval_dataset.py:38 -- the evaluator's y-up branch tests for '3DPW', evaluator.py:250)

    in_dicts = []
    for seq in seqs:
        pose_dir = os.path.join(args.out_dir, seq, 'pose_est')
        if not os.path.exists(os.path.join(pose_dir, 'pose.pkl')):
            run_pose_est = _need('pose_est.run_pose_est_demo', 'run_pose_est_on_video', 'sequence %s without %s/pose.pkl' % (seq, pose_dir))
            run_pose_est(None, pose_dir, cfg['grecon_model_specs']['est_type'], image_dir=os.path.join(paths.get('image', ''), seq),
                         bbox_file=os.path.join(paths.get('bbox', ''), seq + '.pkl'), cached_pose=int(args.cached), gpu_index=(device.index or 0) if device.type == 'cuda' else args.gpu)
        with open(os.path.join(pose_dir, 'pose.pkl'), 'rb') as f:
            est = pickle.load(f)
        gt, meta = {}, {}
        if gt_dir:
            with open(os.path.join(gt_dir, seq + '.pkl'), 'rb') as f:
                num_fr = len(next(iter(est.values()))['bboxes_dict']['exist'])
                gt, meta = wire.normalise_gt(pickle.load(f), num_frames=num_fr)
        in_dicts.append(wire.make_in_dict(est, seq, gt=gt, gt_meta=meta))

    per_seed = {seq: [] for seq in seqs}
    for seed in seeds:
        np.random.seed(seed)
        torch.manual_seed(seed)
        files = [os.path.join(args.out_dir, seq, 'grecon', '%s_seed%d.pkl' % (seq, seed)) for seq in seqs]
        todo = [i for i, f in enumerate(files) if not (args.cached and os.path.exists(f))]
        outs = dict(zip(todo, model.optimize_batch([in_dicts[i] for i in todo]))) if todo else {}
        for i, (seq, fn) in enumerate(zip(seqs, files)):
            if i in outs:
                os.makedirs(os.path.dirname(fn), exist_ok=True)
                with open(fn, 'wb') as f:
                    pickle.dump(outs[i], f)
                out = outs[i]
            else:
                with open(fn, 'rb') as f:
                    out = pickle.load(f)
            if evaluator is not None:
                # prepare_seq() trims and extends the dictionaries in place: the ground truth is shared with the next seed's input
                work = dict(out, gt=copy.deepcopy(out['gt']), person_data=copy.deepcopy(out['person_data']))
                per_seed[seq].append(evaluator.compute_sequence_metrics(work, '%s_seed%d' % (seq, seed), accumulate=False))
    # every rank's (sequence, metrics over seeds) pairs -> rank 0, in global sequence order (blocks are contiguous and gathered by rank)
    combined = [(seq, evaluator.metrics_from_multiple_seeds(per_seed[seq])) for seq in seqs] if evaluator is not None else [(seq, None) for seq in seqs]
    gathered = parallel.gather_results(combined)
    line = None
    if gathered is not None:                       # rank 0 (or the only process)
        assert [s for s, _ in gathered] == all_seqs, 'gathered sequences out of order'
        if evaluator is not None:
            for seq, m in gathered:                # eval_dataset.py: best / mean over seeds per sequence, then accumulate
                evaluator.update_accumulated_metrics(m, seq)
            line = evaluator.print_metrics(prefix='%s %s: ' % (dataset, args.cfg), print_accum=False)
    if world > 1:
        import torch.distributed as dist
        parallel.barrier()
        dist.destroy_process_group()
    return line


if __name__ == '__main__':
    main()
