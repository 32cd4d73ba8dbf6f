"""Reconstruct the test sequences of a dataset: the reference's `global_recon/run_dataset.py` on the MI355X path, with its command
line (run_dataset.py:42-47; README "Evaluation"):

    python -m glamr_amd.global_recon.run_dataset --dataset 3dpw --cfg glamr_3dpw --out_dir out/3dpw [--seeds 1,2,3] [--cached 1]

Per sequence it expects `<out_dir>/<seq>/pose_est/pose.pkl` (HybrIK, pose_est/hybrik_demo/demo.py; when the file is missing the
reference's own `run_pose_est_on_video` is called if it can be imported, as run_dataset.py:79-81 does) and the ground-truth pickle
`<gt_dir>/<seq>.pkl` (preprocess/preprocess_3dpw.py; `--gt_dir` defaults to the dataset's `processed_v1/pose` directory, :27-38), and
writes `<out_dir>/<seq>/grecon/<seq>_seed<k>.pkl` (:91-101, same dictionary).  Extensions: `--seqs` restricts / replaces the sequence
list, `--seeds` also takes space-separated integers, and with ground truth present the metric line of `eval_dataset.py` is printed
(returned by main()).  All sequences of one seed are optimised as ONE batch on the device."""
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
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
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
    device = torch.device('cuda', args.gpu)
    torch.cuda.set_device(args.gpu)
    cfg = get_config(args.cfg)
    model = model_dict[cfg.get('grecon_model_name', 'global_recon_model')](cfg, device, None)
    evaluator = Evaluator(algo=args.cfg, dataset=dataset, device=device, smpl=model.smpl) if gt_dir else None
    # (the name is handed on as typed, like eval_dataset.py:38 -- the evaluator's y-up branch tests for '3DPW', evaluator.py:250)

    in_dicts = []
    for seq in seqs:
        pose_dir = os.path.join(args.out_dir, seq, 'pose_est')
        if not os.path.exists(os.path.join(pose_dir, 'pose.pkl')):
            run_pose_est = _need('pose_est.run_pose_est_demo', 'run_pose_est_on_video', 'sequence %s without %s/pose.pkl' % (seq, pose_dir))
            run_pose_est(None, pose_dir, cfg['grecon_model_specs']['est_type'], image_dir=os.path.join(paths.get('image', ''), seq),
                         bbox_file=os.path.join(paths.get('bbox', ''), seq + '.pkl'), cached_pose=int(args.cached), gpu_index=args.gpu)
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
    if evaluator is None:
        return None
    for seq in seqs:                               # eval_dataset.py: best / mean over seeds per sequence, then accumulate
        evaluator.update_accumulated_metrics(evaluator.metrics_from_multiple_seeds(per_seed[seq]), seq)
    return evaluator.print_metrics(prefix='%s %s: ' % (dataset, args.cfg), print_accum=False)


if __name__ == '__main__':
    main()
