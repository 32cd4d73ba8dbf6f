"""Reconstruct and evaluate a list of sequences: the reference's `global_recon/run_dataset.py` (:60-120) after pose estimation --
pose.pkl + ground-truth pickle per sequence in, grecon/<seq>_seed<k>.pkl and the metric table out.

    python -m glamr_amd.global_recon.run_dataset --cfg glamr_3dpw --dataset 3DPW --seqs seq_a seq_b \\
        --out_dir out/3dpw --gt_dir datasets/3DPW/processed_v1/pose --seeds 1 2 3

Expects `<out_dir>/<seq>/pose_est/pose.pkl` (HybrIK, pose_est/hybrik_demo/demo.py) and `<gt_dir>/<seq>.pkl`
(preprocess/preprocess_3dpw.py).  All sequences of one seed are optimised as ONE batch on the device."""
import argparse
import copy
import os
import pickle

import numpy as np
import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='glamr_3dpw')
    ap.add_argument('--dataset', default='3DPW')
    ap.add_argument('--seqs', nargs='+', required=True)
    ap.add_argument('--out_dir', required=True)
    ap.add_argument('--gt_dir', default=None)
    ap.add_argument('--seeds', nargs='+', type=int, default=[1])
    ap.add_argument('--gpu', type=int, default=0)
    args = ap.parse_args(argv)

    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.utils.evaluator import Evaluator
    from glamr_amd.utils import wire

    device = torch.device('cuda', args.gpu)
    torch.cuda.set_device(args.gpu)
    cfg = get_config(args.cfg)
    model = model_dict[cfg.get('grecon_model_name', 'global_recon_model')](cfg, device, None)
    evaluator = Evaluator(algo=args.cfg, dataset=args.dataset, device=device, smpl=model.smpl) if args.gt_dir else None

    in_dicts = []
    for seq in args.seqs:
        with open(os.path.join(args.out_dir, seq, 'pose_est', 'pose.pkl'), 'rb') as f:
            est = pickle.load(f)
        gt, meta = {}, {}
        if args.gt_dir:
            with open(os.path.join(args.gt_dir, seq + '.pkl'), 'rb') as f:
                num_fr = len(next(iter(est.values()))['bboxes_dict']['exist'])
                gt, meta = wire.normalise_gt(pickle.load(f), num_frames=num_fr)
        in_dicts.append(wire.make_in_dict(est, seq, gt=gt, gt_meta=meta))

    per_seed = {seq: [] for seq in args.seqs}
    for seed in args.seeds:
        np.random.seed(seed)
        torch.manual_seed(seed)
        outs = model.optimize_batch(in_dicts)
        for seq, out in zip(args.seqs, outs):
            d = os.path.join(args.out_dir, seq, 'grecon')
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, '%s_seed%d.pkl' % (seq, seed)), 'wb') as f:
                pickle.dump(out, f)
            if evaluator is not None:
                # prepare_seq() trims and extends the dictionaries in place: the ground truth is shared with the next seed's input
                work = dict(out, gt=copy.deepcopy(out['gt']), person_data=copy.deepcopy(out['person_data']))
                per_seed[seq].append(evaluator.compute_sequence_metrics(work, '%s_seed%d' % (seq, seed), accumulate=False))
    if evaluator is None:
        return None
    for seq in args.seqs:                          # run_dataset.py / eval_dataset.py: best / mean over seeds per sequence, then accumulate
        evaluator.update_accumulated_metrics(evaluator.metrics_from_multiple_seeds(per_seed[seq]), seq)
    return evaluator.print_metrics(prefix='%s %s: ' % (args.dataset, args.cfg), print_accum=False)


if __name__ == '__main__':
    main()
