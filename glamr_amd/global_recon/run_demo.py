"""Global reconstruction of one sequence from a pose-estimation result: the part of the reference's `global_recon/run_demo.py`
(:44-82) between HybrIK (`pose.pkl`) and the visualiser, on the MI355X path.

    python -m glamr_amd.global_recon.run_demo --cfg glamr_dynamic --pose_est_dir out/glamr_dynamic/running/pose_est --out_dir out/glamr_dynamic/running

Reads `<pose_est_dir>/pose.pkl`, writes `<out_dir>/grecon/<seq_name>_seed<seed>.pkl` (same dictionary as the reference).  Working
directory conventions are the reference's: `data/body_models/smpl/`, `data/J_regressor_extra.npy`, `results/...` checkpoints."""
import argparse
import os
import pickle

import numpy as np
import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='glamr_dynamic')
    ap.add_argument('--pose_est_dir', required=True)
    ap.add_argument('--out_dir', required=True)
    ap.add_argument('--seq_name', default=None)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--cached', type=int, default=1)
    args = ap.parse_args(argv)

    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.utils import wire

    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    device = torch.device('cuda', args.gpu)
    torch.cuda.set_device(args.gpu)
    cfg = get_config(args.cfg)
    seq_name = args.seq_name or os.path.basename(os.path.normpath(args.out_dir))
    out_file = os.path.join(args.out_dir, 'grecon', '%s_seed%d.pkl' % (seq_name, args.seed))
    if args.cached and os.path.exists(out_file):
        print('cached result:', out_file)
        return out_file
    in_dict = wire.load_pose_pkl(os.path.join(args.pose_est_dir, 'pose.pkl'), seq_name=seq_name)
    model = model_dict[cfg.get('grecon_model_name', 'global_recon_model')](cfg, device, None)
    out_dict = model.optimize(in_dict)
    os.makedirs(os.path.dirname(out_file), exist_ok=True)
    with open(out_file, 'wb') as f:
        pickle.dump(out_dict, f)
    print('saved', out_file, '| losses of the last evaluation:', np.round(model.last_losses[0], 4).tolist())
    return out_file


if __name__ == '__main__':
    main()
