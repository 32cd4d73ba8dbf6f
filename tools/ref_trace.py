"""DEVELOPMENT AID (build container only: needs /root/reference).  Runs the UNMODIFIED reference on the 300-frame sequence of
BASELINE.json configs[1] (GLAMR_TRACE_DATA_SEED selects the synthetic sequence, default 0) and records, for every Adam iteration, the parameter values after the step and the gradients the step
used, so a kernel trajectory can be compared with it iteration by iteration (tools/diverge_probe.py).

    python tools/ref_trace.py [gap|nogap] [out.npz] [--threads N] [--eps E --seed S]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DATA_SEED = int(os.environ.get('GLAMR_TRACE_DATA_SEED', '0'))      # seed of the synthetic sequence (and of its latent draws)


def main(argv):
    which = argv[0] if argv and not argv[0].startswith('--') else 'gap'
    out = argv[1] if len(argv) > 1 and not argv[1].startswith('--') else '/tmp/ref_trace_%s.npz' % which
    threads = int(argv[argv.index('--threads') + 1]) if '--threads' in argv else None
    eps = float(argv[argv.index('--eps') + 1]) if '--eps' in argv else 0.0
    seed = int(argv[argv.index('--seed') + 1]) if '--seed' in argv else 0
    if threads:
        torch.set_num_threads(threads)
    from oracle import ref_harness as rh
    from oracle import make_golden as mg
    from glamr_amd.utils import synth
    rh.setup()
    trace = {'names': None, 'p': [], 'g': []}

    class Log:
        def info(self, *a, **k):
            if 'params' not in trace or ' | ' not in str(a[0]):
                return
            ps = trace['params']
            trace['p'].append(np.concatenate([p.detach().numpy().ravel() for p in ps]))
            trace['g'].append(np.concatenate([(p.grad if p.grad is not None else torch.zeros_like(p)).numpy().ravel() for p in ps]))

    model, cfg = rh.reference_optimizer('glamr_dynamic', log=Log())
    in_dict = synth.make_in_dict(seed=DATA_SEED, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model(), gap=None if which == 'gap' else (0, 0))
    keep = model.init_opt

    def init_opt(data, opt_variables, opt_lr):
        if eps:
            rng = np.random.RandomState(seed)
            cp = data['cam_pose']
            cp.mul_(torch.from_numpy((1 + eps * rng.uniform(-1, 1, tuple(cp.shape))).astype(np.float32)))
        optimizer, param_list = keep(data, opt_variables, opt_lr)
        trace['params'] = param_list
        trace['names'] = mg._param_names(model, data, opt_variables)
        trace['shapes'] = [tuple(p.shape) for p in param_list]
        trace['p0'] = np.concatenate([p.detach().numpy().ravel() for p in param_list])
        return optimizer, param_list
    model.init_opt = init_opt
    data, init_state = mg.run_reference(model, cfg.opt_stage_specs, in_dict, mg.latents_for(in_dict, DATA_SEED))
    fin = mg._flatten_state(data, mg.PERSON_KEYS_OPT + ['vis_frames'], mg.TOP_KEYS)
    np.savez_compressed(out, p=np.stack(trace['p']), g=np.stack(trace['g']), p0=trace['p0'], names=np.array(trace['names']),
                        sizes=np.array([int(np.prod(s)) for s in trace['shapes']]), **{'fin_' + k: v for k, v in fin.items()})
    print('wrote', out, np.stack(trace['p']).shape)


if __name__ == '__main__':
    main(sys.argv[1:])
