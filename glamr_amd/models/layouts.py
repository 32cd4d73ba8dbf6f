"""Checkpoint layouts (state_dict key -> shape) of the two motion priors, fp32, PyTorch `(out, in)` weight convention.

These are the key names a Lightning `.ckpt` of the reference carries, i.e. the contract between a
trained GLAMR checkpoint and this package's weight packer.  Reference construction sites:
motion_infiller/models/motion_infiller_vae.py:22-90,126-202,252-342 and traj_pred/models/traj_pred_vae.py:20-70,95-157,202-266
with the shipped specs motion_infiller/cfg/motion_infiller_demo.yml and traj_pred/cfg/traj_pred_demo.yml.
"""

D, FF, NZ = 256, 512, 128


def _linear(prefix, out_dim, in_dim):
    return [(prefix + '.weight', (out_dim, in_dim)), (prefix + '.bias', (out_dim,))]


def _attn(prefix):
    return [(prefix + '.in_proj_weight', (3 * D, D)), (prefix + '.in_proj_bias', (3 * D,))] + _linear(prefix + '.out_proj', D, D)


def _enc_layer(prefix):
    out = _attn(prefix + '.self_attn') + _linear(prefix + '.linear1', FF, D) + _linear(prefix + '.linear2', D, FF)
    for n in ('norm1', 'norm2'):
        out += [(prefix + '.%s.weight' % n, (D,)), (prefix + '.%s.bias' % n, (D,))]
    return out


def _dec_layer(prefix):
    out = _attn(prefix + '.self_attn') + _attn(prefix + '.multihead_attn')
    out += _linear(prefix + '.linear1', FF, D) + _linear(prefix + '.linear2', D, FF)
    for n in ('norm1', 'norm2', 'norm3'):
        out += [(prefix + '.%s.weight' % n, (D,)), (prefix + '.%s.bias' % n, (D,))]
    return out


def _mlp(prefix, in_dim, hdims):
    out = []
    for i, h in enumerate(hdims):
        out += _linear(prefix + '.affine_layers.%d' % i, h, in_dim)
        in_dim = h
    return out


def infiller_layout():
    L = _linear('context_encoder.in_fc', D, 69) + _linear('context_encoder.pos_enc.fc', D, 2 * D)
    for i in range(2):
        L += _enc_layer('context_encoder.temporal_net.layers.%d' % i)
    L += [('data_encoder.mu_token', (D,)), ('data_encoder.logvar_token', (D,))]
    L += _linear('data_encoder.in_fc', D, 69) + _linear('data_encoder.pos_enc.fc', D, 2 * D)
    for i in range(2):
        L += _dec_layer('data_encoder.temporal_net.layers.%d' % i)
    L += _linear('data_encoder.q_z_mu_net', NZ, D) + _linear('data_encoder.q_z_logvar_net', NZ, D)
    L += [('data_decoder.mu_token', (D,)), ('data_decoder.logvar_token', (D,))]
    L += _linear('data_decoder.pos_enc.fc', D, NZ + D)
    for i in range(2):
        L += _dec_layer('data_decoder.temporal_net.layers.%d' % i)
    L += _mlp('data_decoder.out_mlp', D, (FF, D)) + _linear('data_decoder.out_fc', 69, D)
    L += _linear('data_decoder.prior_pos_enc.fc', D, 2 * D)
    L += _dec_layer('data_decoder.prior_temporal_net.layers.0')
    L += _linear('data_decoder.p_z_mu_net', NZ, D) + _linear('data_decoder.p_z_logvar_net', NZ, D)
    return L


def _bilstm(prefix, in_dim, hid=128):
    out = []
    for d in ('rnn_f', 'rnn_b'):
        out += [(prefix + '.%s.weight_ih' % d, (4 * hid, in_dim)), (prefix + '.%s.weight_hh' % d, (4 * hid, hid)),
                (prefix + '.%s.bias_ih' % d, (4 * hid,)), (prefix + '.%s.bias_hh' % d, (4 * hid,))]
    return out


def trajpred_layout():
    L = _mlp('context_encoder.in_mlp', 69, (FF, D))
    for i in range(2):
        L += _bilstm('context_encoder.temporal_net.%d' % i, D)
    L += _mlp('context_encoder.out_mlp', D, (FF, D))
    L += _mlp('data_encoder.in_mlp', 6, (FF, D))
    for i in range(2):
        L += _bilstm('data_encoder.temporal_net.%d' % i, D)
    L += _mlp('data_encoder.out_mlp', D, (FF, D)) + _mlp('data_encoder.fusion_mlp', 2 * D, (FF, D))
    L += _linear('data_encoder.q_z_net', 2 * NZ, D)
    L += _mlp('data_decoder.out_mlp', D + NZ, (FF, D)) + _linear('data_decoder.out_fc', 11, D)
    L += _mlp('data_decoder.prior_mlp', D, (FF, D)) + _linear('data_decoder.p_z_net', 2 * NZ, D)
    return L


INFILLER_LAYOUT = infiller_layout()
TRAJPRED_LAYOUT = trajpred_layout()
