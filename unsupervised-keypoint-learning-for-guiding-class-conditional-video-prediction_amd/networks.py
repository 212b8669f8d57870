"""Network builders with the reference's names and scopes (reference: models/networks/__init__.py).

Same layer order, filter schedule and variable scopes as the reference; differences are fusions only:
BN+ReLU in one kernel, resize written straight into the skip-concat buffer, one 4-channel conv for the crude+mask
heads, and ``bn_groups`` so that weight-sharing calls on different images run as one batched launch with separate
batch-norm statistics per call.
"""
from . import layers, ops
from . import model_utils
from .layers import ACT_LRELU, ACT_NONE
import os as _os

from .variables import Sym, default_store, is_sym

# pose_encoder's 1x1 head and the two get_coord reductions as ONE op that never writes the [B,H,W,K] logits (ops.KeypointHeadProjFn);
# False keeps the conv + head pair (also used whenever the caller asks for the logits); a module constant the tests flip
FUSE_KEYPOINT_HEAD = True


def _upsample_concat(x, skip):
    if is_sym(x):
        n, h, w, c = x.shape
        return Sym(n, 2 * h, 2 * w, c + (skip.shape[-1] if skip is not None else 0))
    return ops.upsample2x_concat(x, skip)


def encoder(x, train_mode, filters=32, bn_groups=1, skip_last_block=False, update_moving=True):
    """reference encoder (networks/__init__.py:7-26).  Returns the 4 block features."""
    st = default_store()
    with st.variable_scope('encoder'):
        feats = []
        # f43_fwd=False on every encoder / detector layer: their forward stays on F(2x2,3x3) (ops.WINO43: the float64-arbiter finding)
        x = layers.conv_bn_relu(x, filters, 7, 1, train_mode, 'conv_1', 'b_norm_1', bn_groups, update_moving=update_moving, f43_fwd=False)
        x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_2', 'b_norm_2', bn_groups, update_moving=update_moving, f43_fwd=False)
        feats.append(x)
        for i in range(3):
            filters *= 2
            x = layers.conv_bn_relu(x, filters, 3, 2, train_mode, 'conv_%d' % (i * 2 + 3), 'b_norm_%d' % (i * 2 + 3), bn_groups, update_moving=update_moving, f43_fwd=False)
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d' % (i * 2 + 4), 'b_norm_%d' % (i * 2 + 4), bn_groups, update_moving=update_moving, f43_fwd=False)
            feats.append(x)
        return feats


def image_encoder(x, train_mode, update_moving=True):
    """reference image_encoder (:29-33): [x] + the 4 encoder blocks."""
    st = default_store()
    with st.variable_scope('image_encoder'):
        return [x] + encoder(x, train_mode, update_moving=update_moving)


def pose_encoder(x, n_pts, train_mode, final_res=128, filters=128, bn_groups=1, return_logits=False, update_moving=True):
    """reference pose_encoder (:36-72): encoder + U-Net decoder + 1x1 head -> separable-softmax key-points [B,K,2]."""
    st = default_store()
    with st.variable_scope('pose_encoder'):
        block_features = encoder(x, train_mode, bn_groups=bn_groups, update_moving=update_moving)
        x = block_features[-1]
        size = x.shape[1]
        conv_id = 1
        for i in range(4):
            # i > 0: x is already the [up-sampled ‖ skip] concat buffer written by the previous stage (:44)
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_0' % conv_id, 'b_norm_%d_0' % conv_id, bn_groups, update_moving=update_moving, f43_fwd=False)
            # (the last block's output is read by the fp32 key-point head: it stays fp32 in the bf16 configuration too)
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_1' % conv_id, 'b_norm_%d_1' % conv_id, bn_groups, update_moving=update_moving, f43_fwd=False,
                                    out_f32=size == final_res)
            if size == final_res:
                if FUSE_KEYPOINT_HEAD and not return_logits and (is_sym(x) or (x.is_cuda and ops.keypoint_head_proj_eligible(x.shape, n_pts))):
                    # 1x1 head (:54) + get_coord x2 (:68-71) as one op: the logits are consumed by the two axis means only
                    gauss_mu, _, _ = layers.conv1x1_keypoints(x, n_pts)
                    return gauss_mu
                x = layers.conv(x, n_pts, kernel=1, stride=1, out_f32=True)        # default scope 'conv_0' (:54)
                break
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_0' % (conv_id + 1), 'b_norm_%d_0' % (conv_id + 1), bn_groups, update_moving=update_moving, f43_fwd=False)
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_1' % (conv_id + 1), 'b_norm_%d_1' % (conv_id + 1), bn_groups, update_moving=update_moving, f43_fwd=False)
            x = _upsample_concat(x, block_features[-1 * (i + 2)])     # resize (:63) + next stage's concat (:44)
            size = 2 * size
            conv_id += 2
            if filters >= 8:
                filters //= 2
        gauss_mu, _, _ = model_utils.get_coords_xy(x)                 # :68-71
        return (gauss_mu, x) if return_logits else gauss_mu


def translator(x, train_mode, final_res=128, filters=256, cin=None, update_moving=True):
    """reference translator (:75-102).  Returns the fused 4-channel head output raw4 = crude(3) ‖ mask-logit(1);
    sigmoid + blend happen in ops.head_blend."""
    st = default_store()
    with st.variable_scope('translator'):
        size = x.shape[1]
        conv_id = 1
        while size <= final_res:
            # the first stage (conv_1_*, conv_2_*: 256-deep sums over the fewest pixels) keeps F(2x2,3x3) in the forward direction (ops.WINO43)
            f43 = conv_id > 1
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_0' % conv_id, 'b_norm_%d_0' % conv_id, cin=cin, update_moving=update_moving, f43_fwd=f43)
            cin = None
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_1' % conv_id, 'b_norm_%d_1' % conv_id, update_moving=update_moving, f43_fwd=f43)
            if size == final_res:
                # conv_N_0 (crude, 3 ch, :87) and conv_N_1 (mask, 1 ch, :88) as one 4-channel conv
                return layers.conv(x, 4, kernel=3, stride=1, scope='conv_%d_0+1' % (conv_id + 1), head31=True, out_f32=True)
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_0' % (conv_id + 1), 'b_norm_%d_0' % (conv_id + 1), update_moving=update_moving, f43_fwd=f43)
            x = layers.conv_bn_relu(x, filters, 3, 1, train_mode, 'conv_%d_1' % (conv_id + 1), 'b_norm_%d_1' % (conv_id + 1), update_moving=update_moving, f43_fwd=f43)
            x = _upsample_concat(x, None)                              # :98
            size = 2 * size
            conv_id += 2
            if filters >= 8:
                filters //= 2
    raise ValueError('translator input (%d) larger than final_res (%d)' % (size, final_res))


def img_discr(x):
    """reference img_discr (:141-151): 6x [pad 1 + conv4x4 s2 + bias + leaky_relu(0.01)] + D_logit 3x3 (no bias)."""
    st = default_store()
    with st.variable_scope('img_discr'):
        channel = 64
        # a plain chain: every activated tensor has exactly one consumer, the next conv, whose data gradient applies the leaky-ReLU backward
        # in its epilogue (layers.conv: act_bwd_by_consumer / input_act)
        x = layers.conv(x, channel, kernel=4, stride=2, pad=1, use_bias=True, scope='conv_0', act=ACT_LRELU, act_bwd_by_consumer=True)
        for i in range(1, 6):
            x = layers.conv(x, channel * 2, kernel=4, stride=2, pad=1, use_bias=True, scope='conv_' + str(i), act=ACT_LRELU,
                            input_act=ACT_LRELU, act_bwd_by_consumer=True)
            channel = channel * 2
        return layers.conv(x, channels=1, kernel=3, stride=1, pad=1, use_bias=False, scope='D_logit', act=ACT_NONE, input_act=ACT_LRELU, out_f32=True)


def vae_decoder(x, f_pt, act_code, cell_info, vae_dim, n_pts, n_steps=32):
    """reference vae_decoder (networks/__init__.py:116-129): z ‖ first key-points ‖ action code -> fc(32, relu) -> stacked
    LSTM, the fc output as input at step 0 and zeros afterwards, to_coord(tanh) per step -> [B, 32, 2*n_pts]."""
    st = default_store()
    with st.variable_scope('vae_decoder'):
        cell = layers.lstm_model(cell_info)
        if is_sym(x):
            inp = layers.fully_connected(Sym(x.shape[0], x.shape[1] + f_pt.shape[1] + act_code.shape[1]), 32)
            cell.declare(32)
            layers.to_coord(Sym(x.shape[0], cell_info[-1]), cell_info[-1], n_pts * 2)
            return Sym(x.shape[0], n_steps, n_pts * 2)
        import torch
        b = x.shape[0]
        cat = torch.empty((b, x.shape[1] + f_pt.shape[1] + act_code.shape[1]), dtype=torch.float32, device=x.device)
        off = 0
        for part in (x, f_pt, act_code):                                  # tf.concat([x, f_pt, act_code], -1) (:120)
            part = part.contiguous()
            ops.copy_channels_raw(part.data_ptr(), part.shape[1], cat.data_ptr() + 4 * off, cat.shape[1], b, part.shape[1])
            off += part.shape[1]
        input_ = layers.fully_connected(cat, 32)
        empty_input = torch.zeros_like(input_)                            # :121
        state = cell.zero_state(b, x.device)
        outputs = torch.empty((b, n_steps, n_pts * 2), dtype=torch.float32, device=x.device)
        for i in range(n_steps):                                          # :123-128
            output, state = cell(input_ if i == 0 else empty_input, state)
            o = layers.to_coord(output, cell_info[-1], n_pts * 2)
            ops.copy_channels_raw(o.data_ptr(), n_pts * 2, outputs.data_ptr() + 4 * i * n_pts * 2, n_steps * n_pts * 2, b, n_pts * 2)
        return outputs


# ------------------------------------------------------------------------------------------------ stage-2 training graphs
def _cat_features(parts):
    """tf.concat(parts, -1) for [B, n] tensors (torch glue: tiny, differentiable)."""
    import torch
    return torch.cat([p_ for p_ in parts], dim=-1)


def vae_encoder(x, f_pt, act_code, cell_info, vae_dim):
    """reference vae_encoder (networks/__init__.py:105-114): dynamic_rnn over x [B,32,2K], last output ++ first point ++ action
    code -> fully_connected(2*vae_dim) with the contrib default ReLU -> logit = [mu | stddev]  [B, 2*vae_dim]."""
    st = default_store()
    with st.variable_scope('vae_encoder'):
        cell = layers.lstm_model(cell_info)
        if is_sym(x):
            with st.variable_scope('rnn'):
                cell.declare(x.shape[-1])
            return layers.fully_connected(Sym(x.shape[0], cell_info[-1] + f_pt.shape[1] + act_code.shape[1]), vae_dim * 2)
        with st.variable_scope('rnn'):                                    # dynamic_rnn's variable scope
            out = cell.sequence(x.transpose(0, 1).contiguous())            # [T,B,U]
        return layers.fully_connected(_cat_features([out[-1], f_pt, act_code]), vae_dim * 2, trainable=True)


def vae_decoder_train(z, f_pt, act_code, cell_info, vae_dim, n_pts, n_steps=32):
    """Differentiable vae_decoder (same variables as ``vae_decoder``): the LSTM input is the fc output at step 0 and zeros
    afterwards (:121-128), so the stack runs as whole-sequence layers and to_coord is one GEMM over all T*B rows."""
    import torch
    st = default_store()
    with st.variable_scope('vae_decoder'):
        cell = layers.lstm_model(cell_info)
        inp = layers.fully_connected(_cat_features([z, f_pt, act_code]), 32, trainable=True)               # [B,32]
        x_seq = torch.cat([inp.unsqueeze(0), torch.zeros((n_steps - 1,) + tuple(inp.shape), dtype=inp.dtype, device=inp.device)], dim=0)
        out = cell.sequence(x_seq)                                                                         # [T,B,U]
        t, b, u = out.shape
        coords = layers.to_coord(out.reshape(t * b, u), cell_info[-1], n_pts * 2, trainable=True)           # [T*B, 2K]
        return coords.reshape(t, b, n_pts * 2).transpose(0, 1)                                             # [B,T,2K]


def seq_discr(x, discr_cells=(1024, 1024)):
    """reference seq_discr (networks/__init__.py:132-138): 2-layer LSTM(1024), fully_connected(outputs, 1) (ReLU), logit of the
    last step [B,1].  Only the last step's fc output is used, so only that row block is computed."""
    st = default_store()
    with st.variable_scope('seq_discr'):
        cell = layers.lstm_model(list(discr_cells))
        if is_sym(x):
            with st.variable_scope('rnn'):
                cell.declare(x.shape[-1])
            return layers.fully_connected(Sym(x.shape[0], discr_cells[-1]), 1)
        with st.variable_scope('rnn'):
            out = cell.sequence(x.transpose(0, 1).contiguous())
        return layers.fully_connected(out[-1], 1, trainable=True)
