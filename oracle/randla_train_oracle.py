"""oracle/randla_train_oracle.py -- TEST INFRASTRUCTURE ONLY.

The PointSegment graph in torch-CPU, float64 with autograd as the gradient yardstick (and float32 as the CPU-baseline leg of
bench.py):

    forward / train_step     PointSegment/RandLANet.py:110-152, 314-401
    BatchNorm                tf.layers.batch_normalization (helper_tf_util.py:167,246; RandLANet.py:115): training=True -> batch mean /
                             population variance over all but the channel axis, eps 1e-6, moving statistics momentum 0.99;
                             training=False -> moving statistics
    loss                     RandLANet.py:62-84 (rows whose label is ignored are dropped, mean over the valid ones) and :267-274
                             (class-weighted softmax cross-entropy)
    optimiser                one tf.train.AdamOptimizer step (RandLANet.py:89; lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps 1e-8)

Dropout is not modelled (tests run with keep_prob = 1; TF's RNG stream cannot be reproduced).
`bf16_rule(kind, cin, cout)` (kind in "fwd", "dgrad", "wgrad") selects the GEMMs whose OPERANDS are rounded to bfloat16
(round-to-nearest-even) before an exact product -- the yardstick for BASELINE configs[2]'s "bf16 MLPs" mode, whose product
kernels round the same operands and accumulate in fp32.  `act_rule(h)` selects the levels whose LFA rows (the LocSE output and LFA
mlp2's, [N, K, h]) are STORED as bfloat16 in that mode (ps_train_options.act_bf16): rounded once where they are produced.
PARITY UNPINNED for the same reason as oracle/randla_oracle.py (TensorFlow 1.11 is not installable here); the eval-mode
forward is cross-checked against the NumPy restatement in tests/test_oracle_network.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-6


def _rb(t):
    """bfloat16 rounding (RNE) of a tensor, kept in the tensor's own dtype."""
    return t.detach().float().bfloat16().to(t.dtype) if t.dtype != torch.float32 else t.detach().bfloat16().float()


class _StoreRounded(torch.autograd.Function):
    """A tensor STORED as bfloat16 (ps_train_options.act_bf16): consumers see the rounded value, and the rows of its GRADIENT are stored
    in the same format -- the producer's backward reads bfloat16(dy)."""

    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, dy):
        return _rb(dy)


class _GradRounded(torch.autograd.Function):
    """Identity whose GRADIENT is stored as bfloat16 rows: the gathered half of a pooling's neighbour set at the levels of act_rule (the rows
    the fused backward leaves for the gather-reduction, csrc/attpool_train.hip RowStore)."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, dy):
        return _rb(dy)


class _StoreRoundedFork(torch.autograd.Function):
    """The same for a stored tensor with TWO consumers (f_xyz: the first pooling and LFA mlp2, RandLANet.py:328-331): returns the rounded
    value twice; the gradient buffer is written by the convolution's backward (second output) and the pooling's backward then adds into
    the stored rows: bfloat16(bfloat16(g_conv) + g_pool)  (csrc/trainer.hip attpool_split / conv_bn_fused)."""

    @staticmethod
    def forward(ctx, x):
        r = _rb(x)
        return r, r.clone()

    @staticmethod
    def backward(ctx, g_pool, g_conv):
        return _rb(_rb(g_conv) + g_pool)


class _RoundedLinear(torch.autograd.Function):
    """y = x . W with the three GEMMs (forward, input gradient, weight gradient) each optionally on bf16-rounded operands."""

    @staticmethod
    def forward(ctx, x, W, rf, rd, rw):
        ctx.save_for_backward(x, W)
        ctx.flags = (rd, rw)
        return (_rb(x) @ _rb(W)) if rf else x @ W

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        rd, rw = ctx.flags
        x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
        dx = ((_rb(dy) @ _rb(W).T) if rd else dy @ W.T)
        dW = (_rb(x2).T @ _rb(dy2)) if rw else x2.T @ dy2
        return dx, dW, None, None, None


def _graph(P, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, dtype, training, new_buf, bf16_rule, act_rule=None):
    def lin(x, W):
        cin, cout = W.shape
        if bf16_rule is None:
            return x @ W
        return _RoundedLinear.apply(x, W, bool(bf16_rule("fwd", cin, cout)), bool(bf16_rule("dgrad", cin, cout)), bool(bf16_rule("wgrad", cin, cout)))

    def bn(x, s):
        if not training:
            return (x - P[s + "/moving_mean"]) / torch.sqrt(P[s + "/moving_variance"] + BN_EPS) * P[s + "/gamma"] + P[s + "/beta"]
        flat = x.reshape(-1, x.shape[-1])
        mean = flat.mean(0)
        var = flat.var(0, unbiased=False)
        new_buf[s + "/moving_mean"] = 0.99 * P[s + "/moving_mean"] + 0.01 * mean.detach()
        new_buf[s + "/moving_variance"] = 0.99 * P[s + "/moving_variance"] + 0.01 * var.detach()
        return (x - mean) / torch.sqrt(var + BN_EPS) * P[s + "/gamma"] + P[s + "/beta"]

    def conv(x, s, use_bn=True, act=True):
        y = lin(x, P[s + "/weights"]) + P[s + "/biases"]
        if use_bn:
            y = bn(y, s + "/batch_normalization")
        return F.leaky_relu(y, 0.2) if act else y

    def deconv(x, s):
        return F.leaky_relu(bn(lin(x, P[s + "/weights"].T) + P[s + "/biases"], s + "/batch_normalization"), 0.2)

    def gather(pc, idx):
        idx = torch.as_tensor(np.asarray(idx)).long()
        return torch.stack([pc[b][idx[b]] for b in range(pc.shape[0])])

    def att(fset, name):
        s = F.softmax(lin(fset, P[name + "fc/kernel"]), dim=2)
        return conv((fset * s).sum(2), name + "mlp")

    f = lin(torch.as_tensor(np.asarray(features)).to(dtype), P["fc0/kernel"]) + P["fc0/bias"]
    f = F.leaky_relu(bn(f, "batch_normalization"), 0.2)
    enc = []
    for i in range(num_layers):
        n = "Encoder_layer_%d" % i
        X, idx = f, neigh_idx[i]
        xyz_i = torch.as_tensor(np.asarray(xyz[i])).to(dtype)
        nb = gather(xyz_i, idx)
        ctr = xyz_i[:, :, None, :].expand_as(nb)
        rel = ctr - nb
        enc10 = torch.cat([rel.pow(2).sum(-1, keepdim=True).sqrt(), rel, ctr, nb], -1)
        f_pc = conv(X, n + "mlp1")
        f_xyz = conv(enc10, n + "LFAmlp1")
        stored16 = act_rule is not None and bool(act_rule(f_xyz.shape[-1]))  # the level's [N, K, h] rows are stored as bfloat16
        f_xyz_pool = f_xyz
        if stored16:
            f_xyz_pool, f_xyz = _StoreRoundedFork.apply(f_xyz)
        gr = _GradRounded.apply if stored16 else (lambda t: t)
        f_agg = att(torch.cat([gr(gather(f_pc, idx)), f_xyz_pool], -1), n + "LFAatt_pooling_1")
        f_xyz = conv(f_xyz, n + "LFAmlp2")
        if stored16:
            f_xyz = _StoreRounded.apply(f_xyz)
        f_agg = att(torch.cat([gr(gather(f_agg, idx)), f_xyz], -1), n + "LFAatt_pooling_2")
        f_enc = F.leaky_relu(conv(f_agg, n + "mlp2", act=False) + conv(X, n + "shortcut", act=False), 0.2)
        f = torch.amax(gather(f_enc, sub_idx[i]), dim=2)  # ties share the gradient evenly, like tf.reduce_max
        if i == 0:
            enc.append(f_enc)
        enc.append(f)
    f = conv(enc[-1], "decoder_0")
    for j in range(num_layers):
        f = deconv(torch.cat([enc[-j - 2], gather(f, interp_idx[-j - 1])[:, :, 0]], -1), "Decoder_layer_%d" % j)
    return conv(conv(conv(f, "fc1"), "fc2"), "fc", use_bn=False, act=False)


def forward(params, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, dtype=torch.float64):
    """Inference-mode logits [B, N0, classes] (moving-statistics BatchNorm, no dropout), no autograd."""
    with torch.no_grad():
        P = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in params.items()}
        return _graph(P, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, dtype, False, {}, None).numpy()


def train_step(params, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, labels, class_weights, lr, step=1, dtype=torch.float64,
               bf16_rule=None, act_rule=None):
    """One optimisation step from zero Adam moments.  Labels outside [0, classes) mark ignored points (already renumbered by the
    caller like the reference's reducing_list).  Returns dict(loss, grads{name: array}, new_params{name: array}, logits)."""
    P = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if not k.endswith(("moving_mean", "moving_variance")):
            t.requires_grad_(True)
        P[k] = t
    new_buf = {}
    logits = _graph(P, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, dtype, True, new_buf, bf16_rule, act_rule)
    z = logits.reshape(-1, logits.shape[-1])
    y = torch.as_tensor(np.asarray(labels).reshape(-1)).long()
    valid = (y >= 0) & (y < z.shape[1])
    zv, yv = z[valid], y[valid]
    w = torch.tensor(np.asarray(class_weights, np.float64).reshape(-1)).to(dtype)[yv]
    loss = (F.cross_entropy(zv, yv, reduction="none") * w).mean()
    loss.backward()
    grads = {k: v.grad.numpy().copy() for k, v in P.items() if v.requires_grad}
    b1, b2, eps = 0.9, 0.999, 1e-8
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    new_params = {}
    for k, g in grads.items():  # first step from zero moments
        m = (1 - b1) * g
        v = (1 - b2) * g * g
        new_params[k] = P[k].detach().numpy() - lr_t * m / (np.sqrt(v) + eps)
    for k, v in new_buf.items():
        new_params[k] = v.numpy()
    return dict(loss=float(loss.detach()), grads=grads, new_params=new_params, logits=logits.detach().numpy())
