"""oracle/randla_train_oracle.py -- TEST INFRASTRUCTURE ONLY.

Training-mode restatement of the PointSegment graph in torch-CPU float64 with autograd as the gradient yardstick:
forward of PointSegment/RandLANet.py:110-152, 314-401 with tf.layers.batch_normalization(training=True)
(helper_tf_util.py:167,246: batch mean / population variance over all but the channel axis, eps 1e-6), the
class-weighted softmax cross-entropy of RandLANet.py:267-274 (mean over points), one tf.train.AdamOptimizer step
(RandLANet.py:89; lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps 1e-8) and the moving-statistics update (momentum 0.99).
Dropout is not modelled (tests run with keep_prob = 1; TF's RNG stream cannot be reproduced).
PARITY UNPINNED for the same reason as oracle/randla_oracle.py (TensorFlow 1.11 is not installable here).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-6


def train_step(params, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, labels, class_weights, lr, step=1):
    """Returns dict(loss, grads{name: array}, new_params{name: array}, logits)."""
    P = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=torch.float64)
        if not k.endswith(("moving_mean", "moving_variance")):
            t.requires_grad_(True)
        P[k] = t
    new_buf = {}

    def bn(x, s):
        flat = x.reshape(-1, x.shape[-1])
        mean = flat.mean(0)
        var = flat.var(0, unbiased=False)
        new_buf[s + "/moving_mean"] = 0.99 * P[s + "/moving_mean"] + 0.01 * mean.detach()
        new_buf[s + "/moving_variance"] = 0.99 * P[s + "/moving_variance"] + 0.01 * var.detach()
        return (x - mean) / torch.sqrt(var + BN_EPS) * P[s + "/gamma"] + P[s + "/beta"]

    def conv(x, s, use_bn=True, act=True):
        y = x @ P[s + "/weights"] + P[s + "/biases"]
        if use_bn:
            y = bn(y, s + "/batch_normalization")
        return F.leaky_relu(y, 0.2) if act else y

    def deconv(x, s):
        return F.leaky_relu(bn(x @ P[s + "/weights"].T + P[s + "/biases"], s + "/batch_normalization"), 0.2)

    def gather(pc, idx):
        idx = torch.as_tensor(np.asarray(idx)).long()
        return torch.stack([pc[b][idx[b]] for b in range(pc.shape[0])])

    def att(fset, name):
        s = F.softmax(fset @ P[name + "fc/kernel"], dim=2)
        return conv((fset * s).sum(2), name + "mlp")

    f = torch.tensor(features, dtype=torch.float64) @ P["fc0/kernel"] + P["fc0/bias"]
    f = F.leaky_relu(bn(f, "batch_normalization"), 0.2)
    enc = []
    for i in range(num_layers):
        n = "Encoder_layer_%d" % i
        X, idx = f, neigh_idx[i]
        xyz_i = torch.tensor(xyz[i], dtype=torch.float64)
        nb = gather(xyz_i, idx)
        ctr = xyz_i[:, :, None, :].expand_as(nb)
        rel = ctr - nb
        enc10 = torch.cat([rel.pow(2).sum(-1, keepdim=True).sqrt(), rel, ctr, nb], -1)
        f_pc = conv(X, n + "mlp1")
        f_xyz = conv(enc10, n + "LFAmlp1")
        f_agg = att(torch.cat([gather(f_pc, idx), f_xyz], -1), n + "LFAatt_pooling_1")
        f_xyz = conv(f_xyz, n + "LFAmlp2")
        f_agg = att(torch.cat([gather(f_agg, idx), f_xyz], -1), n + "LFAatt_pooling_2")
        f_enc = F.leaky_relu(conv(f_agg, n + "mlp2", act=False) + conv(X, n + "shortcut", act=False), 0.2)
        f = torch.amax(gather(f_enc, sub_idx[i]), dim=2)  # ties share the gradient evenly, like tf.reduce_max
        if i == 0:
            enc.append(f_enc)
        enc.append(f)
    f = conv(enc[-1], "decoder_0")
    for j in range(num_layers):
        f = deconv(torch.cat([enc[-j - 2], gather(f, interp_idx[-j - 1])[:, :, 0]], -1), "Decoder_layer_%d" % j)
    logits = conv(conv(conv(f, "fc1"), "fc2"), "fc", use_bn=False, act=False)
    z = logits.reshape(-1, logits.shape[-1])
    y = torch.as_tensor(np.asarray(labels).reshape(-1)).long()
    w = torch.tensor(np.asarray(class_weights, np.float64).reshape(-1))[y]
    loss = (F.cross_entropy(z, y, reduction="none") * w).mean()
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
    return dict(loss=float(loss), grads=grads, new_params=new_params, logits=logits.detach().numpy())
