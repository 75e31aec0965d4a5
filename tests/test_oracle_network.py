"""Independent cross-check of the network restatement: the same graph written with torch-CPU ops
(conv as F.linear, batch_norm as F.batch_norm in eval mode, softmax as F.softmax) must agree with
oracle/randla_oracle.py.  This does not pin the oracle to TensorFlow (unavailable) but guards the formulas."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import netcase


def _torch_forward(p, L, xyz, nbr, pool, up, feats):
    t = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731
    P = {k: t(v).double() for k, v in p.items()}

    def bn(x, s):
        return F.batch_norm(x.reshape(-1, x.shape[-1]), P[s + "/moving_mean"], P[s + "/moving_variance"], P[s + "/gamma"], P[s + "/beta"],
                            False, 0.0, 1e-6).reshape(x.shape)

    def conv(x, s, use_bn=True, act=True):
        y = F.linear(x, P[s + "/weights"].T, P[s + "/biases"])
        if use_bn:
            y = bn(y, s + "/batch_normalization")
        return F.leaky_relu(y, 0.2) if act else y

    def deconv(x, s):
        return F.leaky_relu(bn(F.linear(x, P[s + "/weights"], P[s + "/biases"]), s + "/batch_normalization"), 0.2)

    def gather(pc, idx):
        return torch.stack([pc[b][idx[b].long()] for b in range(pc.shape[0])])

    def att(fset, name):
        s = F.softmax(fset @ P[name + "fc/kernel"], dim=2)
        return conv((fset * s).sum(2), name + "mlp")

    f = F.leaky_relu(bn(F.linear(t(feats).double(), P["fc0/kernel"].T, P["fc0/bias"]), "batch_normalization"), 0.2)
    enc = []
    for i in range(L):
        n = "Encoder_layer_%d" % i
        X, idx = f, t(nbr[i])
        xyz_i = t(xyz[i]).double()
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
        f = gather(f_enc, t(pool[i])).max(2).values
        if i == 0:
            enc.append(f_enc)
        enc.append(f)
    f = conv(enc[-1], "decoder_0")
    for j in range(L):
        f = deconv(torch.cat([enc[-j - 2], gather(f, t(up[-j - 1]))[:, :, 0]], -1), "Decoder_layer_%d" % j)
    return conv(conv(conv(f, "fc1"), "fc2"), "fc", use_bn=False, act=False).numpy()


def test_numpy_restatement_agrees_with_torch_cpu(oracle):
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    cfg, xyz, feats = netcase.small_deep(1200, seed=4)
    cfg.d_out = [16, 32, 32, 16, 16]
    params = weights.init_params(cfg, seed=9, randomize_bn=True)
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    a = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64)
    b = _torch_forward(params, cfg.num_layers, pts, nbr, pool, up, feats)
    assert np.abs(a - b).max() < 1e-10


def test_loss_restatement():
    from oracle import randla_oracle as ro
    rng = np.random.default_rng(0)
    z = rng.standard_normal((50, 4))
    y = rng.integers(0, 4, 50)
    w = np.array([2.0, 3.0, 4.0, 5.0])
    want = float((F.cross_entropy(torch.from_numpy(z), torch.from_numpy(y), reduction="none") * torch.from_numpy(w)[y]).mean())
    assert abs(ro.weighted_ce_loss(z, y, w) - want) < 1e-12


def test_torch_oracle_forward_and_training_variants(oracle):
    """oracle/randla_train_oracle.py: (1) its eval-mode forward (bench.py's all-cores CPU-baseline leg runs it in float32) equals the
    NumPy restatement; (2) ignored labels drop out of the loss and its mean (RandLANet.py:62-84); (3) the bf16 operand rounding
    changes the step by bf16-sized amounts only where the rule says so."""
    from oracle import randla_oracle as ro
    from oracle import randla_train_oracle as rto
    from point_unet_amd import weights
    cfg, xyz, feats = netcase.small_deep(1200, seed=4)
    cfg.d_out = [16, 32, 32, 16, 16]
    params = weights.init_params(cfg, seed=9, randomize_bn=True)
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    a = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64)
    assert np.abs(rto.forward(params, cfg.num_layers, pts, nbr, pool, up, feats, torch.float64) - a).max() < 1e-10
    assert np.abs(rto.forward(params, cfg.num_layers, pts, nbr, pool, up, feats, torch.float32) - a).max() < 1e-3
    rng = np.random.default_rng(1)
    labels = rng.integers(0, 4, xyz.shape[:2])
    cw = np.array([1.0, 2.0, 0.5, 3.0])
    full = rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, labels, cw, lr=1e-3)
    masked = labels.copy()
    masked[:, ::3] = -1
    part = rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, masked, cw, lr=1e-3)
    z = torch.from_numpy(part["logits"].reshape(-1, 4))
    y = torch.from_numpy(masked.reshape(-1))
    keep = y >= 0
    want = float((F.cross_entropy(z[keep], y[keep], reduction="none") * torch.from_numpy(cw)[y[keep]]).mean())
    assert abs(part["loss"] - want) < 1e-12 and abs(part["loss"] - full["loss"]) > 1e-6
    none = rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, labels, cw, lr=1e-3, bf16_rule=lambda kind, cin, cout: False)
    assert none["loss"] == full["loss"]
    bf = rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, labels, cw, lr=1e-3, bf16_rule=lambda kind, cin, cout: True)
    assert 1e-6 < abs(bf["loss"] - full["loss"]) < 3e-2 * abs(full["loss"])
