"""Parameters of the PointSegment RandLA-Net: initialisation, inference-mode BatchNorm folding and the flat
weight blob that ps_randla_set_weights expects.

Parameter names are the reference's TF variable names without the leading "layers/" scope (scope strings are
concatenated without separators, PointSegment/RandLANet.py:121, 315-334, 395, 400):

    fc0/{kernel,bias}                                        tf.layers.dense           RandLANet.py:114
    batch_normalization/{gamma,beta,moving_mean,moving_variance}   fc0's BN            RandLANet.py:115
    Encoder_layer_{i}{mlp1,LFAmlp1,LFAatt_pooling_1mlp,LFAmlp2,LFAatt_pooling_2mlp,mlp2,shortcut}/
        {weights,biases,batch_normalization/*}               helper_tf_util.conv2d     helper_tf_util.py:148-170
    Encoder_layer_{i}LFAatt_pooling_{1,2}fc/kernel           tf.layers.dense, no bias  RandLANet.py:395
    decoder_0/...                                            conv2d                    RandLANet.py:130-132
    Decoder_layer_{j}/{weights [out,in],biases,batch_normalization/*}   conv2d_transpose   helper_tf_util.py:208-250
    fc1/..., fc2/..., fc/{weights,biases}                    conv2d (fc: no BN, no act) RandLANet.py:146-150

Conv kernels are stored squeezed: conv2d [Cin,Cout] (TF [1,1,Cin,Cout]); conv2d_transpose [Cout,Cin]
(TF [1,1,Cout,Cin]).
"""
import numpy as np

BN_EPS = 1e-6


def layer_dims(cfg):
    """[(scope, kind, cin, cout)] in graph order. kind: dense | dense_nobias | conv | conv_nobn | deconv"""
    L = cfg.num_layers
    d_out = list(cfg.d_out)[:L]
    out = [("fc0", "dense", cfg.in_channels, 8)]
    d_in = 8
    for i in range(L):
        d, h = d_out[i], d_out[i] // 2
        n = "Encoder_layer_%d" % i
        out += [
            (n + "mlp1", "conv", d_in, h),
            (n + "LFAmlp1", "conv", 10, h),
            (n + "LFAatt_pooling_1fc", "dense_nobias", d, d),
            (n + "LFAatt_pooling_1mlp", "conv", d, h),
            (n + "LFAmlp2", "conv", h, h),
            (n + "LFAatt_pooling_2fc", "dense_nobias", d, d),
            (n + "LFAatt_pooling_2mlp", "conv", d, d),
            (n + "mlp2", "conv", d, 2 * d),
            (n + "shortcut", "conv", d_in, 2 * d),
        ]
        d_in = 2 * d
    out.append(("decoder_0", "conv", d_in, d_in))
    chans = [2 * d_out[0]] + [2 * d for d in d_out]  # f_encoder_list channel widths (RandLANet.py:119-127)
    up = d_in
    for j in range(L):
        skip = chans[-j - 2]
        out.append(("Decoder_layer_%d" % j, "deconv", skip + up, skip))
        up = skip
    out += [("fc1", "conv", up, 64), ("fc2", "conv", 64, 32), ("fc", "conv_nobn", 32, cfg.num_classes)]
    return out


def _truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: samples beyond 2 sigma are re-drawn."""
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return x * stddev


def init_params(cfg, seed=0, randomize_bn=False):
    """Random parameters following the reference's initialisers (helper_tf_util.py:26-55):
    conv kernels round(truncated_normal(sigma = sqrt(2 / shape[-1])) * 1000) / 1000, biases 0;
    tf.layers.dense kernels glorot-uniform, bias 0; BatchNorm gamma 1, beta 0, moving mean 0, moving var 1.
    `randomize_bn` draws non-trivial BN statistics instead (what a trained checkpoint looks like) so that the
    folding path is actually exercised by the parity tests."""
    rng = np.random.default_rng(seed)
    p = {}

    def bn(scope, c):
        if randomize_bn:
            p[scope + "/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            p[scope + "/beta"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            p[scope + "/moving_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            p[scope + "/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        else:
            p[scope + "/gamma"] = np.ones(c, np.float32)
            p[scope + "/beta"] = np.zeros(c, np.float32)
            p[scope + "/moving_mean"] = np.zeros(c, np.float32)
            p[scope + "/moving_variance"] = np.ones(c, np.float32)

    for scope, kind, cin, cout in layer_dims(cfg):
        if kind in ("dense", "dense_nobias"):
            lim = np.sqrt(6.0 / (cin + cout))
            p[scope + "/kernel"] = rng.uniform(-lim, lim, (cin, cout)).astype(np.float32)
            if kind == "dense":
                p[scope + "/bias"] = np.zeros(cout, np.float32)
                bn("batch_normalization", cout)  # fc0's un-scoped BN (RandLANet.py:115)
        else:
            shape = (cout, cin) if kind == "deconv" else (cin, cout)  # TF [1,1,out,in] vs [1,1,in,out]
            w = _truncated_normal(rng, shape, np.sqrt(2.0 / shape[-1]))
            p[scope + "/weights"] = (np.round(w * 1000.0) / 1000.0).astype(np.float32)
            p[scope + "/biases"] = np.zeros(cout, np.float32)
            if kind != "conv_nobn":
                bn(scope + "/batch_normalization", cout)
    return p


def _fold(w_in_out, b, p, bn_scope):
    """y = BN(x.W + b)  ==  x.(W*s) + ((b - mean)*s + beta),  s = gamma / sqrt(var + eps); float64 then fp32."""
    w = w_in_out.astype(np.float64)
    b = b.astype(np.float64)
    if bn_scope is not None:
        s = p[bn_scope + "/gamma"].astype(np.float64) / np.sqrt(p[bn_scope + "/moving_variance"].astype(np.float64) + BN_EPS)
        w = w * s[None, :]
        b = (b - p[bn_scope + "/moving_mean"].astype(np.float64)) * s + p[bn_scope + "/beta"].astype(np.float64)
    return w.astype(np.float32), b.astype(np.float32)


def fold_to_blob(cfg, p):
    """The flat fp32 blob of ps_randla_set_weights: for every layer in layer_dims() order, W[cin,cout] row-major
    followed by b[cout], BatchNorm folded in (csrc/randla.hip: plan_specs)."""
    parts = []
    for scope, kind, cin, cout in layer_dims(cfg):
        if kind == "dense":
            w, b = _fold(p[scope + "/kernel"], p[scope + "/bias"], p, "batch_normalization")
        elif kind == "dense_nobias":
            w, b = p[scope + "/kernel"].astype(np.float32), np.zeros(cout, np.float32)
        elif kind == "deconv":
            w, b = _fold(p[scope + "/weights"].T, p[scope + "/biases"], p, scope + "/batch_normalization")
        elif kind == "conv_nobn":
            w, b = _fold(p[scope + "/weights"], p[scope + "/biases"], p, None)
        else:
            w, b = _fold(p[scope + "/weights"], p[scope + "/biases"], p, scope + "/batch_normalization")
        assert w.shape == (cin, cout) and b.shape == (cout,), (scope, w.shape, b.shape)
        parts += [np.ascontiguousarray(w).ravel(), b]
    return np.concatenate(parts).astype(np.float32)


def num_params(cfg):
    """Trainable parameter count (kernels, biases, BN gamma/beta)."""
    n = 0
    for scope, kind, cin, cout in layer_dims(cfg):
        n += cin * cout
        if kind != "dense_nobias":
            n += cout
        if kind in ("dense", "conv", "deconv"):
            n += 2 * cout
    return n


def from_tf_variables(cfg, variables):
    """Maps a dict of TF-1.x checkpoint variables of the reference model (as `tf.train.load_checkpoint(...)` would give
    them: names under the 'layers/' scope, conv kernels [1,1,in,out] / [1,1,out,in], optimizer slots such as
    '.../Adam' mixed in, RandLANet.py:56,101-102) onto this package's parameter dict.  Missing variables raise KeyError."""
    out = {}
    src = {}
    for k, v in variables.items():
        k = k[len("layers/"):] if k.startswith("layers/") else k
        if k.endswith((":0",)):
            k = k[:-2]
        src[k] = np.asarray(v)
    for scope, kind, cin, cout in layer_dims(cfg):
        if kind in ("dense", "dense_nobias"):
            out[scope + "/kernel"] = src[scope + "/kernel"].astype(np.float32).reshape(cin, cout)
            if kind == "dense":
                out[scope + "/bias"] = src[scope + "/bias"].astype(np.float32)
                for n in ("gamma", "beta", "moving_mean", "moving_variance"):
                    out["batch_normalization/" + n] = src["batch_normalization/" + n].astype(np.float32)
        else:
            w = src[scope + "/weights"].astype(np.float32)
            out[scope + "/weights"] = w.reshape((cout, cin) if kind == "deconv" else (cin, cout))
            out[scope + "/biases"] = src[scope + "/biases"].astype(np.float32)
            if kind != "conv_nobn":
                for n in ("gamma", "beta", "moving_mean", "moving_variance"):
                    out[scope + "/batch_normalization/" + n] = src[scope + "/batch_normalization/" + n].astype(np.float32)
    return out
