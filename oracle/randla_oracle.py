"""oracle/randla_oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h for the rules of use).

NumPy restatement of the PointSegment RandLA-Net graph, one function per reference method:

    inference               PointSegment/RandLANet.py:110-152
    dilated_res_block       PointSegment/RandLANet.py:314-321
    building_block          PointSegment/RandLANet.py:323-335
    relative_pos_encoding   PointSegment/RandLANet.py:337-343
    random_sample           PointSegment/RandLANet.py:345-360
    nearest_interpolation   PointSegment/RandLANet.py:362-375
    gather_neighbour        PointSegment/RandLANet.py:377-386
    att_pooling             PointSegment/RandLANet.py:388-401
    conv2d                  PointSegment/helper_tf_util.py:115-170   (1x1 conv = row-major GEMM, +bias, BN, LeakyReLU 0.2)
    conv2d_transpose        PointSegment/helper_tf_util.py:173-250   (kernel [1,1,out,in]  =>  Y = X . W[0,0]^T)
    dropout                 PointSegment/helper_tf_util.py:553-574   (identity when not training)
    tf_map (index pyramid)  PointSegment/runBraTS.py:140-161
    get_loss                PointSegment/RandLANet.py:267-274

PARITY UNPINNED: the arithmetic of the reference lives in tensorflow-gpu 1.11.0 (environment.yml:158-160),
which is not installable in the build container, and the reference ships no golden vectors for the network.
This file follows the call sites above literally; torch-CPU is used by tests/test_oracle_network.py as an
independent cross-check of the same formulas.  `dtype` selects float64 (the yardstick) or float32.

Tensors keep the reference's logical shapes without the dummy axis: features [B,N,C], neighbour tensors
[B,N,K,C].  Parameters come in a flat dict keyed by the TF variable names (scope concatenation without
separators, RandLANet.py:121,315-334,395,400): e.g. "Encoder_layer_0LFAatt_pooling_1fc/kernel".
"""
import numpy as np

BN_EPS = 1e-6  # tf.layers.batch_normalization(x, -1, 0.99, 1e-6) RandLANet.py:115, helper_tf_util.py:167,246
LEAKY = 0.2  # tf.nn.leaky_relu default alpha (RandLANet.py:115,321) and explicit alpha=0.2 (helper_tf_util.py:169,249)


def leaky_relu(x):
    return np.where(x >= 0, x, x * np.asarray(LEAKY, x.dtype))


def batch_norm_eval(x, p, name, dt):
    """Inference-mode tf.layers.batch_normalization: (x-mean)/sqrt(var+eps)*gamma+beta over the last axis."""
    g = p[name + "/gamma"].astype(dt)
    b = p[name + "/beta"].astype(dt)
    m = p[name + "/moving_mean"].astype(dt)
    v = p[name + "/moving_variance"].astype(dt)
    return (x - m) / np.sqrt(v + np.asarray(BN_EPS, dt)) * g + b


def conv2d(x, p, scope, dt, bn=True, act=True):
    """helper_tf_util.conv2d with a 1x1 kernel: weights [Cin,Cout] (TF [1,1,Cin,Cout] squeezed)."""
    y = x @ p[scope + "/weights"].astype(dt) + p[scope + "/biases"].astype(dt)
    if bn:
        y = batch_norm_eval(y, p, scope + "/batch_normalization", dt)
    if act:
        y = leaky_relu(y)
    return y


def conv2d_transpose(x, p, scope, dt):
    """helper_tf_util.conv2d_transpose, 1x1 stride 1: weights [Cout,Cin] (TF [1,1,Cout,Cin] squeezed)."""
    y = x @ p[scope + "/weights"].astype(dt).T + p[scope + "/biases"].astype(dt)
    y = batch_norm_eval(y, p, scope + "/batch_normalization", dt)
    return leaky_relu(y)


def gather_neighbour(pc, idx):
    """pc [B,N,d], idx [B,N',K] -> [B,N',K,d] (tf.batch_gather)."""
    B = pc.shape[0]
    return np.stack([pc[b][idx[b]] for b in range(B)], 0)


def relative_pos_encoding(xyz, idx):
    nbr = gather_neighbour(xyz, idx)
    tile = np.broadcast_to(xyz[:, :, None, :], nbr.shape)
    rel = tile - nbr
    dis = np.sqrt(np.sum(np.square(rel), axis=-1, keepdims=True))
    return np.concatenate([dis, rel, tile, nbr], axis=-1)


def att_pooling(fset, p, name, dt):
    """fset [B,N,K,d]: dense d->d (no bias), softmax over K, weighted sum, then conv2d d->d_out."""
    act = fset @ p[name + "fc/kernel"].astype(dt)
    act = act - act.max(axis=2, keepdims=True)
    e = np.exp(act)
    scores = e / e.sum(axis=2, keepdims=True)
    agg = np.sum(fset * scores, axis=2)
    return conv2d(agg, p, name + "mlp", dt)


def building_block(xyz, feature, idx, p, name, dt):
    f_xyz = relative_pos_encoding(xyz, idx)
    f_xyz = conv2d(f_xyz, p, name + "mlp1", dt)
    f_nb = gather_neighbour(feature, idx)
    f_agg = att_pooling(np.concatenate([f_nb, f_xyz], -1), p, name + "att_pooling_1", dt)
    f_xyz = conv2d(f_xyz, p, name + "mlp2", dt)
    f_nb = gather_neighbour(f_agg, idx)
    return att_pooling(np.concatenate([f_nb, f_xyz], -1), p, name + "att_pooling_2", dt)


def dilated_res_block(feature, xyz, idx, p, name, dt):
    f_pc = conv2d(feature, p, name + "mlp1", dt)
    f_pc = building_block(xyz, f_pc, idx, p, name + "LFA", dt)
    f_pc = conv2d(f_pc, p, name + "mlp2", dt, act=False)
    shortcut = conv2d(feature, p, name + "shortcut", dt, act=False)
    return leaky_relu(f_pc + shortcut)


def random_sample(feature, pool_idx):
    return gather_neighbour(feature, pool_idx).max(axis=2)


def nearest_interpolation(feature, interp_idx):
    return gather_neighbour(feature, interp_idx)[:, :, 0, :]


def inference(p, num_layers, xyz, neigh_idx, sub_idx, interp_idx, features, dtype=np.float64, tap=None):
    """Eval-mode forward. xyz/neigh_idx/sub_idx/interp_idx: lists of length num_layers; features [B,N0,Cin].
    Returns logits [B,N0,num_classes].  `tap` (dict) receives named intermediates if given."""
    dt = np.dtype(dtype)
    xyz = [x.astype(dt) for x in xyz]
    f = features.astype(dt) @ p["fc0/kernel"].astype(dt) + p["fc0/bias"].astype(dt)
    f = leaky_relu(batch_norm_eval(f, p, "batch_normalization", dt))
    if tap is not None:
        tap["fc0"] = f
    enc = []
    for i in range(num_layers):
        f_enc = dilated_res_block(f, xyz[i], neigh_idx[i], p, "Encoder_layer_%d" % i, dt)
        f = random_sample(f_enc, sub_idx[i])
        if i == 0:
            enc.append(f_enc)
        enc.append(f)
        if tap is not None:
            tap["enc%d" % i] = f_enc
            tap["pool%d" % i] = f
    f = conv2d(enc[-1], p, "decoder_0", dt)
    if tap is not None:
        tap["decoder_0"] = f
    for j in range(num_layers):
        f_interp = nearest_interpolation(f, interp_idx[-j - 1])
        f = conv2d_transpose(np.concatenate([enc[-j - 2], f_interp], -1), p, "Decoder_layer_%d" % j, dt)
        if tap is not None:
            tap["dec%d" % j] = f
    f = conv2d(f, p, "fc1", dt)
    f = conv2d(f, p, "fc2", dt)
    # dropout(keep_prob=0.5) is the identity when is_training is False (helper_tf_util.py:571-573)
    return conv2d(f, p, "fc", dt, bn=False, act=False)


def build_pyramid(knn_search, xyz, k_n, ratios):
    """tf_map of runBraTS.py:147-156 with any `knn_search(support, query, k) -> int [B,N2,k]`."""
    pts, nbr, pool, up = [], [], [], []
    cur = xyz
    for r in ratios:
        n = cur.shape[1]
        idx = knn_search(cur, cur, k_n).astype(np.int32)
        sub = cur[:, :n // r, :]
        pts.append(cur)
        nbr.append(idx)
        pool.append(idx[:, :n // r, :])
        up.append(knn_search(sub, cur, 1).astype(np.int32))
        cur = sub
    return pts, nbr, pool, up


def weighted_ce_loss(logits, labels, class_weights):
    """get_loss, RandLANet.py:267-274 (float64)."""
    z = logits.reshape(-1, logits.shape[-1]).astype(np.float64)
    y = labels.reshape(-1)
    z = z - z.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    w = np.asarray(class_weights, np.float64)[y]
    return float(np.mean(-logp[np.arange(len(y)), y] * w))
