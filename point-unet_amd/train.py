"""Training step of the PointSegment RandLA-Net on MI355X: forward in training mode (batch-statistics BatchNorm,
dropout), class-weighted cross-entropy, backward, Adam -- the device counterpart of

    Network.__init__ loss / optimizer      PointSegment/RandLANet.py:62-90, 267-274
    Network.train's sess.run([train_op, extra_update_ops, ...])   PointSegment/RandLANet.py:162-169
    tf.layers.batch_normalization(..., training=True)             helper_tf_util.py:167,246; RandLANet.py:115

The reference relies on TF autodiff.  Here the whole step is ONE call through the C ABI, `ps_randla_train_step`
(csrc/trainer.hip: the tape, the activation pool and the BatchNorm moving-statistics updates live in C++; include/pointseg.h).
`Trainer` below is the host-side holder: it owns the flat parameter / gradient / Adam / statistics buffers as torch tensors
(named views in the reference's variable names), binds them to the native trainer, and supplies the collective -- torch.distributed's
all-reduce (RCCL on the GPU box) wrapped as the C callback the library asks for.

`Trainer(engine="python")` keeps the earlier host-side tape (class Tape below: the same op-level kernels recorded and replayed
from Python) as the A/B reference of the native engine and as the home of experimental pieces that are not part of the native
step (fused_convbn).

Multi-GPU: one cloud per GPU, gradients averaged with ONE all-reduce of the flat fp32 buffer
(4 992 852 floats for BraTS).  BatchNorm statistics are per-GPU by default (the reference itself never runs batch > 1);
Trainer(sync_bn=True) shares them between the ranks (two 2*C-float all-reduces per BatchNorm layer and step), which makes
8 GPUs x 1 cloud the same optimisation step as 1 GPU x 8 clouds.
"""
import ctypes

import numpy as np
import torch

from . import _lib, runtime, weights

BN_EPS = 1e-6
BN_MOMENTUM = 0.99


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _rowmajor(t):
    """t as a 2-D tensor with contiguous rows (any row stride): column blocks of a wider tensor pass through untouched,
    the strided entry points (ps_op_*_ex) take the row stride."""
    return t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()


def allreduce_mean_(flat, dist):
    """Gradient synchronisation of config 4: ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) of the
    flat fp32 gradient buffer, then the mean over ranks.  19.97 MB for the BraTS model: latency-bound, so no bucketing."""
    dist.all_reduce(flat)
    flat.div_(dist.get_world_size())
    return flat


class Tape:
    """Records (output, backward closure) pairs; gradients are keyed by tensor identity and accumulated on the device."""

    def __init__(self, ctx, sync=None, bf16=False):
        """sync: an initialised torch.distributed module for BatchNorm statistics shared by all ranks (None: per-GPU).
        bf16: the step runs in the bf16-MLP mode (ps_set_train_gemm_bf16 is on around it)."""
        self.bf16 = bool(bf16)
        self.ctx = ctx
        self.L = _lib.lib()
        self.h = ctx.handle
        self.ops = []
        self.grads = {}
        self.sync = sync

    def accum(self, t, g):
        k = id(t)
        if k in self.grads:
            have = self.grads[k]
            if have.is_contiguous() and g.is_contiguous():
                _lib.check(self.L.ps_op_axpy(self.h, 1.0, _p(g), g.numel(), _p(have)))
            else:  # one of them is a column block of a concat buffer
                have.add_(g)
        else:
            self.grads[k] = g

    def accum_buffer(self, t):
        """The gradient buffer of t for ops that ADD into their output (scatter-add, max-pool backward): the one already
        recorded, or a fresh zero tensor that becomes it -- no temporary, no axpy."""
        k = id(t)
        if k not in self.grads:
            self.grads[k] = torch.zeros_like(t)
        return self.grads[k]

    def backward(self, out, dout):
        self.grads[id(out)] = dout
        for t, bw in reversed(self.ops):
            g = self.grads.pop(id(t), None)
            if g is not None:
                bw(g)
        self.ops = []
        self.grads = {}

    # ---- ops -------------------------------------------------------------------------------------------------------
    def linear(self, x, W, b, gW, gb, transposed=False, into=None, fp32_only=False):
        """y = x . W (+ b).  W is [cin,cout], or [cout,cin] when transposed (conv2d_transpose kernels).
        into: an existing [R, cout] tensor of the tape that the product is ADDED to (the GEMM's accumulate epilogue); the result is that
        same tensor, and its gradient is handed on unchanged to the op that produced it."""
        if fp32_only and self.bf16:
            # not one of the "bf16 MLPs" (the LocSE convolution 10 -> h): forward now, backward later, both with fp32 operands
            _lib.check(self.L.ps_set_train_gemm_bf16(self.h, 0))
            try:
                y = self.linear(x, W, b, gW, gb, transposed, into)
            finally:
                _lib.check(self.L.ps_set_train_gemm_bf16(self.h, 1))
            t, inner = self.ops[-1]

            def bw32(dy):
                _lib.check(self.L.ps_set_train_gemm_bf16(self.h, 0))
                try:
                    inner(dy)
                finally:
                    _lib.check(self.L.ps_set_train_gemm_bf16(self.h, 1))

            self.ops[-1] = (t, bw32)
            return y
        Wm = W.t().contiguous() if transposed else W
        x_in = x  # the tensor the tape knows (gradients are keyed by identity)
        x = _rowmajor(x)
        R, cin = x.shape
        ldx = x.stride(0)
        cout = Wm.shape[1]
        if into is None:
            y = torch.empty((R, cout), dtype=torch.float32, device=x.device)
            _lib.check(self.L.ps_op_conv1x1_ex(self.h, _p(x), ldx, _p(Wm), _p(b), R, cin, cout, 0, 0, _p(y), cout))
        else:
            y = into
            _lib.check(self.L.ps_op_conv1x1_ex(self.h, _p(x), ldx, _p(Wm), _p(b), R, cin, cout, 0, 1, _p(y), y.stride(0)))

        def bw(dy):
            if into is not None:
                self.grads[id(into)] = dy  # d(into + x.W)/d(into) = 1: the producer of `into` (earlier on the tape) gets the same gradient
            dy = _rowmajor(dy)
            lddy = dy.stride(0)
            pgb = _p(gb) if gb is not None else None
            if transposed or not gW.is_contiguous():
                dW = torch.empty((cin, cout), dtype=torch.float32, device=x.device)
                _lib.check(self.L.ps_op_linear_wgrad_ex(self.h, _p(x), ldx, _p(dy), lddy, R, cin, cout, _p(dW), pgb))
                gW.copy_(dW.t() if transposed else dW)
            else:  # straight into the parameter's slice of the flat gradient buffer
                _lib.check(self.L.ps_op_linear_wgrad_ex(self.h, _p(x), ldx, _p(dy), lddy, R, cin, cout, _p(gW), pgb))
            if x_in.requires_grad_flag:
                Wt = Wm.t().contiguous()
                have = self.grads.get(id(x_in))
                if have is not None and have.dim() == 2 and have.stride(1) == 1:
                    # x already has a gradient from another consumer: add this one in the GEMM epilogue
                    _lib.check(self.L.ps_op_conv1x1_ex(self.h, _p(dy), lddy, _p(Wt), None, R, cout, cin, 0, 1, _p(have), have.stride(0)))
                else:
                    dx = torch.empty((R, cin), dtype=torch.float32, device=x.device)
                    _lib.check(self.L.ps_op_conv1x1_ex(self.h, _p(dy), lddy, _p(Wt), None, R, cout, cin, 0, 0, _p(dx), cin))
                    self.accum(x_in, dx)

        y.requires_grad_flag = True
        self.ops.append((y, bw))
        return y

    def bn_act(self, x, gamma, beta, ggamma, gbeta, mov_mean, mov_var, leaky, out=None):
        """out: optional [R, C] column block of a wider tensor (rows contiguous) that receives y."""
        R, C = x.shape
        y = torch.empty_like(x) if out is None else out
        ldy = y.stride(0)
        stats = torch.empty((5, C), dtype=torch.float32, device=x.device)  # mean, invstd, var, [sum x | sum x^2]
        sync = self.sync
        if sync is None:
            _lib.check(self.L.ps_op_bn_train_fwd_ex(self.h, _p(x), _p(gamma), _p(beta), R, C, BN_EPS, 1 if leaky else 0, _p(y), ldy, _p(stats[0]),
                                                    _p(stats[1]), _p(stats[2]), _p(stats[3])))
            R_total = R
        else:
            # statistics over the rows of ALL ranks: two small all-reduces per layer (2*C floats forward, 2*C backward)
            R_total = R * sync.get_world_size()
            _lib.check(self.L.ps_op_bn_train_sums(self.h, _p(x), R, C, _p(stats[3])))
            sync.all_reduce(stats[3:5])
            _lib.check(self.L.ps_op_bn_train_apply_ex(self.h, _p(x), _p(gamma), _p(beta), _p(stats[3]), R, R_total, C, BN_EPS, 1 if leaky else 0,
                                                      _p(y), ldy, _p(stats[0]), _p(stats[1]), _p(stats[2])))
        # moving statistics (the reference's extra_update_ops, RandLANet.py:90,163)
        mov_mean.mul_(BN_MOMENTUM).add_(stats[0], alpha=1 - BN_MOMENTUM)
        mov_var.mul_(BN_MOMENTUM).add_(stats[2], alpha=1 - BN_MOMENTUM)

        def bw(dy):
            dx = torch.empty_like(x)
            dyc = _rowmajor(dy)
            lddy = dyc.stride(0)
            if sync is None:
                _lib.check(self.L.ps_op_bn_train_bwd_ex(self.h, _p(dyc), lddy, _p(x), _p(gamma), _p(beta), _p(stats[0]), _p(stats[1]), R, C,
                                                        1 if leaky else 0, _p(dx), _p(ggamma), _p(gbeta)))
            else:
                # local sums are this rank's dgamma / dbeta (averaged with every other gradient later); dx needs the global ones
                _lib.check(self.L.ps_op_bn_train_bwd_sums_ex(self.h, _p(dyc), lddy, _p(x), _p(gamma), _p(beta), _p(stats[0]), _p(stats[1]), R, C,
                                                             1 if leaky else 0, _p(ggamma), _p(gbeta)))
                tot = torch.stack([gbeta.reshape(-1), ggamma.reshape(-1)])
                sync.all_reduce(tot)
                _lib.check(self.L.ps_op_bn_train_bwd_apply_ex(self.h, _p(dyc), lddy, _p(x), _p(gamma), _p(beta), _p(stats[0]), _p(stats[1]),
                                                              _p(tot[0]), _p(tot[1]), R, R_total, C, 1 if leaky else 0, _p(dx)))
            self.accum(x, dx)

        y.requires_grad_flag = True
        self.ops.append((y, bw))
        return y

    def locse_bn_act(self, xyz, idx, B, W, b, gW, gb, gamma, beta, ggamma, gbeta, mov_mean, mov_var, out=None):
        """f_xyz = LeakyReLU(BN_train(relative_pos_encoding(xyz, idx) . W + b)) -> [B*N*K, h] (out: optional column block), with
        nothing but that output in memory: statistics, output and every gradient are recomputed from xyz [B*N,3] and idx [B,N,K]
        (ps_op_locse_train_*).  The [10,h]-sized finishing arithmetic of the weight gradient is done here."""
        N, K = idx.shape[1], idx.shape[2]
        h = W.shape[1]
        R = B * N * K
        dev = xyz.device
        sync = self.sync
        R_total = R if sync is None else R * sync.get_world_size()
        sums = torch.empty(2 * h, dtype=torch.float64, device=dev)  # float64: the variance is a difference of nearly equal sums
        _lib.check(self.L.ps_op_locse_train_sums(self.h, _p(xyz), _p(idx), B, N, K, _p(W), _p(b), h, _p(sums)))
        if sync is not None:
            sync.all_reduce(sums)
        mean64 = sums[:h] / R_total
        var = (sums[h:] / R_total - mean64 * mean64).clamp_min_(0.0).float()  # population variance (tf.nn.moments)
        mean = mean64.float()
        invstd = torch.rsqrt(var + BN_EPS)
        scale = (gamma.reshape(-1) * invstd).contiguous()
        shift = beta.reshape(-1).contiguous()
        y = torch.empty((R, h), dtype=torch.float32, device=dev) if out is None else out
        _lib.check(self.L.ps_op_locse_train_apply(self.h, _p(xyz), _p(idx), B, N, K, _p(W), _p(b), h, _p(mean), _p(scale), _p(shift), _p(y), y.stride(0)))
        mov_mean.mul_(BN_MOMENTUM).add_(mean, alpha=1 - BN_MOMENTUM)
        mov_var.mul_(BN_MOMENTUM).add_(var, alpha=1 - BN_MOMENTUM)
        mean, invstd = mean.contiguous(), invstd.contiguous()

        def bw(dz):
            dzc = _rowmajor(dz)
            acc = torch.empty(23 * h + 16, dtype=torch.float32, device=dev)
            _lib.check(self.L.ps_op_locse_train_bwd(self.h, _p(xyz), _p(idx), B, N, K, _p(W), _p(b), h, _p(scale), _p(shift), _p(mean), _p(invstd),
                                                    _p(dzc), dzc.stride(0), _p(acc)))
            S1, S2, XS = acc[:h], acc[h:2 * h], acc[2 * h:3 * h]
            A, G, E = acc[3 * h:13 * h].view(10, h), acc[13 * h:23 * h].view(10, h), acc[23 * h:23 * h + 10]
            ggamma.copy_(S2.view_as(ggamma))  # this rank's dgamma / dbeta (averaged with every other gradient later)
            gbeta.copy_(S1.view_as(gbeta))
            tot = torch.stack([S1, S2])
            if sync is not None:
                sync.all_reduce(tot)
            m1, m2 = tot[0] / R_total, tot[1] / R_total
            k = gamma.reshape(-1) * invstd
            # dy = k (g - m1 - xh m2)  =>  enc10^T dy and sum dy from the sums of the one pass
            gW.copy_((k[None, :] * (A - E[:, None] * m1[None, :] - G * m2[None, :])).view_as(gW))
            gb.copy_((k * (S1 - R * m1 - XS * m2)).view_as(gb))

        y.requires_grad_flag = True
        self.ops.append((y, bw))
        return y

    def conv_bn_act(self, x, W, b, gW, gb, gamma, beta, ggamma, gbeta, mov_mean, mov_var, out=None):
        """z = LeakyReLU(BN_train(x . W + b)) for W [C, C] on [R, C] rows without storing x . W or its gradient (ps_op_conv_bn_train_*):
        the product is recomputed from x in the statistics pass, the apply pass and the two backward passes.  The [C, C]-sized
        finishing arithmetic of the weight gradient is done here."""
        x_in = x
        x = _rowmajor(x)
        R, C = x.shape
        CP = max(C, 16)
        dev = x.device
        sync = self.sync
        R_total = R if sync is None else R * sync.get_world_size()
        sums = torch.empty(3 * CP, dtype=torch.float64, device=dev)
        _lib.check(self.L.ps_op_conv_bn_train_sums(self.h, _p(x), x.stride(0), _p(W), _p(b), R, C, _p(sums)))
        if sync is not None:
            sync.all_reduce(sums)
        mean64 = sums[:C] / R_total
        var = (sums[CP:CP + C] / R_total - mean64 * mean64).clamp_min_(0.0).float()
        mean = mean64.float().contiguous()
        sumx = sums[2 * CP:2 * CP + C].float()  # (of all ranks under SyncBN; only its local part is needed: see bw)
        invstd = torch.rsqrt(var + BN_EPS).contiguous()
        scale = (gamma.reshape(-1) * invstd).contiguous()
        shift = beta.reshape(-1).contiguous()
        y = torch.empty((R, C), dtype=torch.float32, device=dev) if out is None else out
        _lib.check(self.L.ps_op_conv_bn_train_apply(self.h, _p(x), x.stride(0), _p(W), _p(b), R, C, _p(mean), _p(scale), _p(shift), _p(y), y.stride(0)))
        mov_mean.mul_(BN_MOMENTUM).add_(mean, alpha=1 - BN_MOMENTUM)
        mov_var.mul_(BN_MOMENTUM).add_(var, alpha=1 - BN_MOMENTUM)
        if sync is not None:  # this rank's own sum of x for its own weight gradient
            local = torch.empty(3 * CP, dtype=torch.float64, device=dev)
            _lib.check(self.L.ps_op_conv_bn_train_sums(self.h, _p(x), x.stride(0), _p(W), _p(b), R, C, _p(local)))
            sumx = local[2 * CP:2 * CP + C].float()

        def bw(dz):
            dzc = _rowmajor(dz)
            acc = torch.empty(3 * CP + 2 * CP * CP, dtype=torch.float32, device=dev)
            _lib.check(self.L.ps_op_conv_bn_train_bwd_sums(self.h, _p(x), x.stride(0), _p(W), _p(b), R, C, _p(mean), _p(invstd), _p(scale), _p(shift),
                                                           _p(dzc), dzc.stride(0), _p(acc)))
            S1, S2, XS = acc[:C], acc[CP:CP + C], acc[2 * CP:2 * CP + C]
            A = acc[3 * CP:3 * CP + CP * CP].view(CP, CP)[:C, :C]
            G = acc[3 * CP + CP * CP:].view(CP, CP)[:C, :C]
            ggamma.copy_(S2.view_as(ggamma))
            gbeta.copy_(S1.view_as(gbeta))
            tot = torch.stack([S1, S2])
            if sync is not None:
                sync.all_reduce(tot)
            m1, m2 = (tot[0] / R_total).contiguous(), (tot[1] / R_total).contiguous()
            k = gamma.reshape(-1) * invstd
            gW.copy_((k[None, :] * (A - sumx[:, None] * m1[None, :] - G * m2[None, :])).view_as(gW))
            gb.copy_((k * (S1 - R * m1 - XS * m2)).view_as(gb))
            if getattr(x_in, "requires_grad_flag", False):
                have = self.grads.get(id(x_in))
                if have is not None and have.dim() == 2 and have.stride(1) == 1:
                    _lib.check(self.L.ps_op_conv_bn_train_bwd_apply(self.h, _p(x), x.stride(0), _p(W), _p(b), R, C, _p(mean), _p(invstd), _p(scale),
                                                                    _p(shift), _p(m1), _p(m2), _p(dzc), dzc.stride(0), 1, _p(have), have.stride(0)))
                else:
                    dx = torch.empty((R, C), dtype=torch.float32, device=dev)
                    _lib.check(self.L.ps_op_conv_bn_train_bwd_apply(self.h, _p(x), x.stride(0), _p(W), _p(b), R, C, _p(mean), _p(invstd), _p(scale),
                                                                    _p(shift), _p(m1), _p(m2), _p(dzc), dzc.stride(0), 0, _p(dx), C))
                    self.accum(x_in, dx)

        y.requires_grad_flag = True
        self.ops.append((y, bw))
        return y

    def gather(self, x, idx, B, out=None):
        """x [B*N, d], idx [B, M, K] -> [B*M*K, d]; out: optional column block of a wider tensor that receives the rows"""
        N, d = x.shape[0] // B, x.shape[1]
        M, K = idx.shape[1], idx.shape[2]
        if out is None:
            out = torch.empty((B * M * K, d), dtype=torch.float32, device=x.device)
        _lib.check(self.L.ps_op_gather_neighbour_ex(self.h, _p(x), _p(idx), B, N, M, K, d, _p(out), out.stride(0)))

        def bw(dy):
            dy = _rowmajor(dy)
            _lib.check(self.L.ps_op_scatter_add_rows_ex(self.h, _p(dy), dy.stride(0), _p(idx), B, N, M * K, d, _p(self.accum_buffer(x))))

        out.requires_grad_flag = True
        self.ops.append((out, bw))
        return out

    def cat(self, a, b):
        out = torch.cat([a, b], dim=1)
        ca = a.shape[1]

        def bw(dy):
            if getattr(a, "requires_grad_flag", False):
                self.accum(a, dy[:, :ca].contiguous())
            if getattr(b, "requires_grad_flag", False):
                self.accum(b, dy[:, ca:].contiguous())

        out.requires_grad_flag = True
        self.ops.append((out, bw))
        return out

    def concat_views(self, buf, a, b):
        """buf [R, ca+cb] whose left / right column blocks a and b were written in place by their producers (gather /
        bn_act with out=): the concat costs nothing, and its backward hands the column blocks of the gradient on as views."""
        ca = a.shape[1]

        def bw(dy):
            dy = _rowmajor(dy)
            if getattr(a, "requires_grad_flag", False):
                self.accum(a, dy[:, :ca])
            if getattr(b, "requires_grad_flag", False):
                self.accum(b, dy[:, ca:])

        buf.requires_grad_flag = True
        self.ops.append((buf, bw))
        return buf

    def softpool(self, fset, scores, K):
        RK, d = fset.shape
        R = RK // K
        probs = torch.empty_like(fset)
        agg = torch.empty((R, d), dtype=torch.float32, device=fset.device)
        _lib.check(self.L.ps_op_softmax_pool_fwd(self.h, _p(fset), _p(scores), R, K, d, _p(probs), _p(agg)))

        def bw(dy):
            dfset = torch.empty_like(fset)
            dscores = torch.empty_like(fset)
            _lib.check(self.L.ps_op_softmax_pool_bwd(self.h, _p(dy.contiguous()), _p(fset), _p(probs), R, K, d, _p(dfset), _p(dscores)))
            self.accum(fset, dfset)
            self.accum(scores, dscores)

        agg.requires_grad_flag = True
        self.ops.append((agg, bw))
        return agg

    def attpool(self, fset, W, gW, K):
        """att_pooling's score product + softmax over K + weighted sum as ONE kernel per direction (csrc/attpool_train.hip): neither
        the scores nor the probabilities reach HBM, the backward recomputes them from fset.  fset [R*K, d] may be a concat buffer
        whose column blocks were written in place; W [d, d]."""
        RK, d = fset.shape
        R = RK // K
        fs = _rowmajor(fset)
        agg = torch.empty((R, d), dtype=torch.float32, device=fset.device)
        _lib.check(self.L.ps_op_att_pool_train_fwd(self.h, _p(fs), fs.stride(0), _p(W), R, K, d, _p(agg)))

        def bw(dy):
            dfset = torch.empty((RK, d), dtype=torch.float32, device=fset.device)
            dyc = dy.contiguous()
            if gW.is_contiguous():
                _lib.check(self.L.ps_op_att_pool_train_bwd(self.h, _p(fs), fs.stride(0), _p(W), _p(dyc), R, K, d, _p(dfset), d, _p(gW)))
            else:
                tmp = torch.empty((d, d), dtype=torch.float32, device=fset.device)
                _lib.check(self.L.ps_op_att_pool_train_bwd(self.h, _p(fs), fs.stride(0), _p(W), _p(dyc), R, K, d, _p(dfset), d, _p(tmp)))
                gW.copy_(tmp)
            self.accum(fset, dfset)

        agg.requires_grad_flag = True
        self.ops.append((agg, bw))
        return agg

    def attpool_split(self, f_src, idx, f_xyz, W, gW, B):
        """attpool over fset = [gather(f_src, idx) | f_xyz] without the gather, the concat buffer or the scatter-add of its gradient
        (ps_op_att_pool_train_*_split).  f_src [B*N, h], idx [B, M, K], f_xyz [B*M*K, h] (rows may be strided), W [2h, 2h]."""
        N, h = f_src.shape[0] // B, f_src.shape[1]
        M, K = idx.shape[1], idx.shape[2]
        d = 2 * h
        fs, fx = _rowmajor(f_src), _rowmajor(f_xyz)
        agg = torch.empty((B * M, d), dtype=torch.float32, device=f_src.device)
        _lib.check(self.L.ps_op_att_pool_train_fwd_split(self.h, _p(fs), fs.stride(0), _p(idx), B, N, M, _p(fx), fx.stride(0), _p(W), K, d, _p(agg)))

        def bw(dy):
            dfx = torch.empty((B * M * K, h), dtype=torch.float32, device=f_src.device)
            dsrc = self.accum_buffer(f_src)  # the gathered half's gradient is added in place (float atomics)
            tmp = gW if gW.is_contiguous() else torch.empty((d, d), dtype=torch.float32, device=f_src.device)
            _lib.check(self.L.ps_op_att_pool_train_bwd_split(self.h, _p(fs), fs.stride(0), _p(idx), B, N, M, _p(fx), fx.stride(0), _p(W),
                                                             _p(dy.contiguous()), K, d, _p(dsrc), dsrc.stride(0), _p(dfx), h, _p(tmp)))
            if tmp is not gW:
                gW.copy_(tmp)
            self.accum(f_xyz, dfx)

        agg.requires_grad_flag = True
        self.ops.append((agg, bw))
        return agg

    def maxpool(self, x, pool_idx, B):
        N, d = x.shape[0] // B, x.shape[1]
        M, K = pool_idx.shape[1], pool_idx.shape[2]
        out = torch.empty((B * M, d), dtype=torch.float32, device=x.device)
        _lib.check(self.L.ps_op_random_sample(self.h, _p(x), _p(pool_idx), B, N, M, K, d, _p(out)))

        def bw(dy):
            _lib.check(self.L.ps_op_random_sample_bwd(self.h, _p(dy.contiguous()), _p(out), _p(x), _p(pool_idx), B, N, M, K, d,
                                                      _p(self.accum_buffer(x))))

        out.requires_grad_flag = True
        self.ops.append((out, bw))
        return out

    def add_lrelu(self, a, b):
        y = torch.empty_like(a)
        _lib.check(self.L.ps_op_add_lrelu(self.h, _p(a), _p(b), a.numel(), _p(y)))

        def bw(dy):
            ds = torch.empty_like(a)
            _lib.check(self.L.ps_op_add_lrelu_bwd(self.h, _p(dy.contiguous()), _p(y), a.numel(), _p(ds)))
            self.accum(a, ds)
            self.accum(b, ds.clone())

        y.requires_grad_flag = True
        self.ops.append((y, bw))
        return y

    def dropout(self, x, keep_prob, seed):
        if keep_prob >= 1.0:
            return x
        y = torch.empty_like(x)
        mask = torch.empty_like(x)
        _lib.check(self.L.ps_op_dropout(self.h, _p(x), x.numel(), seed & 0xffffffff, keep_prob, _p(y), _p(mask)))

        def bw(dy):
            dx = torch.empty_like(x)
            _lib.check(self.L.ps_op_mul(self.h, _p(dy.contiguous()), _p(mask), x.numel(), _p(dx)))
            self.accum(x, dx)

        y.requires_grad_flag = True
        self.ops.append((y, bw))
        return y


class Trainer:
    """Parameters (flat fp32 buffer + named views), Adam state and the train step."""

    def __init__(self, config, params=None, device=0, seed=0, learning_rate=None, class_weights=None, keep_prob=0.5, ctx=None, sync_bn=False,
                 mlp_dtype="fp32", ignored_label_inds=None, fused_att=True, fused_locse=True, fused_convbn=None, engine="native", deterministic=True,
                 overlap_wgrad=False, act_bf16=None):
        """sync_bn: with a `dist` passed to train_step, BatchNorm uses the statistics of all ranks' rows, which makes "W GPUs x
        one cloud" numerically the same step as "one GPU x W clouds" (SURVEY 8e); off = per-GPU statistics.
        mlp_dtype: "fp32" (default) or "bf16" -- BASELINE configs[2]'s "bf16 MLPs": the shared-MLP GEMMs (forward, input gradient,
        weight gradient) round their operands to bf16 and accumulate in fp32 (ps_set_train_gemm_bf16); everything else stays fp32.
        engine: "native" (default) = ps_randla_train_step, the tape in C++ (csrc/trainer.hip); "python" = the host-side tape of this
        file (A/B reference).
        deterministic (native engine): the scatter-adds of the backward pass run as fixed-order gather-reductions over inverse indices
        (csrc/invidx.hip) instead of float atomics -- two runs of a step produce bit-identical gradients."""
        if mlp_dtype not in ("fp32", "bf16"):
            raise ValueError("mlp_dtype must be 'fp32' or 'bf16'")
        if engine not in ("native", "python"):
            raise ValueError("engine must be 'native' or 'python'")
        self.engine = engine
        if fused_convbn is None:  # on in the native step (measured 47.9 -> 46.3 ms at batch 8), off in the Python tape (its older passes lose)
            fused_convbn = engine == "native"
        self.deterministic = bool(deterministic)
        self.mlp_bf16 = mlp_dtype == "bf16"
        self.fused_att = bool(fused_att)  # False: the op-by-op attentive pooling everywhere (A/B switch of bench.py --no-fused-att)
        # LocSE branch (relative_pos_encoding -> conv 10->h -> BatchNorm -> LeakyReLU) recomputed from coordinates and indices instead of
        # materialised (csrc/locse_train.hip).  Also in the bf16-MLP mode: the 10 -> h convolution of the position encoding is not one of
        # the "bf16 MLPs" (K = 10 is no matrix-pipe shape): it keeps fp32 operands, fused or not
        self.fused_locse = bool(fused_locse)
        # LFA mlp2 (conv c->c + BatchNorm + LeakyReLU on [N*K] rows) with the pre-BatchNorm product recomputed instead of stored
        # (csrc/smallconv_train.hip); same restriction.  OFF by default: 8 passes instead of 14, but its 16-row MFMA tile kernels are
        # issue bound and measured SLOWER than the streaming kernels they replace (batch 8: +0.7 / +0.2 / +1.0 ms with c = 8 / 32 / 64
        # alone, 58.8 vs 56.9 ms with all three) -- correct (tests/test_gpu_train.py) and kept as the starting point for wider tiles
        self.fused_convbn = bool(fused_convbn) and not self.mlp_bf16
        # native engine (ps_train_options.overlap_wgrad): weight-gradient products on a second HIP stream of the trainer.  OFF by default:
        # measured on MI355X (round 4, same box): batch 1 9.23 -> 9.74 ms (every fork is an event wait across hardware queues, and dependent
        # kernels spread over more queues are scheduled later), batch 8 42.56 -> 42.46 ms (the step is already HBM-bound end to end)
        self.overlap_wgrad = bool(overlap_wgrad)
        # native engine, bf16-MLP mode (ps_train_options.act_bf16): the [N*K, h] rows of the LFA branch AND of their gradients stored as bfloat16
        # at levels 0-2 (measured, batch 8: 33.6 ms without, 32.4 with the activations, 31.9 with the gradient rows too).
        # None = on with mlp_dtype="bf16" (BASELINE configs[2]); the Python tape keeps fp32 storage.
        self.act_bf16 = (mlp_dtype == "bf16") if act_bf16 is None else bool(act_bf16)
        self.collective_at_world_one = False  # call the all-reduce callback even when the process group has ONE rank (measurement)
        # native engine (ps_train_options.fused_convbn): c = 8 on one-thread-per-row kernels (csrc/convbn_rows.hip; fp32 also in the bf16-MLP
        # mode: an 8 x 8 product has no matrix-pipe shape), wider layers on the tile kernels, which round their operands in the bf16 mode
        self._fused_convbn_native = bool(fused_convbn)
        self.sync_bn = bool(sync_bn)
        self.cfg = config
        self.device = torch.device("cuda", device)
        self.ctx = ctx or runtime.default_context(device)
        self.keep_prob = keep_prob
        params = params if params is not None else weights.init_params(config, seed=seed)
        self.lr = float(learning_rate if learning_rate is not None else getattr(config, "learning_rate", 1e-4))
        ign = ignored_label_inds if ignored_label_inds is not None else getattr(config, "ignored_label_inds", [])
        self.ignored_label_inds = sorted(int(v) for v in ign)
        # the native trainer defines the layout of the flat buffers (csrc/trainer.hip: build_layout == weights.layer_dims order)
        lib = _lib.lib()
        rc = _lib.PsRandlaConfig()
        rc.num_layers, rc.k_n, rc.num_classes, rc.in_channels = config.num_layers, config.k_n, config.num_classes, config.in_channels
        for i in range(config.num_layers):
            rc.d_out[i] = config.d_out[i]
        self._h = ctypes.c_void_p()
        _lib.check(lib.ps_trainer_create(self.ctx.handle, ctypes.byref(rc), ctypes.byref(self._options()), ctypes.byref(self._h)))
        self.ctx.register(self)  # (Context.close() destroys the trainers still alive on it: the pool is not torch's memory)
        n_par, n_buf = lib.ps_trainer_param_count(self._h), lib.ps_trainer_buffer_count(self._h)
        self.flat = torch.empty(n_par, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.flat_buffers = torch.empty(n_buf, dtype=torch.float32, device=self.device)
        self.P, self.G, self.buffers, self.names = {}, {}, {}, []
        name = ctypes.create_string_buffer(256)
        off, nr, nc, isb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        for row in range(lib.ps_trainer_layout_rows(self._h)):
            _lib.check(lib.ps_trainer_layout(self._h, row, name, 256, ctypes.byref(off), ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(isb)))
            n = name.value.decode()
            shape = tuple(params[n].shape)
            assert int(np.prod(shape)) == nr.value * nc.value, (n, shape, nr.value, nc.value)
            src = torch.from_numpy(np.ascontiguousarray(params[n]))
            if isb.value:
                self.buffers[n] = self.flat_buffers[off.value:off.value + src.numel()].view(shape)
                self.buffers[n].copy_(src)
            else:
                self.names.append(n)
                self.P[n] = self.flat[off.value:off.value + src.numel()].view(shape)
                self.G[n] = self.grad[off.value:off.value + src.numel()].view(shape)
                self.P[n].copy_(src)
        _lib.check(lib.ps_trainer_bind(self._h, _p(self.flat), _p(self.grad), _p(self.m), _p(self.v), _p(self.flat_buffers)))
        cw = class_weights if class_weights is not None else np.ones(config.num_classes, np.float32)
        self.class_weights = torch.from_numpy(np.asarray(cw, np.float32).reshape(-1)).to(self.device)
        self.label_map = None
        if self.ignored_label_inds:
            # RandLANet.py:77-81: reducing_list = range(C) with a 0 inserted at every ignored index; ignored entries become -1 here
            red = list(range(config.num_classes))
            for v in self.ignored_label_inds:
                red = red[:v] + [-1] + red[v:]
            self.label_map = torch.tensor(red, dtype=torch.int32, device=self.device)
        self.step = 0
        self._rank = 0
        self._coll = None  # (callback object, dist) kept alive while the native trainer may call it
        self._warned_export = False

    def _options(self):
        o = _lib.PsTrainOptions()
        o.learning_rate, o.keep_prob = self.lr, self.keep_prob
        o.mlp_bf16, o.fused_att, o.fused_locse = int(self.mlp_bf16), int(self.fused_att), int(self.fused_locse)
        o.deterministic = int(self.deterministic)
        o.fused_convbn = int(self._fused_convbn_native)
        o.overlap_wgrad = int(self.overlap_wgrad)
        o.act_bf16 = int(self.act_bf16 and self.mlp_bf16)
        o.num_ignored = len(self.ignored_label_inds)
        for i, v in enumerate(self.ignored_label_inds):
            o.ignored_label_inds[i] = v
        return o

    def rebind(self, flat=None, grad=None, m=None, v=None, flat_buffers=None):
        """Hands the native trainer other flat buffers (a reloaded checkpoint, a buffer swap): ps_trainer_bind.  Buffers not given
        are replaced by fresh copies of the current ones; the per-name views P / G / buffers are rebuilt over the new storage."""
        lib = _lib.lib()
        old = (self.flat, self.flat_buffers)
        self.flat = flat if flat is not None else self.flat.clone()
        self.grad = grad if grad is not None else self.grad.clone()
        self.m = m if m is not None else self.m.clone()
        self.v = v if v is not None else self.v.clone()
        self.flat_buffers = flat_buffers if flat_buffers is not None else self.flat_buffers.clone()
        for n, t in list(self.P.items()):
            off = t.storage_offset()
            self.P[n] = self.flat[off:off + t.numel()].view(t.shape)
            self.G[n] = self.grad[off:off + t.numel()].view(t.shape)
        for n, t in list(self.buffers.items()):
            off = t.storage_offset()
            self.buffers[n] = self.flat_buffers[off:off + t.numel()].view(t.shape)
        _lib.check(lib.ps_trainer_bind(self._h, _p(self.flat), _p(self.grad), _p(self.m), _p(self.v), _p(self.flat_buffers)))
        return old

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().ps_trainer_destroy(h)

    def __del__(self):
        try:
            self.close()  # (ps_trainer_destroy only touches the trainer's own pool, events and stream: safe after the context is gone)
        except Exception:
            pass

    def num_params(self):
        return self.flat.numel()

    def sync_buffers(self, dist):
        """BatchNorm moving statistics averaged over the ranks (they differ between GPUs when sync_bn is off: every rank has
        seen its own clouds).  COLLECTIVE: every rank must call it (it all-reduces) -- call it on all ranks, then let rank 0 write
        the checkpoint."""
        if dist is None or not self.buffers:
            return
        allreduce_mean_(self.flat_buffers, dist)

    def export_params(self, dist=None):
        """Parameter dict in the reference's variable names.  With `dist` this is a COLLECTIVE call (sync_buffers): the pattern
        `if rank == 0: save(trainer.export_params(dist))` deadlocks the other ranks -- call export_params(dist) on every rank and
        save on one.  Without `dist` in a multi-GPU run with per-GPU BatchNorm statistics the export holds THIS rank's moving
        statistics only (the reference has a single set): a warning says so once."""
        if dist is None and self._world > 1 and not self.sync_bn and not self._warned_export:
            import warnings
            warnings.warn("Trainer.export_params() without `dist` in a %d-rank run with per-GPU BatchNorm statistics: the exported moving "
                          "statistics are rank-local; call export_params(dist) (or sync_buffers(dist)) on ALL ranks first" % self._world)
            self._warned_export = True
        self.sync_buffers(dist)
        out = {n: self.P[n].detach().cpu().numpy().copy() for n in self.names}
        out.update({n: b.cpu().numpy().copy() for n, b in self.buffers.items()})
        return out

    _world = 1

    # ---- graph pieces ------------------------------------------------------------------------------------------------
    def _conv(self, t, x, scope, bn=True, act=True, transposed=False, out=None, fp32_only=False):
        y = t.linear(x, self.P[scope + "/weights"], self.P[scope + "/biases"], self.G[scope + "/weights"], self.G[scope + "/biases"], transposed,
                     fp32_only=fp32_only)
        if bn:
            s = scope + "/batch_normalization"
            y = t.bn_act(y, self.P[s + "/gamma"], self.P[s + "/beta"], self.G[s + "/gamma"], self.G[s + "/beta"], self.buffers[s + "/moving_mean"],
                         self.buffers[s + "/moving_variance"], act, out=out)
        return y

    def _conv_bn_square(self, t, x, scope, out=None):
        """conv (c -> c) + BatchNorm + LeakyReLU: the recompute form where it is compiled (c in 8 / 16 / 32 / 64), else the three ops"""
        W = self.P[scope + "/weights"]
        if not (self.fused_convbn and W.shape[0] == W.shape[1] and t.L.ps_op_conv_bn_train_supported(W.shape[0])
                and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0):
            return self._conv(t, x, scope, out=out)
        s = scope + "/batch_normalization"
        return t.conv_bn_act(x, W, self.P[scope + "/biases"], self.G[scope + "/weights"], self.G[scope + "/biases"], self.P[s + "/gamma"],
                             self.P[s + "/beta"], self.G[s + "/gamma"], self.G[s + "/beta"], self.buffers[s + "/moving_mean"],
                             self.buffers[s + "/moving_variance"], out=out)

    def _att_split(self, t, f_src, idx, f_xyz, name, B):
        agg = t.attpool_split(f_src, idx, f_xyz, self.P[name + "fc/kernel"], self.G[name + "fc/kernel"], B)
        return self._conv(t, agg, name + "mlp")

    def _att_pre(self, t, f_src, idx, fcat, f_xyz, name, K, B):
        """att_pooling with the score product in the pre-product form of the inference kernels: fset . Wfc = (f . Wfc[:h])[idx] + f_xyz . Wfc[h:]
        -- the [N*K]-row GEMMs (forward, input gradient, weight gradient) shrink to half their K, the other half runs on N rows.  Used where
        those GEMMs are bound by the matrix pipe (d >= 256); fset (the values of the weighted sum) is still the concat buffer."""
        W, gW = self.P[name + "fc/kernel"], self.G[name + "fc/kernel"]
        h = f_src.shape[1]
        s = t.gather(t.linear(f_src, W[:h], None, gW[:h], None), idx, B)
        s = t.linear(f_xyz, W[h:], None, gW[h:], None, into=s)
        return self._conv(t, t.softpool(fcat, s, K), name + "mlp")

    def _att(self, t, fcat, name, K):
        W, gW = self.P[name + "fc/kernel"], self.G[name + "fc/kernel"]
        if self.fused_att and t.L.ps_op_att_pool_train_supported(K, fcat.shape[1]):
            agg = t.attpool(fcat, W, gW, K)   # levels whose [N*K, d] tensors are large: one kernel per direction
        else:
            s = t.linear(fcat, W, None, gW, None)
            agg = t.softpool(fcat, s, K)
        return self._conv(t, agg, name + "mlp")

    def forward(self, t, pyr, features):
        cfg, L = self.cfg, self.cfg.num_layers
        B, K = features.shape[0], cfg.k_n
        lib, h = t.L, t.h
        x = features.reshape(-1, features.shape[-1]).contiguous()
        x.requires_grad_flag = False
        f = t.linear(x, self.P["fc0/kernel"], self.P["fc0/bias"], self.G["fc0/kernel"], self.G["fc0/bias"])
        f = t.bn_act(f, self.P["batch_normalization/gamma"], self.P["batch_normalization/beta"], self.G["batch_normalization/gamma"],
                     self.G["batch_normalization/beta"], self.buffers["batch_normalization/moving_mean"],
                     self.buffers["batch_normalization/moving_variance"], True)
        enc = []
        for i in range(L):
            n = "Encoder_layer_%d" % i
            idx = pyr.neigh_idx[i]
            N = idx.shape[1]
            feature = f
            f_pc = self._conv(t, feature, n + "mlp1")
            hloc = self.P[n + "LFAmlp1/weights"].shape[1]
            locse_fused = self.fused_locse and lib.ps_op_locse_train_supported(K, hloc)
            if not locse_fused:
                rel = torch.empty((B * N * K, 10), dtype=torch.float32, device=x.device)
                _lib.check(lib.ps_op_relative_pos_encoding(h, _p(pyr.xyz[i]), _p(idx), B, N, K, _p(rel)))
                rel.requires_grad_flag = False

            def locse(out=None):
                if not locse_fused:
                    return self._conv(t, rel, n + "LFAmlp1", out=out, fp32_only=True)
                s, sb = n + "LFAmlp1", n + "LFAmlp1/batch_normalization"
                return t.locse_bn_act(pyr.xyz[i], idx, B, self.P[s + "/weights"], self.P[s + "/biases"], self.G[s + "/weights"], self.G[s + "/biases"],
                                      self.P[sb + "/gamma"], self.P[sb + "/beta"], self.G[sb + "/gamma"], self.G[sb + "/beta"],
                                      self.buffers[sb + "/moving_mean"], self.buffers[sb + "/moving_variance"], out=out)
            # tf.concat([f_neighbours, f_xyz]) (RandLANet.py:328,332): both producers write their column block of the concat
            # buffer directly (no concat copy forward, no split copies backward)
            hc = f_pc.shape[1]
            if self.fused_att and lib.ps_op_att_pool_train_supported(K, 2 * hc):
                # gather_neighbour + concat + att_pooling's core as one kernel per direction: neither the gathered rows nor the concat
                # buffer nor the gathered half of its gradient exist (the backward scatter-adds into f_pc's gradient itself)
                f_xyz = locse()
                f_agg = self._att_split(t, f_pc, idx, f_xyz, n + "LFAatt_pooling_1", B)
                f_xyz2 = self._conv_bn_square(t, f_xyz, n + "LFAmlp2")
                f_agg2 = self._att_split(t, f_agg, idx, f_xyz2, n + "LFAatt_pooling_2", B)
            else:
                cat1 = torch.empty((B * N * K, 2 * hc), dtype=torch.float32, device=x.device)
                f_xyz = locse(out=cat1[:, hc:])
                pre = (not self.mlp_bf16) and 2 * hc >= 256  # (d = 128: measured slower, 58.1 vs 57.0 ms -- HBM bound there; bf16 mode: its yardstick rounds the operands of the ONE d x d product)
                f_nb = t.gather(f_pc, idx, B, out=cat1[:, :hc])
                fcat1 = t.concat_views(cat1, f_nb, f_xyz)
                f_agg = self._att_pre(t, f_pc, idx, fcat1, f_xyz, n + "LFAatt_pooling_1", K, B) if pre else self._att(t, fcat1, n + "LFAatt_pooling_1", K)
                cat2 = torch.empty((B * N * K, 2 * hc), dtype=torch.float32, device=x.device)
                f_xyz2 = self._conv_bn_square(t, f_xyz, n + "LFAmlp2", out=cat2[:, hc:])
                f_nb2 = t.gather(f_agg, idx, B, out=cat2[:, :hc])
                fcat2 = t.concat_views(cat2, f_nb2, f_xyz2)
                f_agg2 = self._att_pre(t, f_agg, idx, fcat2, f_xyz2, n + "LFAatt_pooling_2", K, B) if pre else self._att(t, fcat2, n + "LFAatt_pooling_2", K)
            a = self._conv(t, f_agg2, n + "mlp2", act=False)
            b = self._conv(t, feature, n + "shortcut", act=False)
            f_enc = t.add_lrelu(a, b)
            f = t.maxpool(f_enc, pyr.sub_idx[i], B)
            if i == 0:
                enc.append(f_enc)
            enc.append(f)
        f = self._conv(t, enc[-1], "decoder_0")
        for j in range(L):
            up = t.gather(f, pyr.interp_idx[-j - 1], B)
            f = self._conv(t, t.cat(enc[-j - 2], up), "Decoder_layer_%d" % j, transposed=True)
        f = self._conv(t, f, "fc1")
        f = self._conv(t, f, "fc2")
        # every rank draws its own mask (N GPUs x 1 cloud behaves like 1 GPU x N clouds, where the clouds sit at different element offsets)
        f = t.dropout(f, self.keep_prob, 0x9e3779b9 * (self.step + 1) + 0x85ebca6b * self._rank)
        return self._conv(t, f, "fc", bn=False, act=False)

    def train_step(self, pyr, features, labels, dist=None):
        """One optimisation step on the batch; returns the loss (device scalar tensor).  With `dist` (an initialised
        torch.distributed) the flat gradient buffer is averaged over ranks with one all-reduce before Adam."""
        own = getattr(self.ctx, "_stream", None)
        if own is None:
            # torch's helper ops (cat, add_, mul_) run on torch's current stream: the kernels behind the C ABI must run there too
            self.ctx.use_torch_stream()
            return self._train_step(pyr, features, labels, dist)
        with torch.cuda.stream(own):  # a context with its own stream (a pipeline lane): torch's helper ops follow it
            return self._train_step(pyr, features, labels, dist)

    def _collective(self, dist):
        """torch.distributed's all-reduce as the C callback of ps_trainer_set_collective.  The library hands over a raw device pointer
        on the context's stream (= torch's current stream here); it is wrapped without a copy through __cuda_array_interface__."""
        if self._coll is not None and self._coll[1] is dist:
            return self._coll[0]
        device = self.device

        class _Raw:
            def __init__(self, ptr, count, f64):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8" if f64 else "<f4", "data": (int(ptr), False), "version": 2}

        def cb(user, buf, count, dtype, stream):
            try:
                t = torch.as_tensor(_Raw(buf, count, dtype == 1), device=device)
                dist.all_reduce(t)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import sys
                print("all-reduce callback failed: %r" % (e,), file=sys.stderr)
                return 1

        fn = _lib.PS_ALLREDUCE_FN(cb)
        self._coll = (fn, dist)
        return fn

    def _train_step(self, pyr, features, labels, dist):
        if self.engine == "python":
            return self._train_step_python(pyr, features, labels, dist)
        lib = _lib.lib()
        self._rank = dist.get_rank() if dist is not None else 0
        self._world = dist.get_world_size() if dist is not None else 1
        # (one rank under a process group: the library skips the callback -- a one-rank mean is the identity, the plain single-rank step
        #  is 15 % faster -- unless asked with PS_COLLECTIVE_AT_WORLD_ONE: bench.py sets collective_at_world_one to measure the calls)
        if dist is not None and (self._world > 1 or self.collective_at_world_one):
            flags = (1 if self.sync_bn else 0) | (2 if self.collective_at_world_one else 0)  # (2 = PS_COLLECTIVE_AT_WORLD_ONE)
            _lib.check(lib.ps_trainer_set_collective(self._h, self._collective(dist), None, self._world, self._rank, flags))
        else:
            _lib.check(lib.ps_trainer_set_collective(self._h, _lib.PS_ALLREDUCE_FN(), None, 1, 0, 0))  # NULL callback: single rank
        _lib.check(lib.ps_trainer_set_options(self._h, ctypes.byref(self._options())))
        _lib.check(lib.ps_trainer_set_step(self._h, self.step))
        B, n0 = features.shape[0], features.shape[1]
        feats = features.reshape(-1, features.shape[-1]).contiguous()
        if feats.dtype != torch.float32:
            feats = feats.float()
        lab = labels.reshape(-1).to(torch.int32).contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=feats.device)
        logits = torch.empty((B * n0, self.cfg.num_classes), dtype=torch.float32, device=feats.device)
        _lib.check(lib.ps_randla_train_step(self._h, ctypes.byref(pyr.struct), _p(feats), _p(lab), _p(self.class_weights), _p(loss), _p(logits)))
        self.step = int(lib.ps_trainer_get_step(self._h))
        self.last_logits = logits
        return loss

    def backward_only(self, pyr, features, labels):
        """ps_randla_backward: training-mode forward + loss + backward, gradients left in self.grad / self.G (this rank's, no collective,
        no optimiser step) -- for hosts that run their own gradient synchronisation or optimiser."""
        lib = _lib.lib()
        _lib.check(lib.ps_trainer_set_options(self._h, ctypes.byref(self._options())))
        _lib.check(lib.ps_trainer_set_step(self._h, self.step))
        B, n0 = features.shape[0], features.shape[1]
        feats = features.reshape(-1, features.shape[-1]).contiguous().float()
        lab = labels.reshape(-1).to(torch.int32).contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=feats.device)
        logits = torch.empty((B * n0, self.cfg.num_classes), dtype=torch.float32, device=feats.device)
        own = getattr(self.ctx, "_stream", None)
        if own is None:
            self.ctx.use_torch_stream()
        _lib.check(lib.ps_randla_backward(self._h, ctypes.byref(pyr.struct), _p(feats), _p(lab), _p(self.class_weights), _p(loss), _p(logits)))
        self.last_logits = logits
        return loss

    def set_profile(self, on=True):
        """Per-section device time of every following step (ps_trainer_set_profile); read with profile()."""
        _lib.check(_lib.lib().ps_trainer_set_profile(self._h, 1 if on else 0))

    def profile(self):
        rows = (_lib.PsTimingRow * 64)()
        n = ctypes.c_int(0)
        _lib.check(_lib.lib().ps_trainer_profile(self._h, rows, 64, ctypes.byref(n)))
        return [(rows[i].name.decode(), rows[i].ms) for i in range(n.value)]

    def collective_stats(self):
        """Collectives of the last step (ps_trainer_collective_stats): dict(calls, bytes, host_ms, device_ms); device_ms is filled on
        profiled steps only (set_profile)."""
        calls, nbytes = ctypes.c_int64(0), ctypes.c_int64(0)
        host, dev = ctypes.c_double(0.0), ctypes.c_double(0.0)
        _lib.check(_lib.lib().ps_trainer_collective_stats(self._h, ctypes.byref(calls), ctypes.byref(nbytes), ctypes.byref(host), ctypes.byref(dev)))
        return dict(calls=calls.value, bytes=nbytes.value, host_ms=host.value, device_ms=dev.value)

    def pool_peak_bytes(self):
        return int(_lib.lib().ps_trainer_pool_peak_bytes(self._h))

    def _train_step_python(self, pyr, features, labels, dist):
        lib, h = _lib.lib(), self.ctx.handle
        self._rank = dist.get_rank() if dist is not None else 0
        self._world = dist.get_world_size() if dist is not None else 1
        t = Tape(self.ctx, sync=dist if (self.sync_bn and dist is not None) else None, bf16=self.mlp_bf16)
        if self.mlp_bf16:
            _lib.check(lib.ps_set_train_gemm_bf16(h, 1))
        try:
            logits = self.forward(t, pyr, features)
            R, C = logits.shape
            loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
            dlogits = torch.empty_like(logits)
            lab = labels.reshape(-1).to(torch.int32).contiguous()
            if self.label_map is not None:
                inside = (lab >= 0) & (lab < self.label_map.numel())
                lab = torch.where(inside, self.label_map[lab.clamp(0, self.label_map.numel() - 1).long()], torch.full_like(lab, -1)).contiguous()
            _lib.check(lib.ps_op_weighted_ce(h, _p(logits), _p(lab), _p(self.class_weights), R, C, _p(loss), _p(dlogits)))
            t.backward(logits, dlogits)
        finally:
            if self.mlp_bf16:  # the context may be shared with inference-side op calls: never leave the mode on
                _lib.check(lib.ps_set_train_gemm_bf16(h, 0))
        if dist is not None:
            allreduce_mean_(self.grad, dist)
        self.step += 1
        _lib.check(lib.ps_op_adam(h, _p(self.flat), _p(self.grad), _p(self.m), _p(self.v), self.flat.numel(), self.lr, 0.9, 0.999, 1e-8, self.step))
        self.last_logits = logits
        return loss
