"""One case of soak_train_r4.py looked at closely: is a deviation outside the bars the product's, or the case's own sensitivity (a leaky-ReLU /
max-pool decision on a value within fp32 rounding noise of the kink)?  Evaluates the oracle in float32 as well: the same tensors moving by the
same amount there means the case, not the kernels.  usage (GPU box): python profiles/tools/exp_soak_case.py [n0 B]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import netcase, test_gpu_train as T
from oracle import randla_train_oracle as rto
n0, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12001, 2)
cfg, xyz, feats = netcase.small_deep(n0, seed=n0, B=B)
tr, pyr, params, labels, cw, (pts, nbr, pool, up) = T._setup(cfg, xyz, feats, mlp_dtype="fp32")
loss = tr.train_step(pyr, torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda())
torch.cuda.synchronize()
want = rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, labels, cw, lr=1e-3, step=1)
alt = rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, labels, cw, lr=1e-3, step=1, dtype=torch.float32)
got = {n: tr.G[n].cpu().numpy() for n in tr.names}
for name, g in (("product vs float64 oracle", got), ("float32 oracle vs float64 oracle", alt["grads"])):
    rel, worst = T._grad_stats(g, want["grads"], tr.names)
    print("%s: grad rel L2 %.2e, worst %s" % (name, rel, [(round(w, 3), n) for w, n in worst[:4]]), flush=True)
rel, worst = T._grad_stats(got, alt["grads"], tr.names)
print("product vs float32 oracle: grad rel L2 %.2e, worst %s" % (rel, [(round(w, 3), n) for w, n in worst[:3]]))
