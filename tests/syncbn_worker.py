"""Worker of tests/test_gpu_train.py::test_sync_bn_ranks_equal_one_rank_with_the_batch: rank r trains on cloud r of a WORLD_SIZE-cloud
batch with BatchNorm statistics shared over the process group (gloo here: all ranks use the same GPU), and writes its loss, the averaged
gradient buffer and the step's collective counts (shared statistics, then one more step with per-GPU statistics) to <out>.rank<r>.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    import torch.distributed as dist
    import netcase
    import test_gpu_train as T
    out = sys.argv[1]
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    cfg, xyz, feats = T.syncbn_case(int(os.environ["WORLD_SIZE"]))
    tr, pyr, params, labels, cw, _ = T._setup(cfg, xyz[rank:rank + 1], feats[rank:rank + 1], labels=T.syncbn_labels(cfg, xyz)[rank:rank + 1],
                                              sync_bn=True, oracle_pyramid=False)
    loss = tr.train_step(pyr, torch.from_numpy(feats[rank:rank + 1]).cuda(), torch.from_numpy(labels).cuda(), dist=dist)
    torch.cuda.synchronize()
    st = tr.collective_stats()
    grad, flat = tr.grad.cpu().numpy(), tr.flat.cpu().numpy()
    # the same rank once more with per-GPU statistics (the reference's own batch-1 semantics, helper_tool.py:29): ONE call, the gradient buffer
    tr2, pyr2, _, _, _, _ = T._setup(cfg, xyz[rank:rank + 1], feats[rank:rank + 1], labels=T.syncbn_labels(cfg, xyz)[rank:rank + 1],
                                     sync_bn=False, oracle_pyramid=False)
    tr2.train_step(pyr2, torch.from_numpy(feats[rank:rank + 1]).cuda(), torch.from_numpy(labels).cuda(), dist=dist)
    torch.cuda.synchronize()
    st2 = tr2.collective_stats()
    np.savez(out + ".rank%d.npz" % rank, loss=float(loss), grad=grad, flat=flat, calls_sync_bn=st["calls"], bytes_sync_bn=st["bytes"],
             calls_local_bn=st2["calls"], bytes_local_bn=st2["bytes"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
