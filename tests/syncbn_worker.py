"""Worker of tests/test_gpu_train.py::test_sync_bn_two_ranks_equal_one_rank_batch_two: rank r trains on cloud r of a two-cloud
batch with BatchNorm statistics shared over the process group (gloo here: both ranks use the same GPU), and writes its
loss and the averaged gradient buffer to <out>.rank<r>.npz."""
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
    cfg, xyz, feats = T.syncbn_case()
    tr, pyr, params, labels, cw, _ = T._setup(cfg, xyz[rank:rank + 1], feats[rank:rank + 1], labels=T.syncbn_labels(cfg, xyz)[rank:rank + 1],
                                              sync_bn=True, oracle_pyramid=False)
    loss = tr.train_step(pyr, torch.from_numpy(feats[rank:rank + 1]).cuda(), torch.from_numpy(labels).cuda(), dist=dist)
    torch.cuda.synchronize()
    np.savez(out + ".rank%d.npz" % rank, loss=float(loss), grad=tr.grad.cpu().numpy(), flat=tr.flat.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
