#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (run in the BUILD container only).

KNN / pyramid / grid-subsampling vectors come from the REAL reference C++ compiled unmodified into oracle/_ref
(`make -C oracle ref`; knn_.cxx + nanoflann.hpp, grid_subsampling.cpp + cloud.cpp).  The network vector comes from
the float64 NumPy restatement (oracle/randla_oracle.py) -- the reference's TensorFlow graph cannot be executed
here, so that one is a regression pin, not a reference pin.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from conftest import brats_cloud, uniform_cloud  # noqa: E402
from oracle import bindings as ob  # noqa: E402
from oracle import randla_oracle as ro  # noqa: E402

assert ob.have_ref(), "build the reference first: make -C oracle ref"


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
    print(name, {k: getattr(v, "shape", v) for k, v in kw.items()})


def knn_case(name, support, queries, K):
    idx = ob.ref_knn_batch(support, queries, K, omp=True)
    save("knn_" + name, support=support, queries=queries, K=np.int64(K), idx=idx)


rng = np.random.default_rng(0)
u = uniform_cloud(3000, 11)[None]
lat = brats_cloud(3000, 12, grid=(24, 24, 16))[None]            # heavy ties
knn_case("uniform_k16", u, u, 16)
knn_case("lattice_k16", lat, lat, 16)
knn_case("lattice_k32", lat, lat, 32)
knn_case("lattice_k1", lat, lat, 1)
knn_case("upsample_k1", lat[:, :750], lat, 1)                     # queries outside the support bbox
d = np.repeat(rng.random((40, 3), dtype=np.float32), 30, axis=0)
rng.shuffle(d)
knn_case("duplicates_k16", d[None], d[None], 16)
tiny = rng.random((1, 7, 3), dtype=np.float32)
knn_case("fewer_than_k", tiny, tiny, 16)                          # n < K: trailing slots are 0
knn_case("batch2_k16", np.stack([uniform_cloud(1500, 13), brats_cloud(1500, 14, grid=(20, 20, 16))]),
         (rng.random((2, 600, 3), dtype=np.float32) * 1.5 - 0.25), 16)

# index pyramid: the loop of PointSegment/runBraTS.py:147-156 driven with the real reference knn_batch
xyz = brats_cloud(6000, 15, grid=(32, 32, 24))[None]
pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: ob.ref_knn_batch(s, q, k, omp=True), xyz, 16, [4, 4, 4, 4, 2])
save("pyramid_brats6000", xyz=xyz, K=np.int64(16), ratios=np.array([4, 4, 4, 4, 2]),
     **{"neigh_%d" % i: nbr[i] for i in range(5)}, **{"sub_%d" % i: pool[i] for i in range(5)},
     **{"interp_%d" % i: up[i] for i in range(5)})

# grid subsampling (labels constant per voxel block => no majority ties, whose order the reference leaves
# implementation defined, grid_subsampling.cpp:100-101); outputs stored after the canonical row sort
p = rng.random((20000, 3), dtype=np.float32) * np.array([1.0, 0.7, 0.4], np.float32)
f = rng.standard_normal((20000, 4)).astype(np.float32)
dl = 0.03
lab = (np.floor(p[:, 0] / np.float32(0.09)).astype(np.int32) % 5)
gp, gf, gl = ob.canonical_rows(*ob.ref_grid_subsample(p, f, lab, dl))
save("grid_all", points=p, features=f, classes=lab, sampleDl=np.float32(dl), out_points=gp, out_features=gf, out_classes=gl)
gp, _, _ = ob.canonical_rows(*ob.ref_grid_subsample(p, None, None, dl))
save("grid_points_only", points=p, sampleDl=np.float32(dl), out_points=gp)
neg = (rng.random((5000, 3), dtype=np.float32) - 0.5) * 4                # negative coordinates: floor() matters
gp, gf, _ = ob.canonical_rows(*ob.ref_grid_subsample(neg, f[:5000], None, 0.25))
save("grid_negative_coords", points=neg, features=f[:5000], sampleDl=np.float32(0.25), out_points=gp, out_features=gf)

# network: BASELINE config 1 (18 000 points, K=16, 2 layers), float64 restatement, seeded weights with
# non-trivial BatchNorm statistics.  Inputs are regenerated from seeds by tests/netcase.py.
import netcase  # noqa: E402
from point_unet_amd import weights  # noqa: E402

cfg, xyz, feats = netcase.config1()
params = weights.init_params(cfg, seed=2, randomize_bn=True)
pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: ob.ref_knn_batch(s, q, k, omp=True), xyz, cfg.k_n, cfg.sub_sampling_ratio)
tap = {}
logits = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64, tap=tap)
save("net_config1", logits=logits.astype(np.float32), enc0_rows=tap["enc0"][0, :64].astype(np.float32),
     pool1_rows=tap["pool1"][0, :64].astype(np.float32), neigh_0_rows=nbr[0][0, :256], interp_0_rows=up[0][0, :256])

# binary PLY (N2): a small BraTS-shaped cloud written by the REAL reference writer (PointSegment/helper_ply.py write_ply,
# imported here only) -- the committed .ply is data; tests check this package's reader against it and this package's writer
# for byte equality with it.
sys.path.insert(0, "/root/reference/PointSegment")
import helper_ply as ref_ply  # noqa: E402

prng = np.random.default_rng(21)
pxyz = brats_cloud(500, 22, grid=(16, 16, 12))
pmods = prng.standard_normal((500, 4)).astype(np.float32)
pcls = (prng.random(500) < 0.2).astype(np.int32) * prng.integers(1, 4, 500).astype(np.int32)
ref_ply.write_ply(os.path.join(HERE, "brats_example.ply"), [pxyz, pmods, pcls], ["x", "y", "z", "t1ce", "t1", "flair", "t2", "class"])
back = ref_ply.read_ply(os.path.join(HERE, "brats_example.ply"))
save("ply_example", xyz=pxyz, mods=pmods, cls=pcls, read_back_x=back["x"], read_back_class=back["class"])

# volume -> cloud (N4): the REAL reference functions itensity_normalize_one_volume and convert_pc2ply, cut out of
# PointSegment/utils/dataPrepareBraTS.py with ast (the module itself needs nibabel) and run on a small synthetic case with
# their file output captured.  Only the resulting arrays are committed.
import ast  # noqa: E402
import pickle  # noqa: E402
import tempfile  # noqa: E402

from sklearn.neighbors import KDTree  # noqa: E402

ref_src = open("/root/reference/PointSegment/utils/dataPrepareBraTS.py").read()
tree = ast.parse(ref_src)
wanted = {}
for node in ast.walk(tree):
    if isinstance(node, ast.FunctionDef) and node.name in ("itensity_normalize_one_volume", "convert_pc2ply"):
        wanted[node.name] = ast.Module(body=[node], type_ignores=[])
captured = {}
tmpd = tempfile.mkdtemp()


class _DP:
    @staticmethod
    def grid_sub_sampling(points, features, labels, dl):
        return ob.ref_grid_subsample(points, features, labels.astype(np.int32), dl)


def _write_ply(path, fields, names):
    captured[os.path.basename(os.path.dirname(path))] = [np.asarray(f) for f in fields]
    return True


env = dict(np=np, os=os, pickle=pickle, KDTree=KDTree, DP=_DP, write_ply=_write_ply, out_format=".ply", sub_grid_size=0.11,
           original_pc_folder=os.path.join(tmpd, "full"), sub_pc_folder=os.path.join(tmpd, "sub"), print=lambda *a, **k: None)
os.makedirs(env["original_pc_folder"]); os.makedirs(env["sub_pc_folder"])
for name in ("itensity_normalize_one_volume", "convert_pc2ply"):
    exec(compile(wanted[name], "dataPrepareBraTS.py", "exec"), env)
vrng = np.random.default_rng(31)
raw = (vrng.random((4, 22, 18, 14)) * 900).astype(np.int16)   # BraTS NIfTI files hold int16 intensities: NumPy reduces them in float64
hole = vrng.random((22, 18, 14)) < 0.45
raw[:, hole] = 0                                      # background voxels
raw[1, vrng.random((22, 18, 14)) < 0.1] = 0           # a modality may be zero where the others are not
# labels constant inside every sub-sampling cell (slabs cut at multiples of the 0.11 cell size): no majority ties, whose
# resolution the reference leaves to unordered_map iteration order (grid_subsampling.cpp:100-101)
segv = (np.select([np.arange(22) < 5, np.arange(22) < 10, np.arange(22) < 15], [0, 1, 2], 4)[:, None, None] * np.ones((1, 18, 14), np.int64) * (~hole)).astype(np.int32)
vol5 = np.empty((5, 22, 18, 14))
for m in range(4):
    vol5[m] = env["itensity_normalize_one_volume"](raw[m])
seg3 = segv.copy(); seg3[seg3 == 4] = 3
vol5[4] = seg3
env["convert_pc2ply"](vol5, "case")
full, sub = captured["full"], captured["sub"]
origin = np.load(os.path.join(env["sub_pc_folder"], "case_xyz_origin.npy"))
proj_idx, proj_labels = pickle.load(open(os.path.join(env["sub_pc_folder"], "case_proj.pkl"), "rb"))
sp, sf, sl = ob.canonical_rows(sub[0], sub[1], np.asarray(sub[2]).reshape(-1, 1).astype(np.int32))
proj_d = np.linalg.norm(full[0].astype(np.float64) - sub[0][proj_idx].astype(np.float64), axis=1)
save("volume_to_cloud", raw=raw, seg=segv, xyz=full[0], colors=full[1], labels=full[2], xyz_origin=origin.astype(np.int32),
     sub_grid_size=np.float32(0.11), sub_xyz=sp, sub_colors=sf, sub_labels=sl, proj_dist=proj_d)
