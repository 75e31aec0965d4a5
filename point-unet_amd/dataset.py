"""Input side of the hot path: from a prepared .ply volume to the (xyz, features, labels, point index) tuple the network
consumes -- the generators of PointSegment/runBraTS.py:93-130 and runPancreas.py:96-118 as plain functions.

BraTS: all tumour voxels plus a uniform sample of background voxels up to cfg.num_points, shuffled (the shuffle is what
makes the later prefix slices `xyz[:, :N//r]` a random sub-sample, runBraTS.py:114,147).  Pancreas: the whole cloud in file
order.  Features are the reference's: xyz is concatenated in front of the modalities by tf_map (runBraTS.py:142)."""
import numpy as np

from .helper_ply import read_ply

BRATS_MODALITIES = ("t1ce", "t1", "flair", "t2")  # runBraTS.py:105


def sample_brats_cloud(data, num_points, rng=None):
    """data: structured array with x, y, z, the four modalities and `class` (a prepared BraTS .ply).  Returns
    (xyz f32 [n,3], modalities f32 [n,4], labels, queried_idx i32 [n]) with n = max(num_points, #tumour voxels)."""
    rng = rng or np.random.default_rng()
    labels = np.asarray(data["class"])
    tumour = np.flatnonzero(labels > 0)
    background = np.flatnonzero(labels == 0)
    need = num_points - len(tumour)
    if need > len(background):
        raise ValueError("cloud has %d points, fewer than num_points = %d" % (len(labels), num_points))
    picked = rng.choice(background, size=max(need, 0), replace=False)
    idx = rng.permutation(np.concatenate([tumour, picked]))
    xyz = np.stack([data["x"], data["y"], data["z"]], axis=1)[idx].astype(np.float32)
    mods = np.stack([data[m] for m in BRATS_MODALITIES], axis=1)[idx].astype(np.float32)
    return xyz, mods, labels[idx], idx.astype(np.int32)


def pancreas_cloud(data):
    """data: structured array with x, y, z, value, class (a prepared Pancreas .ply): the whole cloud, file order."""
    xyz = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
    value = np.asarray(data["value"], dtype=np.float32).reshape(-1, 1)
    return xyz, value, np.asarray(data["class"]), np.arange(len(xyz), dtype=np.int32)


def network_features(xyz, modalities):
    """tf_map's `features = concat(xyz, features)` (runBraTS.py:142)."""
    return np.concatenate([xyz, modalities], axis=-1).astype(np.float32)


def load_brats_ply(path, num_points, rng=None):
    return sample_brats_cloud(read_ply(path), num_points, rng)


def load_pancreas_ply(path):
    return pancreas_cloud(read_ply(path))
