"""Host-side mirror of PointSegment/helper_tool.py for the hot path: the configuration classes and the
DataProcessing facade over the native ops, now served by the HIP library.

    ConfigBraTS / ConfigPancreas        helper_tool.py:21-75   (same attribute names and values)
    DataProcessing.knn_search           helper_tool.py:84-94
    DataProcessing.grid_sub_sampling    helper_tool.py:123-143
    DataProcessing.get_class_weights    helper_tool.py:172-184
    DataProcessing.shuffle_idx / shuffle_list / IoU_from_confusions   helper_tool.py:110-121, 146-170
"""
import numpy as np

from .utils.cpp_wrappers.cpp_subsampling import grid_subsampling as cpp_subsampling
from .utils.nearest_neighbors.lib.python import nearest_neighbors


class ConfigBraTS:
    k_n = 16  # KNN
    num_layers = 5  # Number of layers
    num_points = 365000
    num_classes = 4  # Number of valid classes
    sub_grid_size = 0.01  # preprocess_parameter
    batch_size = 1
    val_batch_size = 1
    train_steps = 295
    val_steps = 74
    sub_sampling_ratio = [4, 4, 4, 4, 2]  # sampling ratio of random sampling at each layer
    d_out = [16, 64, 128, 256, 512, 1024, 2048]  # feature dimension
    noise_init = 3.5
    learning_rate = 1e-4
    lr_decays = {i: 0.95 for i in range(0, 500)}
    in_channels = 7  # xyz + 4 MR modalities (runBraTS.py:142: features = concat(xyz, features))


class ConfigPancreas:
    k_n = 16
    num_layers = 5
    num_points = 180000
    num_classes = 2
    sub_grid_size = 0.01
    batch_size = 1
    val_batch_size = 1
    sub_sampling_ratio = [4, 4, 4, 4, 2]
    d_out = [16, 64, 128, 256, 512, 1024, 2048]
    noise_init = 3.5
    learning_rate = 1e-3
    lr_decays = {i: 0.95 for i in range(0, 500)}
    in_channels = 4  # xyz + 1 CT value (runPancreas.py:118,125)


class DataProcessing:
    """Same static-method facade as the reference's DataProcessing; the native calls land in libpointseg_hip.so."""

    @staticmethod
    def knn_search(support_pts, query_pts, k):
        """K nearest support points of every query point, per batch element.
        support_pts [B,N1,3], query_pts [B,N2,3] float32 -> int32 [B,N2,k], ascending distance, nanoflann tie order
        (helper_tool.py:84-94: knn_batch(..., omp=True) cast to int32)."""
        return nearest_neighbors.knn_batch(support_pts, query_pts, k, omp=True).astype(np.int32)

    @staticmethod
    def grid_sub_sampling(points, features=None, labels=None, grid_size=0.1, verbose=0):
        """Voxel-grid subsampling: barycentre of the points (and mean of the features) per cell of side `grid_size`,
        majority vote for integer labels.  Returns points, then features and/or labels when they were given
        (helper_tool.py:123-143)."""
        extra = {}
        if features is not None:
            extra["features"] = features
        if labels is not None:
            extra["classes"] = labels
        return cpp_subsampling.compute(points, sampleDl=grid_size, verbose=verbose, **extra)

    # per-class point counts the reference hard-codes (helper_tool.py:172-184)
    _POINTS_PER_CLASS = {"BraTS20": (1, 1, 1, 1), "BraTS_Block64": (1403, 22, 80, 11), "Pancreas": (1, 1)}

    @staticmethod
    def get_class_weights(dataset_name):
        """Cross-entropy class weights 1 / (class frequency + 0.02), shape [1, C]."""
        counts = np.asarray(DataProcessing._POINTS_PER_CLASS[dataset_name], dtype=np.float64)
        return (1.0 / (counts / counts.sum() + 0.02))[None, :]

    @staticmethod
    def shuffle_idx(x):
        """Random permutation of the rows of x (helper_tool.py:110-114)."""
        return x[np.random.permutation(len(x))]

    @staticmethod
    def shuffle_list(data_list):
        """Random permutation of a Python list, as a new list (helper_tool.py:117-121)."""
        order = np.random.permutation(len(data_list))
        return [data_list[i] for i in order]

    @staticmethod
    def IoU_from_confusions(confusions):
        """Per-class IoU = TP / (TP + FP + FN) from confusion matrices [..., C, C] (rows = truth, columns = prediction);
        classes absent from a matrix take the mean IoU of the present ones (helper_tool.py:146-170)."""
        confusions = np.asarray(confusions)
        tp = np.diagonal(confusions, axis1=-2, axis2=-1)
        denom = confusions.sum(axis=-1) + confusions.sum(axis=-2) - tp
        iou = tp / (denom + 1e-6)
        present = confusions.sum(axis=-1) >= 1e-3
        mean_present = (iou * present).sum(axis=-1, keepdims=True) / (present.sum(axis=-1, keepdims=True) + 1e-6)
        return np.where(present, iou, mean_present)
