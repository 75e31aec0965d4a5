"""Host-side mirror of PointSegment/helper_tool.py for the hot path: the configuration classes and the
DataProcessing facade over the native ops, now served by the HIP library.

    ConfigBraTS / ConfigPancreas        helper_tool.py:21-75   (same attribute names and values)
    DataProcessing.knn_search           helper_tool.py:84-94
    DataProcessing.grid_sub_sampling    helper_tool.py:123-143
    DataProcessing.get_class_weights    helper_tool.py:172-184
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
    @staticmethod
    def knn_search(support_pts, query_pts, k):
        """
        :param support_pts: points you have, B*N1*3
        :param query_pts: points you want to know the neighbour index, B*N2*3
        :param k: Number of neighbours in knn search
        :return: neighbor_idx: neighboring points indexes, B*N2*k  (int32)
        """
        neighbor_idx = nearest_neighbors.knn_batch(support_pts, query_pts, k, omp=True)
        return neighbor_idx.astype(np.int32)

    @staticmethod
    def grid_sub_sampling(points, features=None, labels=None, grid_size=0.1, verbose=0):
        """Grid sub-sampling (barycentre for points and features, majority for labels).
        :param points: (N, 3) matrix of input points
        :param features: optional (N, d) matrix of features (floating number)
        :param labels: optional (N,) matrix of integer labels
        :param grid_size: parameter defining the size of grid voxels
        :return: sub_sampled points, with features and/or labels depending of the input
        """
        if (features is None) and (labels is None):
            return cpp_subsampling.compute(points, sampleDl=grid_size, verbose=verbose)
        elif labels is None:
            return cpp_subsampling.compute(points, features=features, sampleDl=grid_size, verbose=verbose)
        elif features is None:
            return cpp_subsampling.compute(points, classes=labels, sampleDl=grid_size, verbose=verbose)
        else:
            return cpp_subsampling.compute(points, features=features, classes=labels, sampleDl=grid_size,
                                           verbose=verbose)

    @staticmethod
    def get_class_weights(dataset_name):
        # pre-calculate the number of points in each category (helper_tool.py:172-184)
        num_per_class = []
        if dataset_name == 'BraTS20':
            num_per_class = np.array([1, 1, 1, 1])
        elif dataset_name == 'BraTS_Block64':
            num_per_class = np.array([1403, 22, 80, 11])
        elif dataset_name == 'Pancreas':
            num_per_class = np.array([1, 1])
        weight = num_per_class / float(sum(num_per_class))
        ce_label_weight = 1 / (weight + 0.02)
        return np.expand_dims(ce_label_weight, axis=0)
