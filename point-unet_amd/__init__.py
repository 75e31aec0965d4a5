"""point_unet_amd -- MI355X-native PointSegment hot path (KNN pyramid + RandLA-Net forward) of Point-Unet.

Python host code over libpointseg_hip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/pointseg.h).  Module names mirror the reference's PointSegment/ tree:

    helper_tool.DataProcessing.knn_search / grid_sub_sampling, ConfigBraTS, ConfigPancreas   (helper_tool.py)
    RandLANet.Network                                                                         (RandLANet.py)
    utils.nearest_neighbors.lib.python.nearest_neighbors.knn / knn_batch                      (knn.pyx)
    utils.cpp_wrappers.cpp_subsampling.grid_subsampling.compute                               (wrapper.cpp)
    pyramid.build_pyramid                                                                     (runBraTS.py tf_map)
"""
from ._lib import PointSegError, lib  # noqa: F401

__all__ = ["PointSegError", "lib"]
