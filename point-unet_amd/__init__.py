"""point_unet_amd -- MI355X-native PointSegment hot path (KNN pyramid + RandLA-Net forward) of Point-Unet.

Python host code over libpointseg_hip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/pointseg.h).  Module names mirror the reference's PointSegment/ tree:

    helper_tool.DataProcessing.knn_search / grid_sub_sampling, ConfigBraTS, ConfigPancreas   (helper_tool.py)
    RandLANet.Network                                                                         (RandLANet.py)
    utils.nearest_neighbors.lib.python.nearest_neighbors.knn / knn_batch                      (knn.pyx)
    utils.cpp_wrappers.cpp_subsampling.grid_subsampling.compute                               (wrapper.cpp)
    pyramid.build_pyramid                                                                     (runBraTS.py tf_map)
"""
import os as _os

# Environment contract of the pipelined forward (pipeline.py): HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default
# 4) and reads the variable ONCE, when the runtime initialises.  ForwardPipeline's four lanes need a queue each next to the null stream
# and RCCL's: 6 measured best (0.84 ms per 180 000-point cloud; with the default 4 the lanes share queues: 1.33 vs 1.23 ms in round 2's
# sweep).  Set here -- importing this package is the first thing a user of the path does, before any HIP call -- unless
# the user chose a value; ForwardPipeline warns when the runtime was already up with too few queues.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
# (nothing else is touched at import: HSA_ENABLE_IPC_MODE_LEGACY=0, which RCCL / cross-process tensors need on this pool's driver, is a
#  deployment setting of the LAUNCHER -- bench.py and tests/conftest.py set it for their own processes)

from ._lib import PointSegError, lib  # noqa: E402,F401

__all__ = ["PointSegError", "lib"]
