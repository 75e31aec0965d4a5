#!/bin/sh
# Counterpart of the reference's PointSegment/compile_op.sh:1-6 (which builds the nanoflann KNN Cython module
# and the grid-subsampling CPython module): builds the one HIP shared library that replaces both, plus the
# RandLA-Net forward kernels, for gfx950.
set -e
cd "$(dirname "$0")/csrc"
make -j"${JOBS:-8}"
