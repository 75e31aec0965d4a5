// kdtree_host.h -- host builder of the kd-tree in the device layout (see kdtree_host.hip).
#pragma once

#include <vector>

#include "kdtree.h"

namespace ps {

struct HostTree {
    int32_t n = 0;
    TreeMeta meta;
    std::vector<int32_t> vind;
    std::vector<int4> nodes;
    std::vector<float4> pts;
    TreeView view() const
    {
        TreeView v;
        v.nodes = nodes.data();
        v.pts = pts.data();
        v.meta = &meta;
        v.n = n;
        return v;
    }
};

void build_tree_host(const float* pts, int32_t n, HostTree& t);

}  // namespace ps
