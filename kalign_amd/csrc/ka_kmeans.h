// ka_kmeans.h -- library-internal: the 2-means bisection of build_tree_kmeans on the device (ka_kmeans.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

struct KaKmNode {
        int left = -1, right = -1;           // indices into the node vector; left < 0: a leaf cluster
        std::vector<int> cluster;            // its members in the reference's order
};

// dm: numseq x 32 floats (host), 32 anchors.  nodes[0] is the root.  Returns 0, or 1 with `err` set.
int ka_kmeans_device(int device, hipStream_t stream, const float* dm, int numseq, std::vector<KaKmNode>& nodes, std::string& err);
