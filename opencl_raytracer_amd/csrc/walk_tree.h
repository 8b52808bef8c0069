// walk_tree.h -- the tree the GPU walks.  Host only.
//
// The image does not depend on the shape of the tree (DESIGN.md
// section 3): a triangle counts only if its own tight box -- the uploaded leaf's
// box -- passes the slab test and the reference's triangle test accepts it;
// any-hit rays need "some accepted triangle" and the closest hit is the minimum of
// (distance, reference leaf index).  So upload may give the kernels a better tree
// over the same leaves than the one it was handed: fewer box tests per ray.
#pragma once
#include <cstdint>
#include <vector>

#include "device_types.h"

namespace ocrt {

// Surface-area cost of a packed tree: sum over all nodes of area(node) / area(root),
// the expected number of box tests of a random ray that hits the root.
double tree_cost(const std::vector<NodeRec> &nodes);

// Binned-SAH tree (16 bins per axis, one leaf per node of `leaves`) over the leaf
// records of `packed` -- same boxes, same leaf indices --, pre-order skip list like
// the input; inner boxes are exact unions of their children's boxes, so the result
// is nested by construction.  `packed` must be a regular binary tree.
std::vector<NodeRec> rebuild_walk_tree(const std::vector<NodeRec> &packed);

}  // namespace ocrt
