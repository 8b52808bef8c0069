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

// Surface-area cost of a packed tree (any arity): a node's box is tested when its parent's box
// was hit, so a random ray that hits the root tests 1 + sum over the other nodes of
// area(parent) / area(root) boxes.
double tree_cost(const std::vector<NodeRec> &nodes);

// Removes inner nodes that are hit too often to be worth testing: with p = area(node) /
// area(parent), keeping a node with two children costs 1 + 2p tests per ray through the parent,
// dropping it (its children become the parent's) costs 2 -- drop it when p > threshold (0.5).
// The skip list does not care how many children a node has.
std::vector<NodeRec> contract_walk_tree(const std::vector<NodeRec> &nodes, double threshold);

// Binned-SAH tree (16 bins per axis, one leaf per node of `leaves`) over the leaf
// records of `packed` -- same boxes, same leaf indices --, pre-order skip list like
// the input; inner boxes are exact unions of their children's boxes, so the result
// is nested by construction.  `packed` must be a regular binary tree.
std::vector<NodeRec> rebuild_walk_tree(const std::vector<NodeRec> &packed);

// The same tree with every node's children in the order of their boxes' distance from `eye` (the nearest point of the
// box; the box's centre among boxes the eye lies in), nearest first.  The order of a node's children is the order a walk
// meets them in, and every PRIMARY ray starts at the eye (reference src/intersect_kernel.cl:284-286: (0, 0, 2),
// whatever the options): a closest-hit walk then tends to find its nearest triangle before it meets the boxes behind it
// -- which a lane with a hit does not enter (kernels/primary.hip.h, far_limit).  No result depends on the order: the
// closest hit is the minimum of (distance, leaf), an any-hit ray needs some accepted triangle.
std::vector<NodeRec> nearest_children_first(const std::vector<NodeRec> &nodes, const double eye[3]);

}  // namespace ocrt
