// srt_thread.h -- host side of the stackless walks: thread links for trees in bvh.h's fixed visiting order.
//
// bvhNode::hit (bvh.h:97-105) visits left, then right, always, so "where the walk goes when this subtree is done" is a
// property of the tree: the right sibling's subtree for a left child, the parent's successor for a right child, "done"
// for a root.  With that successor stored in every node a ray needs no stack: a box hit goes to the first child, a miss
// (or a finished leaf) to the successor.  Two encodings, both built at srtUploadScene from the flattened node array
// (2 x float4 per node: (bmin.xyz, left) (bmax.xyz, right), node references = index * 32, primitives = ~(index << 1 |
// sphere)), both only for trees whose nodes have two node children or two primitive children (bvh.h:55-95 builds nothing
// else; a caller-built tree may -- then the function returns nothing and the kernels walk with a stack):
//   srtThreadLinks16   one word per node for the LDS-resident-tree kernels (DevScene::nodeThread)
//   srtHybridRecords   the path-pool kernel's hybrid records (DevScene::nodesWf): 32-bit references, nodes renumbered
// Pure host code (no HIP calls): tests/test_thread_links.py drives both through the test hooks of include/srt_hip_test.h and
// compares the stackless walks with the recursion.
#pragma once
#include <cstdint>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

#include "srt_device.h"

namespace srt_thread_detail {
inline int32_t refOf(const std::vector<float4>& nodes, size_t slot) {
  int32_t r;
  memcpy(&r, &nodes[slot].w, 4);
  return r;
}
// every node's children are two nodes or two primitives
inline bool uniformChildren(const std::vector<float4>& nodes) {
  for (size_t i = 0; i + 1 < nodes.size(); i += 2)
    if ((refOf(nodes, i) >= 0) != (refOf(nodes, i + 1) >= 0)) return false;
  return true;
}
}  // namespace srt_thread_detail

// 16 bits per reference (node INDEX, ~primitive, 0x8000 = done); per node: high half = the successor, low half = what
// follows a leaf's first object (its second object, or the successor again for a single-object leaf).  Leaves `out` empty
// when a reference does not fit 15 bits or the world is not a forest of such trees.
inline void srtThreadLinks16(const std::vector<float4>& nodes, const std::vector<int32_t>& world, int64_t numTriangles, int64_t numSpheres,
                             std::vector<int32_t>& out) {
  using namespace srt_thread_detail;
  out.clear();
  const size_t n = nodes.size() / 2;
  if (n == 0 || n >= 32767 || 2 * numTriangles >= 32766 || 2 * numSpheres + 1 >= 32766 || !uniformChildren(nodes)) return;
  const uint32_t kDone = 0x8000;
  auto ref16 = [&](int32_t r) { return (uint32_t)(r >= 0 ? SRT_NODE_INDEX(r) : r) & 0xffffu; };  // r: device reference
  out.assign(n, (int32_t)(kDone << 16 | kDone));
  std::vector<uint8_t> seen(n, 0);
  std::vector<std::pair<int32_t, uint32_t>> todo;  // (node index, its successor as 16 bits)
  for (int32_t wr : world)
    if (wr >= 0) todo.emplace_back(SRT_NODE_INDEX(wr), kDone);
  while (!todo.empty()) {
    const auto [i, after] = todo.back();
    todo.pop_back();
    if (i < 0 || (size_t)i >= n || seen[i]) {  // a node reached twice is not a tree: leave it to the stack walk
      out.clear();
      return;
    }
    seen[i] = 1;
    const int32_t l = refOf(nodes, 2 * (size_t)i), r = refOf(nodes, 2 * (size_t)i + 1);
    if (l >= 0) {
      out[i] = (int32_t)(after << 16 | after);
      todo.emplace_back(SRT_NODE_INDEX(r), after);     // the right subtree is followed by this node's successor
      todo.emplace_back(SRT_NODE_INDEX(l), ref16(r));  // the left subtree by the right child
    } else {
      out[i] = (int32_t)(after << 16 | (r != l ? ref16(r) : after));  // first object -> second object (or on)
    }
  }
}

// Hybrid records: (bmin.xyz, reference taken on a box hit) (bmax.xyz, link), node references = NEW indices, link =
// successor << 2 | what follows a leaf's FIRST object: 0 nothing (a single-object leaf), 1 the next primitive of the same
// array (reference - 2: srtUploadScene numbers a leaf's triangles consecutively), 2 primSecond[~first] (any other pair).
// Successor = a node index or -2^29 ("done").  Renumbering: the `cap` boxes of largest surface area reachable from the
// roots come first (greedy expansion; a child's box lies inside its parent's, so the set is closed upward: a walk leaves
// it once per excursion and comes back through a successor), both groups in the trees' pre-order (a node's first child
// is the next record of its group where both are on the same side).  Returns the number of resident nodes, 0 when the
// world is not a forest of two-node / two-primitive trees (outputs empty then).
inline int32_t srtHybridRecords(const std::vector<float4>& nodes, const std::vector<int32_t>& world, int64_t numTriangles, int64_t numSpheres, size_t cap,
                                std::vector<float4>& nodesWf, std::vector<int32_t>& worldWf, std::vector<int32_t>& primSecond) {
  using namespace srt_thread_detail;
  nodesWf.clear();
  worldWf.clear();
  primSecond.clear();
  const size_t n = nodes.size() / 2;
  const int32_t kDoneW = -(1 << 29);
  // (a record's byte offset, index * 32, is a 32-bit buffer offset in the kernel: 2^26 nodes at most)
  if (n == 0 || cap == 0 || n >= ((size_t)1 << 26) || numTriangles >= (1 << 27) || numSpheres >= (1 << 27) || !uniformChildren(nodes)) return 0;
  // successor of every node (original indices, -1 = done), pre-order of the world's trees
  std::vector<int32_t> succ(n, -2), preorder;
  preorder.reserve(n);
  {
    std::vector<std::pair<int32_t, int32_t>> todo;
    for (auto it = world.rbegin(); it != world.rend(); ++it)
      if (*it >= 0) todo.emplace_back(SRT_NODE_INDEX(*it), -1);
    while (!todo.empty()) {
      const auto [i, after] = todo.back();
      todo.pop_back();
      if (i < 0 || (size_t)i >= n || succ[i] != -2) return 0;  // a node reached twice is not a tree
      succ[i] = after;
      preorder.push_back(i);
      const int32_t l = refOf(nodes, 2 * (size_t)i), r = refOf(nodes, 2 * (size_t)i + 1);
      if (l >= 0) {
        todo.emplace_back(SRT_NODE_INDEX(r), after);              // the right subtree is followed by this node's successor
        todo.emplace_back(SRT_NODE_INDEX(l), SRT_NODE_INDEX(r));  // the left subtree by the right child
      }
    }
  }
  if (preorder.empty()) return 0;
  auto area = [&](size_t i) {
    const float4 &lo = nodes[2 * i], &hi = nodes[2 * i + 1];
    const double x = (double)hi.x - lo.x, y = (double)hi.y - lo.y, z = (double)hi.z - lo.z;
    const double s2 = x * y + y * z + z * x;
    return s2 == s2 ? s2 : 1e300;  // a NaN box is visited like any other: keep it near the top
  };
  std::vector<uint8_t> resident(n, 0);
  std::priority_queue<std::pair<double, int32_t>> open;
  for (int32_t wr : world)
    if (wr >= 0) open.emplace(area((size_t)SRT_NODE_INDEX(wr)), -SRT_NODE_INDEX(wr));  // ties: the lower index first
  size_t k = 0;
  while (k < cap && !open.empty()) {
    const int32_t i = -open.top().second;
    open.pop();
    resident[i] = 1;
    ++k;
    const int32_t l = refOf(nodes, 2 * (size_t)i), r = refOf(nodes, 2 * (size_t)i + 1);
    if (l >= 0) {
      open.emplace(area((size_t)SRT_NODE_INDEX(l)), -SRT_NODE_INDEX(l));
      open.emplace(area((size_t)SRT_NODE_INDEX(r)), -SRT_NODE_INDEX(r));
    }
  }
  std::vector<int32_t> newIndex(n, -1);
  int32_t nextRes = 0, nextGlob = (int32_t)k;
  for (int32_t i : preorder) newIndex[i] = resident[i] ? nextRes++ : nextGlob++;
  for (size_t i = 0; i < n; ++i)
    if (newIndex[i] < 0) newIndex[i] = nextGlob++;  // not part of any tree of the world list: never visited
  primSecond.assign((size_t)2 * (size_t)std::max<int64_t>(numTriangles, numSpheres) + 2, kDoneW);
  nodesWf.resize(2 * n);
  for (size_t i = 0; i < n; ++i) {
    float4 lo = nodes[2 * i], hi = nodes[2 * i + 1];
    const int32_t l = refOf(nodes, 2 * i), r = refOf(nodes, 2 * i + 1);
    const int32_t after = succ[i] >= 0 ? newIndex[succ[i]] : kDoneW;  // (-2, an unreachable node: never read)
    int32_t taken = l, follows = 0;
    if (l >= 0) {
      taken = newIndex[SRT_NODE_INDEX(l)];
    } else if (r != l) {
      if (r == l - 2) {
        follows = 1;
      } else if ((size_t)~l < primSecond.size()) {
        follows = 2;
        primSecond[(size_t)~l] = r;
      } else {  // a reference beyond the primitive arrays (validateScene rejects these before we get here)
        nodesWf.clear();
        primSecond.clear();
        return 0;
      }
    }
    const int32_t link = (int32_t)((uint32_t)after << 2) | follows;
    memcpy(&lo.w, &taken, 4);
    memcpy(&hi.w, &link, 4);
    nodesWf[2 * (size_t)newIndex[i]] = lo;
    nodesWf[2 * (size_t)newIndex[i] + 1] = hi;
  }
  for (int32_t wr : world) worldWf.push_back(wr >= 0 ? newIndex[SRT_NODE_INDEX(wr)] : wr);
  return (int32_t)k;
}
