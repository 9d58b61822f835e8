#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace ptw {
namespace {

struct Box {
  double lo[3], hi[3];
  Box() {
    for (int a = 0; a < 3; ++a) lo[a] = std::numeric_limits<double>::infinity(), hi[a] = -lo[a];
  }
  void grow(const double p[3]) {
    for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], p[a]), hi[a] = std::max(hi[a], p[a]);
  }
  void grow(const Box &b) {
    for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], b.lo[a]), hi[a] = std::max(hi[a], b.hi[a]);
  }
  double area() const {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx < 0 ? 0.0 : 2 * (dx * dy + dy * dz + dz * dx);
  }
};

struct Builder {
  const double *geom;
  std::vector<Box> triBox;
  std::vector<double> centroid; // [ntri][3]
  std::vector<uint32_t> order;  // permutation being partitioned
  double margin = 0;
  Bvh out;

  Box boxOf(uint32_t first, uint32_t count) const {
    Box b;
    for (uint32_t i = first; i < first + count; ++i) b.grow(triBox[order[i]]);
    return b;
  }

  // Describes [first, first + count) as a child slot: a leaf, or a freshly built inner node.
  void makeChild(BvhNode &parent, int slot, uint32_t first, uint32_t count, int depth) {
    const Box b = boxOf(first, count);
    for (int a = 0; a < 3; ++a) parent.lo[slot][a] = b.lo[a] - margin, parent.hi[slot][a] = b.hi[a] + margin;
    out.depth = std::max(out.depth, depth);
    if (count <= static_cast<uint32_t>(kBvhLeafSize) || depth >= kBvhMaxDepth) {
      parent.child[slot] = static_cast<int32_t>(out.leafIndex.size());
      parent.count[slot] = static_cast<int32_t>(count);
      // ascending insertion index inside a leaf (the order is irrelevant for the result - the
      // device compares (t, index) - it only keeps the build deterministic)
      std::sort(order.begin() + first, order.begin() + first + count);
      for (uint32_t i = first; i < first + count; ++i) {
        out.leafIndex.push_back(order[i]);
        out.leafGeom.insert(out.leafGeom.end(), geom + 9 * static_cast<size_t>(order[i]),
                            geom + 9 * static_cast<size_t>(order[i]) + 9);
      }
      return;
    }
    const int32_t me = static_cast<int32_t>(out.nodes.size());
    out.nodes.emplace_back();
    parent.child[slot] = me;
    parent.count[slot] = 0;
    split(me, first, count, depth);
  }

  // Binned surface-area heuristic over the centroids; falls back to a median split.
  void split(int32_t node, uint32_t first, uint32_t count, int depth) {
    Box cb;
    for (uint32_t i = first; i < first + count; ++i) cb.grow(&centroid[3 * static_cast<size_t>(order[i])]);
    int bestAxis = -1;
    double bestCost = std::numeric_limits<double>::infinity(), bestPos = 0;
    constexpr int kBins = 16;
    for (int axis = 0; axis < 3; ++axis) {
      const double lo = cb.lo[axis], extent = cb.hi[axis] - cb.lo[axis];
      if (!(extent > 0)) continue;
      Box bins[kBins];
      uint32_t binCount[kBins] = {0};
      for (uint32_t i = first; i < first + count; ++i) {
        const uint32_t t = order[i];
        int b = static_cast<int>((centroid[3 * static_cast<size_t>(t) + axis] - lo) / extent * kBins);
        b = std::min(std::max(b, 0), kBins - 1);
        bins[b].grow(triBox[t]);
        binCount[b]++;
      }
      Box left[kBins], right[kBins];
      uint32_t nLeft[kBins], nRight[kBins];
      Box acc;
      uint32_t n = 0;
      for (int b = 0; b < kBins; ++b) acc.grow(bins[b]), n += binCount[b], left[b] = acc, nLeft[b] = n;
      acc = Box(), n = 0;
      for (int b = kBins - 1; b >= 0; --b) acc.grow(bins[b]), n += binCount[b], right[b] = acc, nRight[b] = n;
      for (int b = 0; b + 1 < kBins; ++b) {
        if (nLeft[b] == 0 || nRight[b + 1] == 0) continue;
        const double cost = left[b].area() * nLeft[b] + right[b + 1].area() * nRight[b + 1];
        if (cost < bestCost) bestCost = cost, bestAxis = axis, bestPos = lo + extent * (b + 1) / kBins;
      }
    }
    uint32_t mid;
    if (bestAxis >= 0) {
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
        return centroid[3 * static_cast<size_t>(t) + bestAxis] < bestPos;
      });
      mid = static_cast<uint32_t>(it - order.begin());
    } else {
      mid = first;
    }
    if (mid == first || mid == first + count) { // all centroids equal / a degenerate split: halves
      mid = first + count / 2;
      int axis = 0;
      for (int a = 1; a < 3; ++a)
        if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint32_t x, uint32_t y) {
                         return centroid[3 * static_cast<size_t>(x) + axis] < centroid[3 * static_cast<size_t>(y) + axis];
                       });
    }
    // (the node vector may reallocate while the children are built: address it by index)
    BvhNode scratch;
    std::memset(&scratch, 0, sizeof scratch);
    makeChild(scratch, 0, first, mid - first, depth + 1);
    makeChild(scratch, 1, mid, first + count - mid, depth + 1);
    out.nodes[static_cast<size_t>(node)] = scratch;
  }
};

} // namespace

Bvh buildBvh(const double *triGeom, uint32_t ntri) {
  Builder b;
  b.geom = triGeom;
  if (ntri == 0) return b.out;
  b.triBox.resize(ntri);
  b.centroid.resize(3 * static_cast<size_t>(ntri));
  b.order.resize(ntri);
  std::iota(b.order.begin(), b.order.end(), 0u);
  Box scene;
  for (uint32_t t = 0; t < ntri; ++t) {
    const double *g = triGeom + 9 * static_cast<size_t>(t);
    double v[3][3];
    for (int a = 0; a < 3; ++a) v[0][a] = g[a], v[1][a] = g[a] + g[3 + a], v[2][a] = g[a] + g[6 + a];
    for (int k = 0; k < 3; ++k) b.triBox[t].grow(v[k]);
    for (int a = 0; a < 3; ++a) b.centroid[3 * static_cast<size_t>(t) + a] = (v[0][a] + v[1][a] + v[2][a]) / 3;
    scene.grow(b.triBox[t]);
  }
  double extent = 0;
  for (int a = 0; a < 3; ++a) extent = std::max({extent, std::fabs(scene.lo[a]), std::fabs(scene.hi[a])});
  // The margin dwarfs every rounding error involved (e1 / e2 reconstruct the vertices to ~1e-16
  // relative, the slab test and the hit distance err by ~1e-15 relative) and is still far too
  // small to cost any culling.
  b.margin = 1e-7 * extent + 1e-12;
  b.out.nodes.emplace_back();
  if (ntri <= static_cast<uint32_t>(kBvhLeafSize)) {
    BvhNode root;
    std::memset(&root, 0, sizeof root);
    b.makeChild(root, 0, 0, ntri, 1);
    // second slot: an empty box nothing can enter
    for (int a = 0; a < 3; ++a) root.lo[1][a] = 1.0, root.hi[1][a] = -1.0;
    root.child[1] = 0, root.count[1] = 0;
    root.count[1] = -1; // marks "no child"
    b.out.nodes[0] = root;
  } else {
    b.split(0, 0, ntri, 0);
  }
  return b.out;
}

} // namespace ptw
