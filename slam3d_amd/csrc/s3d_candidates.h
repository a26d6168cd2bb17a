// s3d_candidates.h - host-side candidate generation for loop-closure sweeps (include/slam3d_hip.h,
// s3d_link_candidates): what ScanSensor::linkToNeighbors (slam3d/core/ScanSensor.cpp:170-202) reads from the graph
// before it registers anything.  No device code; included by s3d_api.hip so that it ships in the same library.
#pragma once
#include <cfloat>
#include <cmath>
#include <queue>
#include <utility>
#include <vector>

namespace s3d_host {

// BoostGraph::calculateGraphDistance (graph/boost/BoostGraph.cpp:301-324): boost::dijkstra_shortest_paths over every
// stored out-edge, float distances, 1 for an SE(3) edge and 10000 for any other
inline float graph_distance(int n, const std::vector<std::vector<std::pair<int, float>>>& out, int source, int target) {
  std::vector<float> dist((size_t)n, FLT_MAX);
  typedef std::pair<float, int> Item;
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
  dist[(size_t)source] = 0.f;
  heap.push(Item(0.f, source));
  while (!heap.empty()) {
    const Item it = heap.top();
    heap.pop();
    if (it.first > dist[(size_t)it.second]) continue;
    if (it.second == target) break;
    for (const auto& e : out[(size_t)it.second]) {
      const float nd = it.first + e.second;
      if (nd < dist[(size_t)e.first]) { dist[(size_t)e.first] = nd; heap.push(Item(nd, e.first)); }
    }
  }
  return dist[(size_t)target];
}

}  // namespace s3d_host

extern "C" int s3d_link_candidates(int n_vertices, const double* positions, const unsigned char* linkable, int n_edges,
                                   const s3d_graph_edge* edges, int vertex, const s3d_link_policy* policy,
                                   int* out_sources, int capacity, int* n_out) try {
  if (n_out) *n_out = 0;
  if (n_vertices < 0 || !positions || n_edges < 0 || (n_edges > 0 && !edges) || !policy || !n_out || vertex < 0 ||
      vertex >= n_vertices || capacity < 0 || (capacity > 0 && !out_sources))
    return S3D_STATUS_INVALID_ARGUMENT;
  for (int e = 0; e < n_edges; ++e)
    if (edges[e].source < 0 || edges[e].source >= n_vertices || edges[e].target < 0 || edges[e].target >= n_vertices)
      return S3D_STATUS_INVALID_ARGUMENT;
  if (policy->max_neighbor_links == 0) return S3D_STATUS_OK;          // ScanSensor.cpp:172-173
  // Graph::getNearbyVertices (Graph.cpp:240-261)
  const double* p0 = positions + 3 * (size_t)vertex;
  std::vector<int> neighbors;
  for (int v = 0; v < n_vertices; ++v) {
    if (linkable && !linkable[v]) continue;
    const double dx = positions[3 * (size_t)v] - p0[0], dy = positions[3 * (size_t)v + 1] - p0[1],
                 dz = positions[3 * (size_t)v + 2] - p0[2];
    const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (d < (double)policy->neighbor_radius) neighbors.push_back(v);
  }
  std::vector<std::vector<std::pair<int, float>>> out((size_t)n_vertices);
  for (int e = 0; e < n_edges; ++e)
    out[(size_t)edges[e].source].push_back(std::make_pair(edges[e].target, edges[e].se3 ? 1.0f : 10000.0f));
  int count = 0, written = 0;
  for (auto it = neighbors.rbegin(); it != neighbors.rend() && count < policy->max_neighbor_links; ++it) {
    const int index = *it;
    if (index == vertex) continue;
    bool linked = false;                                                // getEdge(vertex, index, mName)
    for (int e = 0; e < n_edges && !linked; ++e)
      linked = edges[e].source == vertex && edges[e].target == index && edges[e].own_sensor != 0;
    if (linked) continue;
    const float dist = s3d_host::graph_distance(n_vertices, out, index, vertex);
    if (dist <= (float)(policy->patch_building_range * 2u) || dist < (float)policy->min_loop_length) continue;
    ++count;
    if (written < capacity) out_sources[written] = index;
    ++written;
    // link(index, vertex) changes the graph before the next neighbour is examined (ScanSensor.cpp:137-166): a
    // tentative edge while it registers, and on success an SE(3) edge index <-> vertex of weight 1 in its place - the
    // next neighbour of the same cluster is then only a hop or two from `vertex` and fails min_loop_length.  The
    // candidates of a sweep are listed BEFORE anything is registered: every accepted link is assumed to succeed (a
    // NoMatch would have removed the tentative edge again and let a later neighbour of that cluster through).
    if (!policy->static_graph) {
      out[(size_t)index].push_back(std::make_pair(vertex, 1.0f));
      out[(size_t)vertex].push_back(std::make_pair(index, 1.0f));
    }
  }
  *n_out = written;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr); }
