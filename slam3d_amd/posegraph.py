"""Candidate generation for loop-closure sweeps.

The slice of slam3d::Graph / BoostGraph that ScanSensor's link policy READS, restated so that the batched
registration back-end can be driven from a pose graph instead of a hand-made pair list:

  * Graph::getNearbyVertices   (slam3d/core/Graph.cpp:240-261)        linear scan, d < radius
  * Graph::getTransform        (Graph.cpp:220-223)                    inv(pose_s) * pose_t
  * BoostGraph::addEdge        (graph/boost/BoostGraph.cpp:74-94)     every edge is stored in both directions
  * BoostGraph::getEdge        (:156-176)                             out-edge source -> target of one sensor
  * BoostGraph::getVerticesInRange (:274-299)                         BFS over SE(3) edges, depth <= range
  * BoostGraph::calculateGraphDistance (:301-324)                     Dijkstra, SE(3) edge = 1, any other = 10000
  * ScanSensor::linkToNeighbors (core/ScanSensor.cpp:170-202)         the candidate policy itself
  * ScanSensor::link / buildPatch (:137-166, :215-270)                patch around both ends, createConstraint(loop)

It is host bookkeeping (numpy / heapq), deliberately small: no solver, no optimisation, no serialisation.  The
registrations it produces go through Context.accumulate / align_batch on device-resident clouds.
"""
import heapq
from collections import deque

import numpy as np

from . import api

SE3, TENTATIVE = "SE(3)", "Tentative"


class PoseGraph:
    def __init__(self):
        self.ids = []            # vertex ids in insertion order (= boost vecS vertex descriptors)
        self.pose = {}           # id -> 4x4 correctedPose
        self.sensor = {}         # id -> sensor name
        self.cloud = {}          # id -> api.Cloud (the vertex' measurement, device-resident) or None
        self.sensor_pose = {}    # id -> 4x4 sensor pose of the measurement
        self.out = {}            # id -> list of (target id, sensor, type)

    def add_vertex(self, vid, pose, sensor="velodyne", cloud=None, sensor_pose=None):
        self.ids.append(vid)
        self.pose[vid] = np.asarray(pose, np.float64)
        self.sensor[vid] = sensor
        self.cloud[vid] = cloud
        self.sensor_pose[vid] = np.eye(4) if sensor_pose is None else np.asarray(sensor_pose, np.float64)
        self.out[vid] = []

    def add_edge(self, source, target, sensor="velodyne", etype=SE3):
        """BoostGraph::addEdge: forward and inverse edge carry the same object."""
        self.out[source].append((target, sensor, etype))
        self.out[target].append((source, sensor, etype))

    def remove_edge(self, source, target, sensor):
        """BoostGraph::removeEdge removes the out-edge source -> target of that sensor only (:96-99)."""
        for k, (t, s, _) in enumerate(self.out[source]):
            if t == target and s == sensor:
                del self.out[source][k]
                return
        raise KeyError((source, target, sensor))

    def has_edge(self, source, target, sensor):
        return any(t == target and s == sensor for t, s, _ in self.out[source])

    def get_transform(self, source, target):
        return np.linalg.inv(self.pose[source]) @ self.pose[target]

    def get_nearby_vertices(self, tf, radius, sensors=()):
        t0 = np.asarray(tf, np.float64)[:3, 3]
        res = []
        for v in self.ids:
            if sensors and self.sensor[v] not in sensors:
                continue
            if np.linalg.norm(self.pose[v][:3, 3] - t0) < radius:
                res.append(v)
        return res

    def get_vertices_in_range(self, source, rng):
        """BFS over SE(3) edges; MaxDepthVisitor stops at the first tree edge that leaves depth `rng`.
        Result in vertex-descriptor order (std::map<Vertex, unsigned>)."""
        depth = {source: 0}
        q = deque([source])
        while q:
            u = q.popleft()
            stop = False
            for t, _, ty in self.out[u]:
                if ty != SE3 or t in depth:
                    continue
                if depth[u] >= rng:      # tree_edge(u, t): throw 0
                    stop = True
                    break
                depth[t] = depth[u] + 1
                q.append(t)
            if stop:
                break
        order = {v: k for k, v in enumerate(self.ids)}
        return sorted(depth, key=order.get)

    def calculate_graph_distance(self, source, target):
        dist = {source: 0.0}
        heap = [(0.0, self.ids.index(source), source)]
        done = set()
        while heap:
            d, _, u = heapq.heappop(heap)
            if u in done:
                continue
            done.add(u)
            if u == target:
                return d
            for t, _, ty in self.out[u]:
                nd = d + (1.0 if ty == SE3 else 10000.0)
                if nd < dist.get(t, np.inf):
                    dist[t] = nd
                    heapq.heappush(heap, (nd, self.ids.index(t), t))
        return float(np.finfo(np.float32).max)   # boost leaves unreachable vertices at the distance type's max


class LinkPolicy:
    """ScanSensor's scalar knobs with the constructor defaults (ScanSensor.cpp:34-41)."""

    def __init__(self, name="velodyne", neighbor_radius=1.0, max_neighbor_links=1, min_loop_length=10,
                 patch_building_range=0, link_sensors=None, static_graph=False):
        self.name = name
        self.neighbor_radius = neighbor_radius
        self.max_neighbor_links = max_neighbor_links
        self.min_loop_length = min_loop_length
        self.patch_building_range = patch_building_range
        self.link_sensors = set(link_sensors) if link_sensors is not None else {name}
        self.static_graph = static_graph      # see link_candidates


def link_candidates(graph, vertex, policy):
    """ScanSensor::linkToNeighbors (ScanSensor.cpp:170-202) up to, not including, the registration:
    the (source, target) pairs it would hand to link(), in its order.

    link() changes the graph before the next neighbour is examined (ScanSensor.cpp:143, :157-158: a tentative edge,
    replaced by an SE(3) edge when the registration succeeds), so with max_neighbor_links > 1 the next neighbour of the
    same cluster is only a hop or two from `vertex` and fails min_loop_length.  The candidates of a sweep are listed
    before anything is registered: every accepted link is assumed to succeed (policy.static_graph = False) - a working
    copy of the adjacency gets its SE(3) edge.  policy.static_graph = True judges every neighbour on the graph as
    given (more candidates than the reference links)."""
    out = []
    saved = {}
    if policy.max_neighbor_links == 0:
        return out
    neighbors = graph.get_nearby_vertices(graph.pose[vertex], policy.neighbor_radius, policy.link_sensors)
    count = 0
    try:
        for index in reversed(neighbors):
            if count >= policy.max_neighbor_links:
                break
            if index == vertex:
                continue
            if graph.has_edge(vertex, index, policy.name):       # getEdge(vertex, index, mName) did not throw
                continue
            dist = graph.calculate_graph_distance(index, vertex)
            if dist <= policy.patch_building_range * 2 or dist < policy.min_loop_length:
                continue
            count += 1
            out.append((index, vertex))                          # link(index, vertex)
            if not policy.static_graph:
                for v in (index, vertex):
                    saved.setdefault(v, list(graph.out[v]))
                graph.add_edge(index, vertex, policy.name, SE3)
    finally:
        for v, edges in saved.items():                           # (the caller's graph is left as it was, also when a
            graph.out[v] = edges                                 # query above raises, e.g. KeyError on a bad vertex id)
    return out


def sweep_candidates(graph, policy, vertices=None):
    """Candidates of a whole-graph sweep (every vertex as the 'last' one), duplicates in either direction removed."""
    seen, out = set(), []
    for v in (graph.ids if vertices is None else vertices):
        for s, t in link_candidates(graph, v, policy):
            if (s, t) in seen or (t, s) in seen:
                continue
            seen.add((s, t))
            out.append((s, t))
    return out


def build_patch(ctx, graph, source, policy):
    """ScanSensor::buildPatch (:215-270) without a patch solver: the vertex' own cloud for range 0, otherwise the
    clouds of the vertices within `patch_building_range` hops, accumulated on the device in the source's frame.
    Returns (Cloud, sensor_pose)."""
    if policy.patch_building_range == 0:
        return graph.cloud[source], graph.sensor_pose[source]
    vs = graph.get_vertices_in_range(source, policy.patch_building_range)
    poses = [graph.pose[v] @ graph.sensor_pose[v] for v in vs]            # PointCloudSensor.cpp:249
    patch = ctx.accumulate([graph.cloud[v] for v in vs], poses, graph.pose[source])
    return patch, np.eye(4)


def register_links(ctx, graph, pairs, policy, fine=None, coarse=None, covariance_scale=1.0, opts=None):
    """ScanSensor::link for a LIST of (source, target) pairs, batched: patches on the device, one coarse and one
    fine align_batch (createConstraint with loop = true, PointCloudSensor.cpp:269-299).
    Returns (records (n, 16) float64 with the SE(3) edge source -> target, statuses list)."""
    fine = fine or api.default_params()
    coarse = coarse or api.default_params()
    n = len(pairs)
    rec = np.zeros((n, api.EDGE_RECORD_DOUBLES))
    if n == 0:
        return rec, []
    patches = {}
    for v in {v for p in pairs for v in p}:
        patches[v] = build_patch(ctx, graph, v, policy)
    src = [patches[s][0] for s, _ in pairs]
    tgt = [patches[t][0] for _, t in pairs]
    guesses = []
    for s, t in pairs:   # guess = inv(source sensor pose) * getTransform(source, target) * target sensor pose (:274)
        guesses.append(np.linalg.inv(patches[s][1]) @ graph.get_transform(s, t) @ patches[t][1])
    guesses = np.stack(guesses)
    c = ctx.align_batch(src, tgt, guesses, coarse, opts)                  # :286-289
    ok = np.flatnonzero(c[:, 15] == 0)
    status = c[:, 15].astype(int).tolist()
    if len(ok):
        g2 = np.stack([api.record_transform(c[i]) for i in ok])
        f = ctx.align_batch([src[i] for i in ok], [tgt[i] for i in ok], g2, fine, opts)   # :292
        for k, i in enumerate(ok):
            rec[i] = f[k]
            status[i] = int(f[k, 15])
            if status[i] == 0:   # :295 sensorPose_s * icp * inv(sensorPose_t)
                T = patches[pairs[i][0]][1] @ api.record_transform(f[k]) @ np.linalg.inv(patches[pairs[i][1]][1])
                rec[i, :12] = T[:3, :4].T.reshape(12)
    for i in range(n):
        rec[i, 15] = status[i]
    return rec, status
