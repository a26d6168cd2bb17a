"""Candidate generation: slam3d_amd/posegraph.py against independent scipy restatements of
the graph queries and against the rules of ScanSensor::linkToNeighbors read off the reference source.  CPU only."""
import numpy as np
import pytest
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import dijkstra

from slam3d_amd.posegraph import SE3, TENTATIVE, LinkPolicy, PoseGraph, link_candidates, sweep_candidates


def pose(x, y, yaw=0.0):
    T = np.eye(4)
    T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    T[:2, 3] = [x, y]
    return T


def loop_graph(n=40, radius=10.0, extra=()):
    """n vertices on a circle of `radius`, odometry edges between consecutive ones, the loop NOT closed."""
    g = PoseGraph()
    for i in range(n):
        a = 2 * np.pi * i / n
        g.add_vertex(i, pose(radius * np.cos(a), radius * np.sin(a), a + np.pi / 2))
    for i in range(n - 1):
        g.add_edge(i, i + 1, "velodyne", SE3)
    for s, t, sensor, ty in extra:
        g.add_edge(s, t, sensor, ty)
    return g


def scipy_dist(g, weights=(1.0, 10000.0)):
    idx = {v: k for k, v in enumerate(g.ids)}
    rows, cols, w = [], [], []
    for u in g.ids:
        for t, _, ty in g.out[u]:
            rows.append(idx[u]); cols.append(idx[t]); w.append(weights[0] if ty == SE3 else weights[1])
    m = csr_matrix((w, (rows, cols)), shape=(len(g.ids),) * 2)
    return dijkstra(m, directed=True), idx


def test_nearby_vertices_is_the_linear_scan():
    g = loop_graph(40)
    g.add_vertex(100, pose(10.0, 0.3), sensor="other")
    tf = pose(10.0, 0.0)
    got = g.get_nearby_vertices(tf, 3.5, {"velodyne"})
    want = [v for v in g.ids if g.sensor[v] == "velodyne" and np.linalg.norm(g.pose[v][:3, 3] - tf[:3, 3]) < 3.5]
    assert got == want and 0 in got and 100 not in got
    assert 100 in g.get_nearby_vertices(tf, 3.5, ())           # empty sensor set = all sensors (BoostGraph.cpp:107)
    assert g.get_nearby_vertices(tf, 0.0) == []                 # strict <


def test_graph_distance_is_dijkstra_with_the_reference_weights():
    g = loop_graph(30, extra=[(2, 20, "velodyne", TENTATIVE), (5, 12, "gps", SE3)])
    d, idx = scipy_dist(g)
    for a, b in [(0, 29), (3, 19), (5, 12), (4, 13), (29, 0), (7, 7)]:
        assert g.calculate_graph_distance(a, b) == d[idx[a], idx[b]]
    assert g.calculate_graph_distance(2, 20) == 12.0           # 3 + 1 (the SE(3) edge of another sensor) + 8 hops beat the 10000-weight tentative edge
    g2 = PoseGraph()
    g2.add_vertex(0, np.eye(4)); g2.add_vertex(1, np.eye(4))
    assert g2.calculate_graph_distance(0, 1) > 1e30            # unreachable


@pytest.mark.parametrize("rng", [0, 1, 2, 5])
def test_vertices_in_range_is_bfs_depth(rng):
    g = loop_graph(30, extra=[(2, 20, "velodyne", TENTATIVE), (5, 12, "gps", SE3)])
    d, idx = scipy_dist(g, weights=(1.0, np.inf))               # SE(3) edges only, hop count
    for src in (0, 5, 12, 29):
        want = [v for v in g.ids if d[idx[src], idx[v]] <= rng]
        assert g.get_vertices_in_range(src, rng) == want


def test_link_candidates_follow_the_reference_policy():
    n = 40
    g = loop_graph(n)
    pol = LinkPolicy(neighbor_radius=3.5, max_neighbor_links=1, min_loop_length=10)
    # the last vertex sees the start of the loop again: one candidate, the LAST nearby vertex in graph order that
    # passes the gates (reverse iteration), as source, the queried vertex as target
    c = link_candidates(g, n - 1, pol)
    near = g.get_nearby_vertices(g.pose[n - 1], 3.5, {"velodyne"})
    passing = [v for v in near if v != n - 1 and g.calculate_graph_distance(v, n - 1) >= 10]
    assert c == [(passing[-1], n - 1)]
    # more links allowed: reverse graph order, capped
    pol3 = LinkPolicy(neighbor_radius=3.5, max_neighbor_links=3, min_loop_length=10, static_graph=True)
    assert link_candidates(g, n - 1, pol3) == [(v, n - 1) for v in reversed(passing)][:3]
    # max_neighbor_links = 0 -> nothing; a vertex in the middle of the open loop has only close-by neighbours
    assert link_candidates(g, n - 1, LinkPolicy(neighbor_radius=3.5, max_neighbor_links=0)) == []
    assert link_candidates(g, 20, pol3) == []
    # an existing edge vertex -> index of this sensor suppresses the candidate (another sensor's does not)
    g.add_edge(n - 1, passing[-1], "velodyne", SE3)
    assert passing[-1] not in [s for s, _ in link_candidates(g, n - 1, pol3)]
    g2 = loop_graph(n, extra=[(n - 1, 0, "gps", SE3)])
    # ... but that gps edge shortens the graph distance to 1 hop: below min_loop_length, no candidate at all
    assert link_candidates(g2, n - 1, pol) == []
    # patch_building_range: dist <= 2 * range is excluded even when min_loop_length would allow it
    polp = LinkPolicy(neighbor_radius=3.5, max_neighbor_links=5, min_loop_length=0, patch_building_range=1)
    for s, t in link_candidates(g, 20, polp):
        assert g.calculate_graph_distance(s, t) > 2


def _reference_link_to_neighbors(g, vertex, pol):
    """A model of ScanSensor::linkToNeighbors (ScanSensor.cpp:170-202) that MUTATES the graph as link() does
    (:137-166) when every registration succeeds: tentative edge in both directions, removed again (removeConstraint
    drops both directions, Graph.cpp), SE(3) edge in both directions - before the next neighbour is examined."""
    calls = []
    if pol.max_neighbor_links == 0:
        return calls
    count = 0
    for index in reversed(g.get_nearby_vertices(g.pose[vertex], pol.neighbor_radius, pol.link_sensors)):
        if count >= pol.max_neighbor_links:
            break
        if index == vertex or g.has_edge(vertex, index, pol.name):
            continue
        dist = g.calculate_graph_distance(index, vertex)
        if dist <= pol.patch_building_range * 2 or dist < pol.min_loop_length:
            continue
        count += 1
        calls.append((index, vertex))
        g.add_edge(index, vertex, pol.name, TENTATIVE)       # addTentativeConstraint
        g.remove_edge(index, vertex, pol.name)               # removeConstraint
        g.remove_edge(vertex, index, pol.name)
        g.add_edge(index, vertex, pol.name, SE3)             # addConstraint(se3)
    return calls


def test_link_candidates_with_several_links_follow_the_mutating_reference():
    """max_neighbor_links > 1 (round-3 advisor finding): link() inserts its SE(3) edge before the next neighbour is
    examined, so the neighbours of one spatial cluster are a hop or two from the vertex after the first link and fail
    min_loop_length - about one link per cluster, not max_neighbor_links of them.  Python restatement and C ABI
    against a model that mutates a copy of the graph; the caller's graph is left untouched."""
    import copy
    n = 40
    g = loop_graph(n)
    pol = LinkPolicy(neighbor_radius=3.5, max_neighbor_links=3, min_loop_length=10)
    before = copy.deepcopy(g.out)
    want = _reference_link_to_neighbors(copy.deepcopy(g), n - 1, pol)
    got = link_candidates(g, n - 1, pol)
    assert got == want and g.out == before
    static = link_candidates(g, n - 1, LinkPolicy(neighbor_radius=3.5, max_neighbor_links=3, min_loop_length=10,
                                                   static_graph=True))
    assert len(want) == 1 and len(static) > 1 and static[0] == want[0]      # one cluster at the loop's start: one link
    assert _c_abi_candidates(g, n - 1, pol) == want
    rng = np.random.default_rng(9)
    for trial in range(30):
        m = int(rng.integers(10, 70))
        r = PoseGraph()
        for i in range(m):
            r.add_vertex(i, pose(*rng.uniform(-5, 5, 2)))
        for i in range(m - 1):
            r.add_edge(i, i + 1, "velodyne", SE3)
        p2 = LinkPolicy(neighbor_radius=float(rng.uniform(2, 6)), max_neighbor_links=int(rng.integers(2, 8)),
                        min_loop_length=int(rng.integers(2, 12)), patch_building_range=int(rng.integers(0, 2)))
        for v in rng.integers(0, m, 5):
            want = _reference_link_to_neighbors(copy.deepcopy(r), int(v), p2)
            assert link_candidates(r, int(v), p2) == want, (trial, int(v))
            assert _c_abi_candidates(r, int(v), p2) == want, (trial, int(v))


def test_sweep_candidates_deduplicates_both_directions():
    g = loop_graph(40)
    pol = LinkPolicy(neighbor_radius=3.5, max_neighbor_links=2, min_loop_length=10)
    c = sweep_candidates(g, pol)
    assert len(c) == len(set(c)) and all((t, s) not in c for s, t in c) and len(c) > 0
    for s, t in c:
        assert np.linalg.norm(g.pose[s][:3, 3] - g.pose[t][:3, 3]) < 3.5
        assert g.calculate_graph_distance(s, t) >= 10


def _c_abi_candidates(g, vertex, pol):
    """The same query through the C ABI (s3d_link_candidates, include/slam3d_hip.h): vertices by insertion index."""
    from slam3d_amd import api
    idx = {v: k for k, v in enumerate(g.ids)}
    pos = np.array([g.pose[v][:3, 3] for v in g.ids])
    edges = [(idx[u], idx[t], int(ty == SE3), int(s == pol.name)) for u in g.ids for t, s, ty in g.out[u]]
    linkable = [int(g.sensor[v] in pol.link_sensors) for v in g.ids]
    src = api.link_candidates(pos, edges, idx[vertex], pol.neighbor_radius, pol.max_neighbor_links, pol.min_loop_length,
                              pol.patch_building_range, linkable, pol.static_graph)
    return [(g.ids[s], vertex) for s in src]


def test_c_abi_link_candidates_equal_the_python_restatement():
    """s3d_link_candidates (C ABI, host code of libslam3d_hip.so, no GPU involved) against posegraph.link_candidates on
    the loop graph with other-sensor / tentative / removed edges and on random graphs: same candidates, same order."""
    n = 40
    g = loop_graph(n, 10.0, extra=[(3, 30, "gps", "GPS"), (5, 25, "velodyne", TENTATIVE), (12, 39, "velodyne", SE3)])
    g.remove_edge(39, 12, "velodyne")          # removeEdge drops ONE direction: 12 -> 39 is still stored
    g.add_vertex(100, pose(9.5, 0.5), sensor="camera")
    for pol in (LinkPolicy(neighbor_radius=3.5, max_neighbor_links=5, min_loop_length=10),
                LinkPolicy(neighbor_radius=25.0, max_neighbor_links=100, min_loop_length=4, patch_building_range=3),
                LinkPolicy(neighbor_radius=6.0, max_neighbor_links=2, min_loop_length=0),
                LinkPolicy(neighbor_radius=6.0, max_neighbor_links=0),
                LinkPolicy(neighbor_radius=8.0, max_neighbor_links=9, link_sensors={"velodyne", "camera"})):
        for v in (39, 20, 12, 0, 100):
            assert _c_abi_candidates(g, v, pol) == link_candidates(g, v, pol), (v, vars(pol))
    rng = np.random.default_rng(3)
    for trial in range(20):
        m = int(rng.integers(5, 60))
        r = PoseGraph()
        for i in range(m):
            r.add_vertex(i, pose(*rng.uniform(-6, 6, 2)), sensor=["velodyne", "cam"][int(rng.integers(0, 4) == 0)])
        for i in range(m - 1):
            if rng.random() < 0.9:
                r.add_edge(i, i + 1, "velodyne", SE3)
        for _ in range(m // 3):
            a, b = rng.integers(0, m, 2)
            if a != b:
                r.add_edge(int(a), int(b), ["velodyne", "gps"][int(rng.integers(0, 2))], [SE3, "GPS", TENTATIVE][int(rng.integers(0, 3))])
        pol = LinkPolicy(neighbor_radius=float(rng.uniform(1, 8)), max_neighbor_links=int(rng.integers(1, 6)),
                         min_loop_length=int(rng.integers(0, 8)), patch_building_range=int(rng.integers(0, 3)),
                         static_graph=bool(trial % 2))
        for v in rng.integers(0, m, 6):
            assert _c_abi_candidates(r, int(v), pol) == link_candidates(r, int(v), pol), (trial, int(v))
    with pytest.raises(ValueError):
        from slam3d_amd import api
        api.link_candidates(np.zeros((3, 3)), [(0, 7, 1, 1)], 0)
