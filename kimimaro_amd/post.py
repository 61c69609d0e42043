"""Row f4: stitching the skeletons of adjacent chunks -- the counterpart of kimimaro/post.py.

    postprocess(skeleton, dust_threshold=1500, tick_threshold=3000)     kimimaro/post.py:49-87
    join_close_components(skeletons, radius=inf, restrict_by_radius)    kimimaro/post.py:89-218
    remove_dust / remove_loops / remove_ticks                            kimimaro/post.py:222-260, 436-563

Host code on graphs of 10^2..10^5 nodes: the reference is Python here and so is this; nothing
in it touches the GPU path.  Written from the behaviour of the reference, including the parts that are visible in its
results: which cycle its depth-first search reports first (skeletontricks.hpp:209-300: neighbours in the order the
edges list them, the search starts at the first edge), float32 branch lengths accumulated outward from the smallest
terminal node (skeletontricks.hpp:303-380), superedges fused at a branch point that drops to two edges and from then
on treated as removable whatever their ends are (post.py:324-336).  Where the reference leaves a choice to the
iteration order of a Python set (equal-length ticks, post.py:339) the smallest (length, nodes) pair is taken.
Pinned by tests/golden/post.npz: outputs of the reference's own post.py run on the vectors' inputs (generator and the
stand-ins it needs: tests/golden/make_golden.py `post`).
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np

from .skeleton import Skeleton


def postprocess(skeleton, dust_threshold=1500.0, tick_threshold=3000.0):
    """kimimaro/post.py:49-87: dust components out, loops out, close components joined, ticks out."""
    label = skeleton.id
    skel = skeleton.consolidate(remove_disconnected_vertices=True)
    skel = remove_dust(skel, dust_threshold)
    skel = remove_loops(skel)
    skel = join_close_components(skel, restrict_by_radius=True)
    skel = remove_ticks(skel, tick_threshold)
    skel.id = label
    return skel.consolidate(remove_disconnected_vertices=True)


# ---------------------------------------------------------------------------------------------------------------
def remove_dust(skeleton, dust_threshold):
    """components whose cable length does not exceed the threshold are dropped (post.py:222-233)."""
    if skeleton.empty() or dust_threshold == 0:
        return skeleton
    return Skeleton.simple_merge([c for c in skeleton.components() if c.cable_length() > dust_threshold])


# ---------------------------------------------------------------------------------------------------------------
def join_close_components(skeletons, radius=np.inf, restrict_by_radius=False):
    """Repeatedly connects the two components whose nearest vertices are closest (post.py:89-218).  With
    restrict_by_radius the search radius becomes twice the largest vertex radius and a pair only qualifies when its
    gap is at most the sum of the two vertices' radii."""
    from scipy.spatial import cKDTree
    if radius is None:
        radius = np.inf
    if radius <= 0:
        raise ValueError("radius must be greater than zero: " + str(radius))
    if isinstance(skeletons, Skeleton):
        skeletons = [skeletons]
    parts = []
    for s in skeletons:
        parts.extend(c.consolidate(remove_disconnected_vertices=True) for c in s.components())
    parts = [p for p in parts if not p.empty()]
    if len(parts) == 1:
        return parts[0]
    if not parts:
        return Skeleton()
    if restrict_by_radius:
        radius = max(2 * max(float(np.max(p.radii)) for p in parts), 0)

    def gap(tree, a, b):
        """nearest pair between parts a (in `tree`) and b: (distance as the float32 the reference stores, index in a, in b)"""
        dist, hit = tree.query(b.vertices, k=1, distance_upper_bound=radius + 0.000001)
        kb = int(np.argmin(dist))
        ka = int(hit[kb])
        d = dist[kb]
        if restrict_by_radius and np.isfinite(d) and d > (a.radii[ka] + b.radii[kb]):
            d = np.inf
        return np.float32(d), ka, kb

    # gaps[i][j] for i < j only (the reference fills both triangles of its matrix and reads the first minimum in row-major
    # order, which lies in the upper one)
    n = len(parts)
    gaps = {}
    for i in range(n):
        tree = cKDTree(parts[i].vertices)
        for j in range(i + 1, n):
            gaps[(i, j)] = gap(tree, parts[i], parts[j])
    while len(parts) > 1:
        n = len(parts)
        best = min(((gaps[(i, j)][0], i, j) for i in range(n) for j in range(i + 1, n)), key=lambda t: (t[0], t[1], t[2]))
        if not np.isfinite(best[0]) or best[0] > radius:
            break
        _, i, j = best
        _, ka, kb = gaps[(i, j)]
        a, b = parts[i], parts[j]
        fused = Skeleton.simple_merge([a, b])
        fused.edges = np.concatenate([fused.edges, np.array([[ka, kb + a.vertices.shape[0]]], dtype=np.uint32)])
        rest = [k for k in range(n) if k not in (i, j)]
        renum = {old: new + 1 for new, old in enumerate(rest)}
        gaps = {(renum[p], renum[q]): v for (p, q), v in gaps.items() if p in renum and q in renum}
        parts = [fused] + [parts[k] for k in rest]
        tree = cKDTree(fused.vertices)
        for j in range(1, len(parts)):
            gaps[(0, j)] = gap(tree, fused, parts[j])
    return Skeleton.simple_merge(parts).consolidate(remove_disconnected_vertices=True)


# ---------------------------------------------------------------------------------------------------------------
def find_cycle(edges):
    """The cycle the reference's search reports for this edge list (skeletontricks.hpp:209-300), as a node sequence whose
    first and last entries are the same node; empty when the search meets no visited node.  Depth first from edges[0][0],
    neighbours in first-mention order, a node's whole neighbour list goes on the stack at once."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if edges.shape[0] == 0:
        return []
    nbrs = defaultdict(dict)       # dict = insertion-ordered set
    for a, b in edges.tolist():
        nbrs[a][b] = None
        nbrs[b][a] = None
    todo = [(int(edges[0, 0]), -1, 0)]
    trail = []
    seen = set()
    node = -1
    while todo:
        node, via, depth = todo.pop()
        del trail[depth:]
        trail.append(node)
        if node in seen:
            break
        seen.add(node)
        todo.extend((c, node, depth + 1) for c in nbrs[node] if c != via)
    if len(trail) <= 1:
        return []
    first = next((k for k in range(len(trail) - 1) if trail[k] == node), len(trail) - 1)
    return trail[first:] if len(trail) - first >= 3 else []


def _drop_edges(edges, doomed):
    """edges (rows sorted) without every row that equals a row of `doomed` (post.py:576-588)."""
    edges = np.sort(edges, axis=1)
    if edges.size == 0 or len(doomed) == 0:
        return edges
    doomed = {(int(a), int(b)) for a, b in np.sort(np.asarray(doomed, dtype=np.int64).reshape(-1, 2), axis=1)}
    keep = np.fromiter(((int(a), int(b)) not in doomed for a, b in edges), dtype=bool, count=edges.shape[0])
    return edges[keep]


def _remove_loops_component(skel):
    """post.py:446-563 on one connected component."""
    nodes = skel.vertices
    edges = skel.edges.astype(np.int64)
    while True:
        cyc = find_cycle(edges)
        if not cyc:
            break
        ring = np.sort(np.stack([cyc[:-1], cyc[1:]], axis=1), axis=1)      # the cycle's edges, in walk order
        on_ring = np.unique(ring)
        ids, deg = np.unique(edges, return_counts=True)
        gates = on_ring[np.isin(on_ring, ids[deg >= 3])]                    # cycle nodes where something else attaches
        if gates.size == 0:                      # an isolated ring: gone
            edges = _drop_edges(edges, ring)
        elif gates.size == 1:                    # a ring on a stalk: replaced by a line to its farthest node
            d2 = np.sum((nodes[on_ring] - nodes[gates]) ** 2, axis=1)
            far = int(on_ring[int(np.argmax(d2))])
            edges = np.concatenate([_drop_edges(edges, ring), np.array([[int(gates[0]), far]], dtype=np.int64)])
        elif gates.size == 2:                    # a way in and a way out: the arc with fewer nodes stays
            walk = np.asarray(cyc[1:])
            at = np.flatnonzero(np.isin(walk, gates))
            if (at[1] - at[0]) < len(walk) / 2:
                arc = walk[at[0]:at[1] + 1]
            else:
                arc = np.concatenate([walk[at[1]:], walk[:at[0] + 1]])
            kept = {(int(a), int(b)) for a, b in np.sort(np.stack([arc[:-1], arc[1:]], axis=1), axis=1)}
            edges = _drop_edges(edges, [e for e in ring.tolist() if (e[0], e[1]) not in kept])
        else:                                    # many ways in: collapse onto the vertex nearest the gates' centroid ...
            centroid = np.mean(nodes[gates], axis=0)
            off = nodes - centroid
            hub = int(np.argmin(np.sum(off * off, axis=1)))
            reach = np.sqrt(np.max(np.sum((nodes[gates] - nodes[hub]) ** 2, axis=1)))
            if reach > skel.radii[hub]:          # ... unless that vertex is too thin to be the hub: snip one edge instead
                edges = _drop_edges(edges, ring[:1])
                continue
            spokes = np.array([[int(g), hub] for g in gates if int(g) != hub], dtype=np.int64).reshape(-1, 2)
            edges = np.concatenate([_drop_edges(edges, ring), spokes])
    skel.edges = edges.astype(np.uint32)
    return skel


def remove_loops(skeleton):
    """skeletons are trees: every cycle is removed, by the rule its number of outside connections selects (post.py:436-563)."""
    if skeleton.empty():
        return skeleton
    return Skeleton.simple_merge([_remove_loops_component(c) for c in skeleton.components()]).consolidate(
        remove_disconnected_vertices=False)   # post.py:260


# ---------------------------------------------------------------------------------------------------------------
def distance_graph(skel):
    """{(larger node, smaller node): float32 path length} between the critical points (terminal and branch nodes) of a
    tree, accumulated outward from its smallest terminal node (skeletontricks.pyx:122-170 -> skeletontricks.hpp:303-380)."""
    v = skel.vertices.astype(np.float32)
    edges = skel.edges.astype(np.int64)
    ids, deg = np.unique(edges, return_counts=True)
    critical = set(ids[(deg == 1) | (deg >= 3)].tolist())
    terminals = ids[deg == 1]
    if terminals.size == 0:
        raise ValueError("distance_graph: the component has no terminal node (it contains a cycle)")
    nbrs = defaultdict(list)
    for a, b in edges.tolist():
        nbrs[a].append(b)
        nbrs[b].append(a)
    start = int(terminals[0])
    out = {}
    seen = set()
    todo = [(start, -1, np.float32(0.0), start)]
    while todo:
        node, via, dist, root = todo.pop()
        if node in seen:
            raise ValueError("distance_graph: cycle detected at node %d" % node)
        seen.add(node)
        if node in critical and node != root:
            out[(max(root, node), min(root, node))] = float(dist)
            dist, root = np.float32(0.0), node
        for c in nbrs[node]:
            if c == via:
                continue
            d = v[node] - v[c]
            d = d * d
            step = np.sqrt(np.float32(np.float32(d[0] + d[1]) + d[2]))
            todo.append((c, node, np.float32(dist + step), root))
    return out


def _remove_ticks_component(skel, threshold):
    """post.py:262-362 on one tree."""
    if skel.empty():
        return skel
    span = distance_graph(skel)                     # superedge -> length, in creation order
    ids, deg = np.unique(skel.edges, return_counts=True)
    terminals = set(ids[deg == 1].tolist())
    arity = defaultdict(int)
    for node, d in zip(ids.tolist(), deg.tolist()):
        if d >= 3:
            arity[node] = d
    nbrs = defaultdict(set)
    for a, b in skel.edges.tolist():
        nbrs[a].add(b)
        nbrs[b].add(a)
    removable = {e for e in span if e[0] in terminals or e[1] in terminals}

    def fuse(x):
        """x is down to two superedges: they become one, which is removable from now on (post.py:324-336)"""
        joined = [e for e in span if x in e]
        total = 0.0
        ends = set()
        for e in joined:
            removable.discard(e)
            total += span.pop(e)
            ends.update(e)
        ends.discard(x)
        e = tuple(sorted(ends, reverse=True))
        span[e] = total
        removable.add(e)
        arity[x] = 0

    while len(span) > 1:
        tick = min(removable, key=lambda e: (span[e], e))
        a, b = tick
        if (arity[a] == 1 and arity[b] == 1) or span[tick] >= threshold:
            break
        # the unique path a .. b in what is left of the tree
        back = {a: None}
        todo = [a]
        while todo:
            node = todo.pop()
            if node == b:
                break
            for c in nbrs[node]:
                if c not in back:
                    back[c] = node
                    todo.append(c)
        node = b
        while back[node] is not None:
            nbrs[node].discard(back[node])
            nbrs[back[node]].discard(node)
            node = back[node]
        del span[tick]
        removable.remove(tick)
        arity[a] -= 1
        arity[b] -= 1
        if arity[a] == 2:
            fuse(a)
        if arity[b] == 2:
            fuse(b)
    out = skel.clone()
    left = sorted({(min(a, b), max(a, b)) for a in nbrs for b in nbrs[a]})
    out.edges = np.asarray(left, dtype=np.uint32).reshape(-1, 2)
    return out


def remove_ticks(skeleton, threshold):
    """Terminal branches shorter than `threshold` are removed one at a time, shortest first, the topology being
    re-evaluated after each removal (post.py:235-362)."""
    if skeleton.empty() or threshold == 0:
        return skeleton
    return Skeleton.simple_merge([_remove_ticks_component(c, threshold) for c in skeleton.components()]).consolidate(
        remove_disconnected_vertices=False)   # post.py:444
