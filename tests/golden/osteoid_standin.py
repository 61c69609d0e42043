"""Stand-in for the third-party `osteoid.Skeleton` (absent from the reference tree and from this image), used ONLY by
tests/golden/make_golden.py to run the reference's own kimimaro/post.py in the build container.

It restates the operations post.py calls (kimimaro/post.py:77-86,115-123,186,218,222-233,256-260,440-444 ...) as
osteoid / cloud-volume document them: consolidate (np.unique over the vertex rows, edges renumbered, each edge
sorted, rows sorted and made unique, self loops dropped, optionally vertices without an edge dropped), components
(the components of the consolidated skeleton, ordered by their smallest vertex; a component keeps its vertices in order
and lists its edges as sorted unique rows), simple_merge, clone, empty, cable_length.  What pins this stand-in to the
real package: the reference's own known-answer tests that go through it (automated_test.py:384-456 join_close_components
with exact edge arrays, :611-632 postprocess -- the latter fails if components() lists a cycle's closing edge twice),
mirrored in tests/test_post.py.  Independent of kimimaro_amd and of oracle/.
"""
from collections import defaultdict

import numpy as np


class Skeleton:
    def __init__(self, vertices=None, edges=None, radii=None, vertex_types=None, segid=None, transform=None, space="physical"):
        self.id = segid
        self.space = space
        self.vertices = np.zeros((0, 3), np.float32) if vertices is None else np.array(vertices, dtype=np.float32).reshape(-1, 3)
        self.edges = np.zeros((0, 2), np.uint32) if edges is None else np.array(edges, dtype=np.uint32).reshape(-1, 2)
        n = len(self.vertices)
        self.radii = -np.ones(n, np.float32) if radii is None else np.array(radii, dtype=np.float32)
        self.vertex_types = np.zeros(n, np.uint8) if vertex_types is None else np.array(vertex_types, dtype=np.uint8)
        self.transform = np.hstack([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)]) if transform is None else transform

    @classmethod
    def simple_merge(cls, skeletons):
        skeletons = list(skeletons)
        if len(skeletons) == 0:
            return cls()
        ct = 0
        edges = []
        for s in skeletons:
            edges.append(s.edges.astype(np.int64) + ct)
            ct += len(s.vertices)
        return cls(np.concatenate([s.vertices for s in skeletons]), np.concatenate(edges).astype(np.uint32),
                   np.concatenate([s.radii for s in skeletons]), np.concatenate([s.vertex_types for s in skeletons]),
                   skeletons[0].id, skeletons[0].transform, skeletons[0].space)

    def empty(self):
        return self.vertices.size == 0 or self.edges.size == 0

    def clone(self):
        return Skeleton(self.vertices.copy(), self.edges.copy(), self.radii.copy(), self.vertex_types.copy(), self.id,
                        self.transform.copy(), self.space)

    def cable_length(self):
        d = self.vertices[self.edges[:, 1]] - self.vertices[self.edges[:, 0]]
        d = d * d
        return float(np.sum(np.sqrt(np.sum(d, axis=1))))

    def consolidate(self, remove_disconnected_vertices=True):
        if self.empty():
            return Skeleton(segid=self.id, transform=self.transform, space=self.space)
        nodes, first, inverse = np.unique(self.vertices, axis=0, return_index=True, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        edges = np.sort(inverse[self.edges.astype(np.int64)], axis=1)
        edges = np.unique(edges, axis=0)
        edges = edges[edges[:, 0] != edges[:, 1]]
        skel = Skeleton(nodes, edges, self.radii[first], self.vertex_types[first], self.id, self.transform, self.space)
        if remove_disconnected_vertices:
            skel = skel.remove_disconnected_vertices()
        return skel

    def remove_disconnected_vertices(self):
        used = np.unique(self.edges)
        if used.size == len(self.vertices):
            return self
        remap = -np.ones(len(self.vertices), np.int64)
        remap[used] = np.arange(used.size)
        return Skeleton(self.vertices[used], remap[self.edges.astype(np.int64)], self.radii[used], self.vertex_types[used],
                        self.id, self.transform, self.space)

    def components(self):
        skel = self.consolidate(remove_disconnected_vertices=False)
        if skel.edges.size == 0:
            return []
        index = defaultdict(set)
        for a, b in skel.edges.tolist():
            index[a].add(b)
            index[b].add(a)
        visited = set()

        def walk(start):
            edge_list = []
            stack, parents = [start], [-1]
            while stack:
                node, parent = stack.pop(), parents.pop()
                edge_list.append((node, parent) if node < parent else (parent, node))
                if node in visited:
                    continue
                visited.add(node)
                for child in sorted(index[node]):      # (the real package iterates a Python set here)
                    if child != parent:
                        stack.append(child)
                        parents.append(node)
            return edge_list[1:]

        forest = []
        for v in np.unique(skel.edges).tolist():
            if v not in visited:
                forest.append(walk(v))
        if len(forest) == 1:
            return [skel]
        out = []
        for edge_list in forest:
            e = np.unique(np.array(edge_list, dtype=np.int64), axis=0)   # (the walk emits the edge that closes a cycle twice)
            vid = np.unique(e)
            remap = -np.ones(len(skel.vertices), np.int64)
            remap[vid] = np.arange(vid.size)
            out.append(Skeleton(skel.vertices[vid], remap[e], skel.radii[vid], skel.vertex_types[vid], skel.id, skel.transform,
                                skel.space))
        return out


class Bbox:  # imported by post.py, never used on the postprocess path
    pass
