"""Minimal Skeleton container, attribute compatible with ``osteoid.Skeleton`` /
``cloudvolume.Skeleton`` (the type ``kimimaro.skeletonize`` returns; reference
kimimaro/trace.py:34,182-192 and kimimaro/intake.py:506-517,587-593).

``osteoid`` is a third-party package that is not in the reference tree; the
semantics below restate what the reference relies on:

* ``from_path``     vertices = the path, edges = consecutive pairs.
* ``simple_merge``  concatenate, offsetting edge indices.
* ``consolidate``   ``np.unique(vertices, axis=0)`` (lexicographically sorted
                    vertices), edges remapped, each edge sorted, rows sorted and
                    uniqued, self loops dropped, radii / vertex_types taken from
                    the first occurrence of each vertex.

When ``osteoid`` is importable ``to_osteoid()`` hands back the real class.
"""
from __future__ import annotations

import numpy as np


class Skeleton:
    def __init__(self, vertices=None, edges=None, radii=None, vertex_types=None,
                 segid=None, transform=None, space="voxel", extra_attributes=None):
        self.id = segid
        self.space = space
        if vertices is None:
            self.vertices = np.zeros((0, 3), dtype=np.float32)
        else:
            self.vertices = np.asarray(vertices, dtype=np.float32).reshape(-1, 3)
        if edges is None:
            self.edges = np.zeros((0, 2), dtype=np.uint32)
        else:
            self.edges = np.asarray(edges, dtype=np.uint32).reshape(-1, 2)
        n = self.vertices.shape[0]
        if radii is None:
            self.radii = -1 * np.ones(n, dtype=np.float32)
        else:
            self.radii = np.asarray(radii, dtype=np.float32)
        if vertex_types is None:
            self.vertex_types = np.zeros(n, dtype=np.uint8)
        else:
            self.vertex_types = np.asarray(vertex_types, dtype=np.uint8)
        if transform is None:
            self.transform = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float32)
        else:
            self.transform = np.asarray(transform, dtype=np.float32).reshape(3, 4)
        self.extra_attributes = extra_attributes if extra_attributes is not None else [
            {"id": "radius", "data_type": "float32", "num_components": 1},
            {"id": "vertex_types", "data_type": "uint8", "num_components": 1},
        ]

    # -- construction -------------------------------------------------------
    @classmethod
    def wrap(cls, vertices, edges, radii, segid, transform, space):
        """a Skeleton around arrays that already have the right dtypes and shapes (f32 (n,3), u32 (m,2), f32 (n)): no
        conversions, no copies -- the assembly of a 512^3 volume builds thousands of them on the host thread of its lane."""
        self = cls.__new__(cls)
        self.id = segid
        self.space = space
        self.vertices = vertices
        self.edges = edges
        self.radii = radii
        self.vertex_types = np.zeros(vertices.shape[0], dtype=np.uint8)
        self.transform = transform
        self.extra_attributes = [
            {"id": "radius", "data_type": "float32", "num_components": 1},
            {"id": "vertex_types", "data_type": "uint8", "num_components": 1},
        ]
        return self

    @classmethod
    def from_path(cls, path):
        path = np.asarray(path, dtype=np.float32).reshape(-1, 3)
        if path.shape[0] == 0:
            return cls()
        n = path.shape[0]
        edges = np.zeros((n - 1, 2), dtype=np.uint32)
        edges[:, 0] = np.arange(n - 1)
        edges[:, 1] = np.arange(1, n)
        return cls(path, edges)

    @classmethod
    def simple_merge(cls, skeletons):
        skeletons = list(skeletons)
        if len(skeletons) == 0:
            return cls()
        if type(skeletons[0]) is np.ndarray:
            skeletons = [skeletons]
        ct = 0
        edges = []
        for skel in skeletons:
            edges.append(skel.edges.astype(np.uint32) + np.uint32(ct))
            ct += skel.vertices.shape[0]
        return cls(
            vertices=np.concatenate([s.vertices for s in skeletons], axis=0),
            edges=np.concatenate(edges, axis=0),
            radii=np.concatenate([s.radii for s in skeletons], axis=0),
            vertex_types=np.concatenate([s.vertex_types for s in skeletons], axis=0),
            segid=skeletons[0].id,
            transform=skeletons[0].transform,
            space=skeletons[0].space,
        )

    # -- queries ------------------------------------------------------------
    def empty(self):
        return self.vertices.size == 0 or self.edges.size == 0

    def clone(self):
        return Skeleton(self.vertices.copy(), self.edges.copy(), self.radii.copy(),
                        self.vertex_types.copy(), self.id, self.transform.copy(), self.space)

    def cable_length(self):
        v1 = self.vertices[self.edges[:, 0]]
        v2 = self.vertices[self.edges[:, 1]]
        d = (v2 - v1).astype(np.float32)
        d *= d
        return float(np.sum(np.sqrt(np.sum(d, axis=1))))

    def voxel_space(self):
        """copy with vertices divided by the (diagonal) anisotropy transform (osteoid.Skeleton.voxel_space)."""
        if self.space == "voxel":
            return self.clone()
        skel = self.clone()
        diag = np.diag(self.transform[:, :3]).astype(np.float32)
        skel.vertices = (skel.vertices - self.transform[:, 3]) / diag
        skel.space = "voxel"
        return skel

    def physical_space(self):
        if self.space == "physical":
            return self.clone()
        skel = self.clone()
        diag = np.diag(self.transform[:, :3]).astype(np.float32)
        skel.vertices = skel.vertices * diag + self.transform[:, 3]
        skel.space = "physical"
        return skel

    def consolidate(self, remove_disconnected_vertices=True):
        """Duplicate vertices merged (rows sorted lexicographically), edges renumbered, each edge sorted, rows sorted and
        made unique, self loops dropped; radii / vertex_types from the first occurrence.  remove_disconnected_vertices
        (default True, as in osteoid / cloud-volume, so that code ported from kimimaro that calls a bare `.consolidate()`
        behaves the same) also drops vertices no edge refers to -- e.g. a one-vertex path (trace.py:182-184 calls it bare)."""
        if self.empty():
            return Skeleton(segid=self.id, transform=self.transform, space=self.space)
        eff_nodes, uniq_idx, inverse = np.unique(
            self.vertices, axis=0, return_index=True, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        eff_edges = inverse[self.edges.astype(np.int64)]
        eff_edges = np.sort(eff_edges, axis=1)
        eff_edges = np.unique(eff_edges, axis=0)
        eff_edges = eff_edges[eff_edges[:, 0] != eff_edges[:, 1]]
        skel = Skeleton(eff_nodes, eff_edges.astype(np.uint32), self.radii[uniq_idx],
                        self.vertex_types[uniq_idx], self.id, self.transform, self.space)
        return skel.remove_disconnected_vertices() if remove_disconnected_vertices else skel

    def remove_disconnected_vertices(self):
        used = np.unique(self.edges)
        if used.size == self.vertices.shape[0]:
            return self
        renumber = np.full(self.vertices.shape[0], -1, dtype=np.int64)
        renumber[used] = np.arange(used.size)
        return Skeleton(self.vertices[used], renumber[self.edges.astype(np.int64)], self.radii[used],
                        self.vertex_types[used], self.id, self.transform, self.space)

    def components(self):
        """Connected components of the consolidated skeleton as a list of Skeletons, ordered by their smallest vertex.
        A skeleton that is one component comes back consolidated; otherwise a component keeps its vertices in order
        and lists its edges as sorted unique rows.  (kimimaro/post.py's loop removal starts its cycle search at a
        component's first edge and counts a node's edges to find branch points, so neither the order nor duplicates are
        free: the reference's test_postprocess, automated_test.py:611-632, pins this.)"""
        skel = self.consolidate()
        if skel.edges.size == 0:
            return []
        n = skel.vertices.shape[0]
        nbrs = [[] for _ in range(n)]
        for a, b in skel.edges.tolist():      # rows are sorted and unique: both lists come out ascending
            nbrs[a].append(b)
            nbrs[b].append(a)
        seen = np.zeros(n, dtype=bool)
        forest = []
        for start in np.unique(skel.edges).tolist():
            if seen[start]:
                continue
            emitted = []
            todo = [(start, -1)]
            while todo:
                node, via = todo.pop()
                emitted.append((min(node, via), max(node, via)))
                if seen[node]:
                    continue
                seen[node] = True
                todo.extend((c, node) for c in nbrs[node] if c != via)
            forest.append(emitted[1:])
        if len(forest) == 1:
            return [skel]
        out = []
        for emitted in forest:
            e = np.unique(np.asarray(emitted, dtype=np.int64), axis=0)
            keep = np.unique(e)
            renumber = np.full(n, -1, dtype=np.int64)
            renumber[keep] = np.arange(keep.size)
            out.append(Skeleton(skel.vertices[keep], renumber[e], skel.radii[keep], skel.vertex_types[keep],
                                skel.id, skel.transform, skel.space))
        return out

    def to_swc(self):
        """SWC text (row f4, the format kimimaro_cli writes)."""
        n = self.vertices.shape[0]
        parent = -np.ones(n, dtype=np.int64)
        adj = [[] for _ in range(n)]
        for a, b in self.edges:
            adj[int(a)].append(int(b))
            adj[int(b)].append(int(a))
        seen = np.zeros(n, dtype=bool)
        order = []
        for s in range(n):
            if seen[s]:
                continue
            stack = [s]
            seen[s] = True
            while stack:
                u = stack.pop()
                order.append(u)
                for v in adj[u]:
                    if not seen[v]:
                        seen[v] = True
                        parent[v] = u
                        stack.append(v)
        newid = np.zeros(n, dtype=np.int64)
        newid[order] = np.arange(1, n + 1)
        lines = ["# SWC generated by kimimaro_amd", "# id type x y z radius parent"]
        for u in order:
            p = -1 if parent[u] < 0 else newid[parent[u]]
            x, y, z = self.vertices[u]
            lines.append("%d %d %.6f %.6f %.6f %.6f %d" % (
                newid[u], int(self.vertex_types[u]), x, y, z, self.radii[u], p))
        return "\n".join(lines) + "\n"

    def to_precomputed(self):
        """the Neuroglancer "precomputed" skeleton encoding cloud-volume / osteoid write (row f4): u32 vertex and edge
        counts, f32 vertices (n, 3), u32 edges (m, 2), then the vertex attributes in the order of
        `extra_attributes` (radius f32, vertex_types u8), all little endian."""
        v = np.ascontiguousarray(self.vertices, dtype="<f4")
        e = np.ascontiguousarray(self.edges, dtype="<u4")
        parts = [np.array([v.shape[0], e.shape[0]], dtype="<u4").tobytes(), v.tobytes(), e.tobytes()]
        for attr in self.extra_attributes:
            data = {"radius": self.radii, "vertex_types": self.vertex_types}[attr["id"]]
            parts.append(np.ascontiguousarray(data, dtype=np.dtype(attr["data_type"]).newbyteorder("<")).tobytes())
        return b"".join(parts)

    @classmethod
    def from_precomputed(cls, blob, segid=None, transform=None, space="physical"):
        """inverse of to_precomputed (default attribute set: radius f32, vertex_types u8)."""
        nv, ne = (int(x) for x in np.frombuffer(blob, dtype="<u4", count=2))
        off = 8
        verts = np.frombuffer(blob, dtype="<f4", count=3 * nv, offset=off).reshape(nv, 3)
        off += 12 * nv
        edges = np.frombuffer(blob, dtype="<u4", count=2 * ne, offset=off).reshape(ne, 2)
        off += 8 * ne
        radii = np.frombuffer(blob, dtype="<f4", count=nv, offset=off)
        off += 4 * nv
        vtypes = np.frombuffer(blob, dtype="u1", count=nv, offset=off)
        if off + nv != len(blob):
            raise ValueError("precomputed skeleton: %d trailing bytes" % (len(blob) - off - nv))
        return cls(verts.copy(), edges.copy(), radii.copy(), vtypes.copy(), segid=segid, transform=transform, space=space)

    def to_osteoid(self):
        try:
            import osteoid
        except ImportError:
            return self
        return osteoid.Skeleton(self.vertices, self.edges, self.radii, self.vertex_types,
                                segid=self.id, transform=self.transform, space=self.space)

    def __eq__(self, other):
        return (isinstance(other, Skeleton) and self.id == other.id
                and np.array_equal(self.vertices, other.vertices)
                and np.array_equal(self.edges, other.edges)
                and np.array_equal(self.radii, other.radii))

    def __repr__(self):
        return "Skeleton(segid=%r, vertices=%d, edges=%d, space=%r)" % (
            self.id, self.vertices.shape[0], self.edges.shape[0], self.space)
