"""oracle.skeleton -- the few osteoid.Skeleton operations the reference's path assembly uses (kimimaro/trace.py:182-192,
kimimaro/intake.py:506-517,587-593), restated for the oracle pipeline.  TEST INFRASTRUCTURE ONLY; independent of the
product (nothing here imports kimimaro_amd).  osteoid itself is not in the reference tree: the semantics are those
of SURVEY.md Appendix A (from_path / simple_merge / consolidate)."""
from __future__ import annotations

import numpy as np


class Skeleton:
    def __init__(self, vertices=None, edges=None, radii=None, segid=None, transform=None, space="voxel"):
        self.vertices = np.zeros((0, 3), np.float32) if vertices is None else np.array(vertices, dtype=np.float32).reshape(-1, 3)
        self.edges = np.zeros((0, 2), np.uint32) if edges is None else np.array(edges, dtype=np.uint32).reshape(-1, 2)
        n = len(self.vertices)
        self.radii = np.full(n, -1, np.float32) if radii is None else np.array(radii, dtype=np.float32)
        self.vertex_types = np.zeros(n, np.uint8)
        self.id = segid
        self.space = space
        self.transform = np.hstack([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)]) if transform is None else transform

    @classmethod
    def from_path(cls, path):
        """vertices = the path, edges = consecutive pairs."""
        path = np.array(path, dtype=np.float32).reshape(-1, 3)
        k = np.arange(max(len(path) - 1, 0), dtype=np.uint32)
        return cls(path, np.stack([k, k + 1], axis=1))

    @classmethod
    def simple_merge(cls, skels):
        """concatenation with the edge indices shifted."""
        skels = list(skels)
        if not skels:
            return cls()
        shift = np.cumsum([0] + [len(s.vertices) for s in skels[:-1]])
        return cls(np.concatenate([s.vertices for s in skels]),
                   np.concatenate([s.edges.astype(np.int64) + o for s, o in zip(skels, shift)]),
                   np.concatenate([s.radii for s in skels]), skels[0].id, skels[0].transform, skels[0].space)

    def empty(self):
        return self.vertices.size == 0 or self.edges.size == 0

    def consolidate(self):
        """identical vertices merged (rows sorted lexicographically), edges renumbered, sorted, made unique,
        self loops dropped; per-vertex attributes come from the first occurrence; vertices no edge refers to are
        dropped (osteoid's default remove_disconnected_vertices=True: kimimaro/trace.py:182-184 calls it bare)."""
        if self.empty():
            return Skeleton(segid=self.id, transform=self.transform, space=self.space)
        order = np.lexsort((self.vertices[:, 2], self.vertices[:, 1], self.vertices[:, 0]))   # stable: first occurrence leads
        sv = self.vertices[order]
        new = np.ones(len(sv), bool)
        new[1:] = np.any(sv[1:] != sv[:-1], axis=1)
        rank = np.cumsum(new) - 1
        remap = np.empty(len(sv), np.int64)
        remap[order] = rank
        e = np.sort(remap[self.edges.astype(np.int64)], axis=1)
        e = e[e[:, 0] != e[:, 1]]
        e = np.unique(e, axis=0) if len(e) else e.reshape(0, 2)
        keep = order[new]
        used = np.zeros(len(keep), bool)
        used[e.ravel()] = True
        if not used.all():
            renum = np.cumsum(used) - 1
            keep, e = keep[used], renum[e]
        return Skeleton(self.vertices[keep], e, self.radii[keep], self.id, self.transform, self.space)

    def cable_length(self):
        d = self.vertices[self.edges[:, 1]] - self.vertices[self.edges[:, 0]]
        return float(np.sqrt((d.astype(np.float32) ** 2).sum(axis=1)).sum())

    def voxel_space(self):
        """a copy whose vertices are divided by the diagonal of `transform` (no-op for voxel-space skeletons)."""
        out = Skeleton(self.vertices.copy(), self.edges.copy(), self.radii.copy(), self.id, self.transform, self.space)
        if self.space != "voxel":
            out.vertices = (out.vertices - self.transform[:, 3]) / np.diag(self.transform[:, :3]).astype(np.float32)
            out.space = "voxel"
        return out
