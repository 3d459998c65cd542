#!/usr/bin/env python
"""Generate the marching-cubes case table used by csrc/mcubes.hip and by the CPU oracle.

The reference has no marching cubes, so the variant is ours and is defined
HERE, algorithmically, rather than by a hand-copied 256x16 table:

* corner i of a cell sits at offset (i & 1, (i >> 1) & 1, (i >> 2) & 1) = (x, y, z);
  a corner is "inside" when its value is > level; case index = sum(inside_i << i);
* the 12 cell edges are numbered axis * 4 + k (axis 0 = x, 1 = y, 2 = z), see EDGES;
* on every cell FACE the crossing edges are joined by segments; a face with four crossings (two
  inside corners on a diagonal) always cuts off each INSIDE corner separately.  The rule only looks
  at the face's own corner signs, so the two cells sharing a face draw the same segments and the
  surface is watertight (unlike the classic Lorensen-Cline table);
* segments are oriented with the inside on their left seen from outside the cell, chained into
  closed loops and fan-triangulated from the loop's lowest edge id; triangles therefore wind
  counter-clockwise seen from the outside (low-occupancy) side.

Writes monoport_amd/csrc/mc_tables.h (product) and oracle/mc_tables.npz (oracle): the same
numbers, generated once -- parity between GPU and oracle is self-parity, as documented.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CORNERS = [(i & 1, (i >> 1) & 1, (i >> 2) & 1) for i in range(8)]
EDGES = []  # (corner_a, corner_b) with a < b along the axis
for axis in range(3):
    for a in range(8):
        if not (a >> axis) & 1:
            EDGES.append((a, a | (1 << axis)))
EDGE_ID = {e: i for i, e in enumerate(EDGES)}
# owner node offset and axis of each cell edge (the edge belongs to its lower endpoint)
EDGE_OWNER = [(CORNERS[a], axis) for axis in range(3) for (a, b) in EDGES[axis * 4:axis * 4 + 4]]


def _edge(a, b):
    return EDGE_ID[(min(a, b), max(a, b))]


def _faces():
    """Each face as 4 corner ids in counter-clockwise order seen from OUTSIDE the cell."""
    faces = []
    for axis in range(3):
        u, v = [(1, 2), (2, 0), (0, 1)][axis]  # right-handed (axis, u, v)
        for side in (0, 1):
            quad = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                c = [0, 0, 0]
                c[axis] = side
                c[u] = du
                c[v] = dv
                quad.append(c[0] | (c[1] << 1) | (c[2] << 2))
            # (u, v) order is counter-clockwise seen from +axis; flip for the low face
            faces.append(quad if side == 1 else quad[::-1])
    return faces


FACES = _faces()


def case_triangles(case):
    inside = [(case >> i) & 1 for i in range(8)]
    nxt = {}  # directed segments: edge id -> edge id
    for quad in FACES:
        # walking the face counter-clockwise (seen from outside): an edge k joins quad[k] and
        # quad[k+1].  Passing from an inside corner to an outside corner = the surface EXITS the
        # inside-on-left ... orient segments so that inside corners are on the left.
        ins = [inside[c] for c in quad]
        crossings = [k for k in range(4) if ins[k] != ins[(k + 1) % 4]]
        if not crossings:
            continue
        # for each inside corner run (maximal run of consecutive inside corners), the segment goes
        # from the crossing AFTER the run to the crossing BEFORE it ... keep inside on the left:
        # travelling counter-clockwise around the inside run, we enter the run at edge k_in
        # (outside -> inside) and leave at k_out (inside -> outside).  A segment from the k_out
        # crossing to the k_in crossing has the run on its left.
        for k in range(4):
            if ins[k] == 0 and ins[(k + 1) % 4] == 1:  # entering an inside run at edge k
                j = (k + 1) % 4
                while ins[(j + 1) % 4] == 1:
                    j = (j + 1) % 4
                # run covers corners k+1 .. j; leaving edge is j (joins quad[j], quad[j+1])
                e_in = _edge(quad[k], quad[(k + 1) % 4])
                e_out = _edge(quad[j], quad[(j + 1) % 4])
                # (seen from outside the cell the inside run then lies on the segment's RIGHT, which
                # makes the fan triangles below wind counter-clockwise seen from the outside side
                # of the SURFACE -- verified by _orientation_check)
                assert e_in not in nxt
                nxt[e_in] = e_out
    # chain into loops
    tris = []
    todo = set(nxt)
    while todo:
        start = min(todo)
        loop = [start]
        todo.discard(start)
        e = nxt[start]
        while e != start:
            loop.append(e)
            todo.discard(e)
            e = nxt[e]
        assert len(loop) >= 3
        for k in range(1, len(loop) - 1):
            tris.append((loop[0], loop[k], loop[k + 1]))
    return tris


def build():
    all_tris = [case_triangles(c) for c in range(256)]
    max_t = max(len(t) for t in all_tris)
    table = -np.ones((256, max_t, 3), dtype=np.int8)
    count = np.zeros(256, dtype=np.uint8)
    for c, tris in enumerate(all_tris):
        count[c] = len(tris)
        for k, t in enumerate(tris):
            table[c, k] = t
    return table, count


def _orientation_check(table, count):
    """Triangles must face away from the inside: check on the single-corner cases."""
    for corner in range(8):
        case = 1 << corner
        assert count[case] == 1
        mids = []
        for e in table[case, 0]:
            a, b = EDGES[e]
            mids.append((np.array(CORNERS[a], float) + np.array(CORNERS[b], float)) / 2)
        n = np.cross(mids[1] - mids[0], mids[2] - mids[0])
        away = np.mean(mids, 0) - np.array(CORNERS[corner], float)
        assert np.dot(n, away) > 0, "triangle normal must point from inside to outside"


def main():
    table, count = build()
    _orientation_check(table, count)
    max_t = table.shape[1]
    edges = np.array(EDGES, dtype=np.int8)
    owner = np.array([[o[0][0], o[0][1], o[0][2], o[1]] for o in EDGE_OWNER], dtype=np.int8)
    np.savez_compressed(os.path.join(ROOT, "oracle", "mc_tables.npz"), tri=table, count=count,
                        edges=edges, owner=owner)
    lines = ["// Generated by tools/gen_mc_tables.py -- do not edit.",
             "// Face-consistent marching-cubes cases: corner i at offset (i&1, (i>>1)&1, (i>>2)&1);",
             "// edge e = axis*4 + k joins kMcEdgeCorner[e][0..1]; it is owned by node offset",
             "// kMcEdgeOwner[e][0..2] along axis kMcEdgeOwner[e][3].",
             "#pragma once",
             "namespace mp {",
             "constexpr int kMcMaxTris = %d;" % max_t,
             "__device__ __constant__ unsigned char kMcTriCount[256] = {%s};"
             % ", ".join(str(int(v)) for v in count),
             "__device__ __constant__ signed char kMcTriEdges[256][%d] = {" % (max_t * 3)]
    for c in range(256):
        lines.append("  {%s}," % ", ".join(str(int(v)) for v in table[c].reshape(-1)))
    lines.append("};")
    lines.append("__device__ __constant__ signed char kMcEdgeCorner[12][2] = {%s};"
                 % ", ".join("{%d, %d}" % (a, b) for a, b in EDGES))
    lines.append("__device__ __constant__ signed char kMcEdgeOwner[12][4] = {%s};"
                 % ", ".join("{%d, %d, %d, %d}" % tuple(int(v) for v in o) for o in owner))
    lines.append("}  // namespace mp")
    with open(os.path.join(ROOT, "monoport_amd", "csrc", "mc_tables.h"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("max triangles per cell:", max_t, "total:", int(count.sum()))


if __name__ == "__main__":
    main()
