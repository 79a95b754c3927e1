"""CPU restatement (numpy) of the two neighbourhood operators behind model/kpconv/ops/grid_subsample.py and radius_search.py.
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference forwards both to `geotransformer.ext` (GeoTransformer's C++ extension,
itself KPConv-PyTorch's cpp_wrappers), which is not vendored under /root/reference and cannot be built here; the functions below
restate its published algorithm:
  grid_subsampling.cpp: originCorner = floor(minCorner * (1 / dl)) * dl; cell = floor((p - originCorner) / dl) per axis;
      per cell: count and float32 sum of its points in input order; output = sum * (1.0 / count), cells in hash-map order
      (here: ascending map index iX + NX iY + NX NY iZ, i.e. ascending (iz, iy, ix));
  radius_neighbors.cpp: per cloud a kd-tree radius search (squared distance < r^2), results sorted by distance, indices offset
      into the stacked support set, rows filled with the total support count up to the fullest row's width
      (radius_search.py then cuts the rows at neighbor_limit).
Distances here are the canonical float32 distances of oracle/knn_oracle.c (ties: lowest index first)."""
import numpy as np


def grid_subsample(points: np.ndarray, lengths, voxel: float):
    outs, counts, start = [], [], 0
    dl = np.float32(voxel)
    for n in lengths:
        p = points[start:start + n].astype(np.float32)
        start += n
        if n == 0:
            counts.append(0)
            continue
        origin = np.floor(p.min(0) * (np.float32(1) / dl)) * dl
        cell = np.floor((p - origin) / dl).astype(np.int64)
        key = (cell[:, 2] << 26) | (cell[:, 1] << 13) | cell[:, 0]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        ends = np.r_[heads[1:], len(ks)]
        out = np.zeros((len(heads), 3), dtype=np.float32)
        ps = p[order]
        pos, live = heads.copy(), np.arange(len(heads))
        while len(live):   # float32 sums in input order inside every cell
            out[live] += ps[pos[live]]
            pos[live] += 1
            live = live[pos[live] < ends[live]]
        inv = (1.0 / (ends - heads).astype(np.float64)).astype(np.float32)
        outs.append(out * inv[:, None])
        counts.append(len(heads))
    return (np.concatenate(outs) if outs else np.zeros((0, 3), np.float32)), np.asarray(counts)


def radius_search(q_points, s_points, q_lengths, s_lengths, radius: float, neighbor_limit: int):
    import knn_c

    k, total_s = int(neighbor_limit), s_points.shape[0]
    rows, q0, s0 = [], 0, 0
    r2 = np.float32(radius) * np.float32(radius)
    for nq, ns in zip(q_lengths, s_lengths):
        if nq:
            out = np.full((nq, k), total_s, dtype=np.int64)
            if ns:
                idx, dist = knn_c.knn(np.ascontiguousarray(s_points[s0:s0 + ns]), np.ascontiguousarray(q_points[q0:q0 + nq]), k, return_dist=True)
                keep = (idx < ns) & (dist < r2)
                out[keep] = idx[keep] + s0
            rows.append(out)
        q0, s0 = q0 + nq, s0 + ns
    out = np.concatenate(rows) if rows else np.zeros((0, k), np.int64)
    width = int((out != total_s).sum(1).max()) if out.size else 0
    return out[:, :width]
