"""Mesh export of a fused volume: Database.get_mesh / the 'ply' and 'test' save modes of
the reference (modules/database.py:118-139,203-261, utils/saving.py:42-47), which run skimage's marching cubes
and trimesh on the host.  Here the iso-surface comes from the HIP marching-tetrahedra kernel
(csrc/ojf_mesh.hip, ``ojf_mesh_extract``), welding and normals are torch ops on the same device, and the PLY
writer is plain numpy - none of skimage / trimesh / plyfile is needed.

Parity note: the surface is the same zero level set (vertices are linear crossings on grid and cell-diagonal
edges), but the triangle list is NOT skimage's: marching tetrahedra emit about 2-3x the triangles of marching cubes
and no golden triangle list of the reference can be matched.  The tests pin it on analytic surfaces instead.
"""
import numpy as np
import torch

from . import _lib


def extract_triangles(tsdf, weights=None, ids=None, iso=0., origin=(0., 0., 0.), resolution=1., keys=False):
    """Triangle list of the level set ``iso`` of a cuda fp16 volume [X,Y,Z].

    weights: optional fp16 volume, cells touching a voxel with weight 0 are skipped (the reference meshes the raw
    volume, whose unobserved voxels hold the init value; pass None for exactly that behaviour).
    ids: optional u8 volume for per-vertex labels.  keys=True also returns the grid-edge id of every vertex.
    Returns (tri f32[T,3,3], labels u8[T,3] or None[, keys int64[T,3]]); the order is deterministic."""
    _lib.require_gpu()
    lib = _lib.load()
    assert tsdf.is_cuda and tsdf.dtype == torch.float16 and tsdf.is_contiguous() and tsdf.dim() == 3
    for vol, dt in ((weights, torch.float16), (ids, torch.uint8)):
        assert vol is None or (vol.is_cuda and vol.dtype == dt and vol.is_contiguous() and vol.shape == tsdf.shape)
    X, Y, Z = tsdf.shape
    org = np.ascontiguousarray(np.asarray(origin, dtype=np.float64).reshape(3))
    count = torch.zeros(1, dtype=torch.int32, device=tsdf.device)
    ws_bytes = lib.ojf_mesh_workspace_bytes(X, Y, Z)
    if ws_bytes == 0:
        raise ValueError('extract_triangles: volume shape {} not meshable'.format(tuple(tsdf.shape)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=tsdf.device)
    st = _lib.stream_ptr(tsdf.device)

    def run(verts, labels, edge, cap):
        rc = lib.ojf_mesh_extract(_lib.ptr(tsdf), _lib.ptr(weights), _lib.ptr(ids), X, Y, Z, float(iso),
                                  org.ctypes.data, float(resolution), _lib.ptr(ws), ws_bytes, _lib.ptr(verts),
                                  _lib.ptr(labels), _lib.ptr(edge), cap, _lib.ptr(count), st)
        _lib.check(rc, 'ojf_mesh_extract')
        return int(count.item())

    n = run(None, None, None, 0)
    tri = torch.empty((n, 3, 3), dtype=torch.float32, device=tsdf.device)
    labels = torch.empty((n, 3), dtype=torch.uint8, device=tsdf.device) if ids is not None else None
    edge = torch.empty((n, 3), dtype=torch.int64, device=tsdf.device) if keys else None
    if n:
        assert run(tri, labels, edge, n) == n
    return (tri, labels, edge) if keys else (tri, labels)


def weld(tri, labels=None, keys=None, n_voxels=None):
    """Indexed mesh from a triangle list.  With the kernel's edge keys: vertices ordered by grid edge, ranked by a
    prefix sum over the dense key space when ``n_voxels`` is given (no sort), else by one integer sort.  Without: vertices that are bit-equal merge (the kernel guarantees that for shared edges) and faces that
    collapse are dropped.  Returns (vertices [V,3], faces int64 [F,3], vertex_labels or None)."""
    flat = tri.reshape(-1, 3)
    if flat.shape[0] == 0:
        return flat, torch.zeros((0, 3), dtype=torch.int64, device=tri.device), (None if labels is None else labels.reshape(-1))
    if keys is not None:
        k = keys.reshape(-1)
        if n_voxels is not None and 8 * int(n_voxels) < (1 << 31):
            # rank of a key among the keys present = position in a prefix sum over the dense key space (8 per voxel):
            # scatter, scan, gather - no sort; same vertex order as the sorted-unique path below
            present = torch.zeros(8 * int(n_voxels), dtype=torch.uint8, device=flat.device)
            present[k] = 1
            rank = torch.cumsum(present, dim=0, dtype=torch.int32)
            del present
            inverse = rank[k].to(torch.int64) - 1
            n_verts = int(rank[-1].item())
            del rank
        else:
            uniq, inverse = torch.unique(k, return_inverse=True)
            n_verts = uniq.shape[0]
        verts = torch.empty((n_verts, 3), dtype=flat.dtype, device=flat.device)
        verts[inverse] = flat  # duplicates carry identical bits
        faces = inverse.reshape(-1, 3)
    else:
        verts, inverse = torch.unique(flat, dim=0, return_inverse=True)
        faces = inverse.reshape(-1, 3)
        keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
        faces = faces[keep]
    vlab = None
    if labels is not None:  # equal positions see the same nearest voxel, any representative will do
        vlab = torch.zeros(verts.shape[0], dtype=torch.uint8, device=tri.device)
        vlab[inverse] = labels.reshape(-1)
    return verts, faces, vlab


def vertex_normals(verts, faces):
    """Area-weighted vertex normals (pointing towards free space, like the face orientation)."""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    fn = torch.cross(b - a, c - a, dim=1)
    vn = torch.zeros_like(verts)
    for k in range(3):
        vn.index_add_(0, faces[:, k], fn)
    return vn / vn.norm(dim=1, keepdim=True).clamp_min(1e-20)


def default_palette():
    """256 distinct-ish colours, id 0 black (the reference's table comes from its own palette file; pass yours via
    ``palette=`` to get identical colours - the label itself always travels in the alpha channel)."""
    i = np.arange(256, dtype=np.uint32)
    table = np.stack([(i * 97 + 31) % 256, (i * 57 + 101) % 256, (i * 193 + 7) % 256], axis=1).astype(np.uint8)
    table[0] = 0
    return table


def extract_mesh(tsdf, weights=None, ids=None, iso=0., origin=(0., 0., 0.), resolution=1., palette=None):
    """Database.get_mesh equivalent: dict(vertices, faces, normals, labels, rgb) of numpy arrays.  rgb in [0,1] with
    id 0 shown grey as database.py:132-135 does; None without ids."""
    tri, labels, keys = extract_triangles(tsdf, weights, ids, iso, origin, resolution, keys=True)
    verts, faces, vlab = weld(tri, labels, keys, n_voxels=tsdf.numel())
    normals = vertex_normals(verts, faces) if faces.shape[0] else torch.zeros_like(verts)
    out = {'vertices': verts.cpu().numpy(), 'faces': faces.to(torch.int32).cpu().numpy(), 'normals': normals.cpu().numpy(),
           'labels': None, 'rgb': None}
    if vlab is not None:
        table = np.array(default_palette() if palette is None else palette, dtype=np.float64)
        table[0] = [128, 128, 128]
        out['labels'] = vlab.cpu().numpy()
        out['rgb'] = table[out['labels']] / 255.0
    return out


# ---- reconstruction F-score on the device (definition: metrics.py surface_points / f_score) --------------------
def surface_points(tsdf, mask, origin, resolution):
    """metrics.surface_points on cuda tensors: zero crossings along the three grid axes between two observed voxels,
    same f64 arithmetic, same order (axis-major, then voxel order).  Returns f64 [M,3] on the device."""
    t = torch.nan_to_num(tsdf.to(torch.float32))
    org = torch.as_tensor(np.asarray(origin, dtype=np.float64), device=tsdf.device)
    pts = []
    for axis in range(3):
        n = t.shape[axis] - 1
        ta, tb = t.narrow(axis, 0, n), t.narrow(axis, 1, n)
        cross = mask.narrow(axis, 0, n) & mask.narrow(axis, 1, n) & ((ta < 0) != (tb < 0))
        idx = cross.nonzero().to(torch.float64)
        if idx.shape[0] == 0:
            continue
        va, vb = ta[cross].to(torch.float64), tb[cross].to(torch.float64)
        idx[:, axis] += va / (va - vb)
        pts.append((idx + 0.5) * float(resolution) + org)
    return torch.cat(pts, dim=0) if pts else torch.zeros((0, 3), dtype=torch.float64, device=tsdf.device)


def points_within(query, points, tau):
    """Number of rows of ``query`` [N,3] with a row of ``points`` [M,3] within ``tau`` (cuda tensors, evaluated in
    f64 by ojf_points_within exactly like ``cKDTree(points).query(query)[0] <= tau``).  Returns (count, hit u8[N])."""
    _lib.require_gpu()
    lib = _lib.load()
    q = query.to(torch.float64).contiguous()
    p = points.to(torch.float64).contiguous()
    hit = torch.zeros(q.shape[0], dtype=torch.uint8, device=q.device)
    if q.shape[0] == 0 or p.shape[0] == 0:
        return 0, hit
    cell = max(float(tau), 1e-30)
    lo = p.min(dim=0).values
    span = p.max(dim=0).values - lo
    # coarsen the bins when tau is tiny against the extent: at most ~2^24 cells (64 MB of offsets)
    extent = [max(float(v), 0.0) for v in span.tolist()]
    cell = max(cell, (max(extent[0], cell) * max(extent[1], cell) * max(extent[2], cell) / float(1 << 24)) ** (1.0 / 3.0))
    c = torch.floor((p - lo) / cell).to(torch.int64)
    G = [int(v) + 1 for v in c.max(dim=0).values.tolist()]
    key = (c[:, 0] * G[1] + c[:, 1]) * G[2] + c[:, 2]
    key, order = torch.sort(key)
    p = p[order].contiguous()
    n_cells = G[0] * G[1] * G[2]
    start = torch.searchsorted(key, torch.arange(n_cells + 1, device=p.device, dtype=torch.int64)).to(torch.int32).contiguous()
    n_hit = torch.zeros(1, dtype=torch.int32, device=p.device)
    org = np.ascontiguousarray(lo.cpu().numpy())
    rc = lib.ojf_points_within(_lib.ptr(q), q.shape[0], _lib.ptr(p), _lib.ptr(start), org.ctypes.data, cell, G[0], G[1], G[2],
                               float(tau), _lib.ptr(hit), _lib.ptr(n_hit), _lib.stream_ptr(q.device))
    _lib.check(rc, 'ojf_points_within')
    return int(n_hit.item()), hit


def f_score(points_est, points_gt, tau):
    """metrics.f_score on the device: precision, recall and F at distance ``tau``."""
    if points_est.shape[0] == 0 or points_gt.shape[0] == 0:
        return {'precision': 0.0, 'recall': 0.0, 'fscore': 0.0}
    p = points_within(points_est, points_gt, tau)[0] / points_est.shape[0]
    r = points_within(points_gt, points_est, tau)[0] / points_gt.shape[0]
    return {'precision': p, 'recall': r, 'fscore': 0.0 if p + r == 0 else 2 * p * r / (p + r)}


def reconstruction_f_score(est, gt, weights, origin, resolution, tau=None):
    """metrics.reconstruction_f_score with the volumes resident on the device (no D2H of the grids)."""
    mask = weights > 0
    tau = 1.5 * float(resolution) if tau is None else tau
    return f_score(surface_points(est, mask, origin, resolution), surface_points(gt, mask, origin, resolution), tau)


def save_ply(filename, vertices, faces, normals=None, rgba=None):
    """Binary little-endian PLY with the element/property names trimesh writes (x y z [nx ny nz] [red green blue
    alpha]; face vertex_indices), so files load wherever the reference's do."""
    vertices = np.asarray(vertices, dtype='<f4')
    faces = np.asarray(faces, dtype='<i4')
    fields = [('x', '<f4'), ('y', '<f4'), ('z', '<f4')]
    if normals is not None:
        fields += [('nx', '<f4'), ('ny', '<f4'), ('nz', '<f4')]
    if rgba is not None:
        fields += [('red', 'u1'), ('green', 'u1'), ('blue', 'u1'), ('alpha', 'u1')]
    vrec = np.zeros(vertices.shape[0], dtype=fields)
    vrec['x'], vrec['y'], vrec['z'] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    if normals is not None:
        n = np.asarray(normals, dtype='<f4')
        vrec['nx'], vrec['ny'], vrec['nz'] = n[:, 0], n[:, 1], n[:, 2]
    if rgba is not None:
        c = np.asarray(rgba, dtype=np.uint8)
        vrec['red'], vrec['green'], vrec['blue'], vrec['alpha'] = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    frec = np.zeros(faces.shape[0], dtype=[('n', 'u1'), ('v', '<i4', (3,))])
    frec['n'] = 3
    frec['v'] = faces
    names = {'<f4': 'float', 'u1': 'uchar'}
    header = ['ply', 'format binary_little_endian 1.0', 'comment ojf_mesh (marching tetrahedra)',
              'element vertex {}'.format(vertices.shape[0])]
    header += ['property {} {}'.format(names[t], n) for n, t in fields]
    header += ['element face {}'.format(faces.shape[0]), 'property list uchar int vertex_indices', 'end_header']
    with open(filename, 'wb') as f:
        f.write(('\n'.join(header) + '\n').encode('ascii'))
        f.write(vrec.tobytes())
        f.write(frec.tobytes())


def load_ply(filename):
    """Reader for the files save_ply writes (tests and downstream tools); returns dict(vertices, faces, normals, rgba)."""
    with open(filename, 'rb') as f:
        assert f.readline().strip() == b'ply'
        fields, n_vert, n_face, element = [], 0, 0, None
        while True:
            line = f.readline().decode('ascii').split()
            if line[0] == 'end_header':
                break
            if line[0] == 'element':
                element = line[1]
                if element == 'vertex':
                    n_vert = int(line[2])
                else:
                    n_face = int(line[2])
            elif line[0] == 'property' and element == 'vertex':
                fields.append((line[2], {'float': '<f4', 'uchar': 'u1'}[line[1]]))
        vrec = np.frombuffer(f.read(np.dtype(fields).itemsize * n_vert), dtype=fields)
        frec = np.frombuffer(f.read(13 * n_face), dtype=[('n', 'u1'), ('v', '<i4', (3,))])
    out = {'vertices': np.stack([vrec['x'], vrec['y'], vrec['z']], axis=1), 'faces': frec['v'].copy(), 'normals': None,
           'rgba': None}
    if 'nx' in vrec.dtype.names:
        out['normals'] = np.stack([vrec['nx'], vrec['ny'], vrec['nz']], axis=1)
    if 'alpha' in vrec.dtype.names:
        out['rgba'] = np.stack([vrec['red'], vrec['green'], vrec['blue'], vrec['alpha']], axis=1)
    return out
