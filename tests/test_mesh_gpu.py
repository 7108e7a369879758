"""MESH: ojf_mesh_extract through the C ABI against an independent numpy statement of the same
level set, and against analytic surfaces.

Parity vs the reference's skimage marching cubes is UNPINNED: skimage is not in this image and the triangulations
differ by construction (marching tetrahedra vs Lewiner tables).  What is pinned here:
* the number of triangles and the set of surface vertices, BIT-EXACT, against a numpy evaluation of the crossings on
  the 7 edge directions of the Kuhn triangulation (integer/index work -> exact; fp32 crossings with the same
  operation order -> exact);
* closedness / orientation / area / enclosed volume on a sphere, exactness on a plane;
* masking, labels (np.round rule of database.py:124-127), capacity handling, NaN cells.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KUHN_DIRS = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
CORNER = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
TETS = [(0, 5, 1, 6), (0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6)]


def np_triangle_count(vol, iso, valid):
    """Triangles marching tetrahedra emit: per tetrahedron 1 (one corner apart) or 2 (two against two)."""
    X, Y, Z = vol.shape
    inside = vol < iso
    corner = [inside[c[0]:X - 1 + c[0], c[1]:Y - 1 + c[1], c[2]:Z - 1 + c[2]].astype(np.int64) for c in CORNER]
    total = 0
    for t in TETS:
        k = sum(corner[c] for c in t)
        total += int(((((k == 1) | (k == 3)) * 1 + (k == 2) * 2) * valid).sum())
    return total


def np_cell_valid(vol, weights):
    X, Y, Z = vol.shape
    ok = ~np.isnan(vol)
    if weights is not None:
        ok &= weights > 0
    valid = np.ones((X - 1, Y - 1, Z - 1), dtype=bool)
    for c in CORNER:
        valid &= ok[c[0]:X - 1 + c[0], c[1]:Y - 1 + c[1], c[2]:Z - 1 + c[2]]
    return valid


def np_vertex_set(vol, iso, valid, origin, res):
    """Unique surface vertices: crossings on every Kuhn edge that belongs to at least one valid cell."""
    X, Y, Z = vol.shape
    vol = vol.astype(np.float32)
    iso = np.float32(iso)
    out = []
    pad = np.zeros((X + 1, Y + 1, Z + 1), dtype=bool)  # pad[i+1,j+1,k+1] = valid[i,j,k]
    pad[1:X, 1:Y, 1:Z] = valid
    for d in KUHN_DIRS:
        d = np.array(d)
        n = np.array([X, Y, Z]) - d
        a = vol[:n[0], :n[1], :n[2]]
        b = vol[d[0]:, d[1]:, d[2]:]
        cross = (a < iso) != (b < iso)
        # cells sharing the edge p..p+d: lower corners p - e with e in {0,1} on the axes where d is 0
        used = np.zeros(cross.shape, dtype=bool)
        free = [ax for ax in range(3) if d[ax] == 0]
        for m in range(1 << len(free)):
            e = np.zeros(3, dtype=int)
            for q, ax in enumerate(free):
                e[ax] = (m >> q) & 1
            sl = tuple(slice(1 - e[ax], 1 - e[ax] + n[ax]) for ax in range(3))
            used |= pad[sl]
        with np.errstate(invalid='ignore'):
            cross &= used
        idx = np.argwhere(cross)
        if idx.shape[0] == 0:
            continue
        va, vb = a[cross], b[cross]
        pa, pb = idx.astype(np.float32), (idx + d).astype(np.float32)
        swap = va > vb
        p0, p1 = np.where(swap[:, None], pb, pa), np.where(swap[:, None], pa, pb)
        v0, v1 = np.where(swap, vb, va), np.where(swap, va, vb)
        t = ((iso - v0) / (v1 - v0)).astype(np.float32)
        p = (p0 + (t[:, None] * (p1 - p0)).astype(np.float32)).astype(np.float32)
        out.append((np.asarray(origin, dtype=np.float64)[None] + p.astype(np.float64) * float(res)).astype(np.float32))
    return np.unique(np.concatenate(out, axis=0), axis=0) if out else np.zeros((0, 3), np.float32)


def sphere(n, r, centre=None, dtype=np.float16):
    c = np.full(3, (n - 1) / 2.0 + 0.137) if centre is None else np.asarray(centre)
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), axis=-1).astype(np.float64)
    return (np.linalg.norm(g - c, axis=-1) - r).astype(dtype), c


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('shape,masked', [((17, 13, 19), False), ((24, 24, 24), True), ((2, 2, 2), False), ((9, 33, 5), True)])
def test_count_and_vertices_match_numpy(shape, masked):
    from online_joint_depthfusion_and_semantic_amd import mesh
    rng = np.random.default_rng(hash(shape) % 1000)
    vol = rng.normal(size=shape).astype(np.float16)
    for ax in range(3):  # smooth a little so the surface is not pure noise, keep it fp16
        vol = ((vol.astype(np.float32) + np.roll(vol, 1, axis=ax).astype(np.float32)) / 2).astype(np.float16)
    weights = None
    if masked:
        weights = (rng.random(shape) > 0.2).astype(np.float16) * 3
        vol[tuple(rng.integers(0, s) for s in shape)] = np.nan
    origin, res = (-1.25, 0.5, 3.0), 0.0125
    valid = np_cell_valid(vol, weights)
    tri, _ = mesh.extract_triangles(dev(vol), None if weights is None else dev(weights), iso=0.03, origin=origin, resolution=res)
    assert tri.shape[0] == np_triangle_count(vol, np.float32(0.03), valid)
    got = np.unique(tri.reshape(-1, 3).cpu().numpy(), axis=0)
    want = np_vertex_set(vol, 0.03, valid, origin, res)
    assert got.shape == want.shape and np.array_equal(got, want)


def edge_table(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], axis=0)
    return e


def test_sphere_is_closed_oriented_and_accurate():
    from online_joint_depthfusion_and_semantic_amd import mesh
    n, r = 64, 20.3
    vol, c = sphere(n, r, dtype=np.float32)
    vol = np.clip(vol, -4, 4).astype(np.float16)  # truncated like a TSDF
    m = mesh.extract_mesh(dev(vol), resolution=1.0)
    v, f, nrm = m['vertices'].astype(np.float64), m['faces'], m['normals']
    assert abs(np.linalg.norm(v - c, axis=1) - r).max() < 0.03  # chord error h^2/(8r) + fp16 rounding of the values
    # closed 2-manifold: every directed edge once, and its reverse once
    e = edge_table(f)
    key = e[:, 0].astype(np.int64) * v.shape[0] + e[:, 1]
    rkey = e[:, 1].astype(np.int64) * v.shape[0] + e[:, 0]
    assert np.unique(key).size == key.size
    assert np.array_equal(np.sort(key), np.sort(rkey))
    assert v.shape[0] - e.shape[0] // 2 + f.shape[0] == 2  # Euler characteristic of a sphere
    a, b, cc = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = np.cross(b - a, cc - a)
    area = 0.5 * np.linalg.norm(fn, axis=1).sum()
    volume = (a * fn).sum() / 6.0  # divergence theorem; positive only for outward orientation
    assert abs(area / (4 * np.pi * r * r) - 1) < 5e-3
    assert abs(volume / (4 / 3 * np.pi * r ** 3) - 1) < 5e-3
    assert ((fn * ((a + b + cc) / 3 - c)).sum(axis=1) > 0).all()  # every face points to free space (positive side)
    assert ((nrm * (v - c)).sum(axis=1) / np.linalg.norm(v - c, axis=1)).min() > 0.95


def test_plane_is_exact():
    from online_joint_depthfusion_and_semantic_amd import mesh
    X, Y, Z = 12, 9, 16
    z0 = 6.25
    vol = np.broadcast_to((np.arange(Z) - z0)[None, None, :], (X, Y, Z)).astype(np.float16)
    res, origin = 0.5, (1.0, -2.0, 4.0)
    tri, _ = mesh.extract_triangles(dev(vol), origin=origin, resolution=res)
    t = tri.cpu().numpy().astype(np.float64)
    assert np.array_equal(t[..., 2], np.full(t.shape[:2], origin[2] + z0 * res))
    fn = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    assert (fn[:, 2] > 0).all() and np.abs(fn[:, :2]).max() == 0  # +z is the positive side
    assert 0.5 * np.linalg.norm(fn, axis=1).sum() == pytest.approx((X - 1) * (Y - 1) * res * res, rel=1e-12)
    assert tri.shape[0] == 8 * (X - 1) * (Y - 1)  # the six tetrahedra of a cell cut by a z-plane: 2+1+1+2+1+1


def test_weights_mask_and_nan_cells():
    from online_joint_depthfusion_and_semantic_amd import mesh
    vol, _ = sphere(32, 10.0)
    w = np.ones_like(vol)
    w[:16] = 0  # half of the volume unobserved
    full, _ = mesh.extract_triangles(dev(vol))
    half, _ = mesh.extract_triangles(dev(vol), dev(w))
    assert 0 < half.shape[0] < full.shape[0]
    assert half[..., 0].min().item() >= 16.0
    bad = vol.copy()
    bad[25, 16, 16] = np.nan  # a voxel next to the surface
    holed, _ = mesh.extract_triangles(dev(bad))
    assert holed.shape[0] < full.shape[0] and torch.isfinite(holed).all()
    empty, _ = mesh.extract_triangles(dev(np.ones((8, 8, 8), np.float16)))
    assert empty.shape == (0, 3, 3)
    m = mesh.extract_mesh(dev(np.ones((8, 8, 8), np.float16)))
    assert m['vertices'].shape == (0, 3) and m['faces'].shape == (0, 3)


def test_labels_follow_nearest_voxel_rule():
    from online_joint_depthfusion_and_semantic_amd import mesh
    n = 24
    vol, _ = sphere(n, 7.4)
    ids = np.random.default_rng(3).integers(0, 40, size=(n, n, n)).astype(np.uint8)
    res = 0.04
    tri, lab = mesh.extract_triangles(dev(vol), ids=dev(ids), resolution=res)
    p = tri.cpu().numpy().reshape(-1, 3).astype(np.float64) / res
    # vertices sit on Kuhn edges: integer on some axes, fractional on the others; np.round as database.py:124-127
    idx = np.round(p.astype(np.float32)).astype(np.int64)
    tie = np.abs(p - np.floor(p) - 0.5).min(axis=1) < 1e-4  # the division above can flip exact ties; skip those
    want = ids[idx[:, 0], idx[:, 1], idx[:, 2]]
    got = lab.cpu().numpy().reshape(-1)
    assert tie.mean() < 0.01 and np.array_equal(got[~tie], want[~tie])
    m = mesh.extract_mesh(dev(vol), ids=dev(ids), resolution=res)
    assert m['labels'].shape[0] == m['vertices'].shape[0] and m['rgb'].shape == (m['vertices'].shape[0], 3)
    assert m['rgb'].min() >= 0 and m['rgb'].max() <= 1


def test_capacity_is_respected_and_count_is_total():
    from online_joint_depthfusion_and_semantic_amd import _lib
    lib = _lib.load()
    vol = dev(sphere(32, 10.0)[0])
    count = torch.zeros(1, dtype=torch.int32, device='cuda')
    org = np.zeros(3)
    wsb = lib.ojf_mesh_workspace_bytes(32, 32, 32)
    assert wsb == 4 * 1 * 8 * 31 and lib.ojf_mesh_workspace_bytes(1, 32, 32) == 0
    ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    st = _lib.stream_ptr(vol.device)

    def call(vol_ptr, X, verts, lab, keys, cap, ws_ptr=ws.data_ptr(), ws_bytes=wsb):
        return lib.ojf_mesh_extract(vol_ptr, None, None, X, 32, 32, 0.0, org.ctypes.data, 1.0, ws_ptr, ws_bytes, verts, lab,
                                    keys, cap, count.data_ptr(), st)
    assert call(vol.data_ptr(), 32, None, None, None, 0) == 0
    total = int(count.item())
    assert total > 1000
    cap = 500
    buf = torch.full((cap + 64, 3, 3), -777.0, device='cuda')
    lab = torch.full((cap + 64, 3), 9, dtype=torch.uint8, device='cuda')
    keys = torch.full((cap + 64, 3), -5, dtype=torch.int64, device='cuda')
    assert call(vol.data_ptr(), 32, buf.data_ptr(), lab.data_ptr(), keys.data_ptr(), cap) == 0 and int(count.item()) == total
    assert (buf[cap:] == -777.0).all() and (lab[cap:] == 9).all() and (keys[cap:] == -5).all()  # nothing past the capacity
    assert (buf[:cap] != -777.0).all() and (lab[:cap] == 0).all() and (keys[:cap] >= 0).all()  # ids NULL -> label 0
    # argument errors fail loudly
    assert call(None, 32, None, None, None, 0) != 0
    assert call(vol.data_ptr(), 1, None, None, None, 0) != 0
    assert call(vol.data_ptr(), 32, None, None, None, 5) != 0
    assert call(vol.data_ptr(), 32, None, None, None, 0, ws_bytes=wsb - 4) != 0
    assert b'workspace' in lib.ojf_last_error()
    assert call(vol.data_ptr(), 32, None, None, None, 0, ws_ptr=None) != 0


def test_order_is_deterministic_and_keys_identify_vertices():
    from online_joint_depthfusion_and_semantic_amd import mesh
    vol = dev(sphere(48, 15.2)[0])
    t1, _, k1 = mesh.extract_triangles(vol, keys=True)
    t2, _, k2 = mesh.extract_triangles(vol, keys=True)
    assert torch.equal(t1, t2) and torch.equal(k1, k2)  # no atomics: same list every run
    # same key <=> same position
    pos, key = t1.reshape(-1, 3).cpu().numpy(), k1.reshape(-1).cpu().numpy()
    order = np.argsort(key, kind='stable')
    same = key[order][1:] == key[order][:-1]
    assert np.array_equal(pos[order][1:][same], pos[order][:-1][same])
    assert np.unique(key).size == np.unique(pos, axis=0).shape[0]
    # the key decodes to the grid edge the vertex lies on
    lin, code = key // 8, key % 8
    lo = np.stack([lin // (48 * 48), (lin // 48) % 48, lin % 48], axis=1)
    d = np.stack([(code >> 0) & 1, (code >> 1) & 1, (code >> 2) & 1], axis=1)
    assert (code > 0).all() and (pos >= lo - 1e-6).all() and (pos <= lo + d + 1e-6).all()
    # both welds describe the same mesh
    va, fa, _ = mesh.weld(t1, None, k1)
    vb, fb, _ = mesh.weld(t1)
    assert va.shape == vb.shape and fa.shape == fb.shape
    assert torch.equal(va[fa], vb[fb])
    vc, fc, _ = mesh.weld(t1, None, k1, n_voxels=vol.numel())  # prefix-sum ranking instead of the sort: same mesh
    assert torch.equal(vc, va) and torch.equal(fc, fa)


def test_database_mesh_modes(tmp_path):
    """Database.get_mesh / save('ply'|'test') of the reference API on a fused-looking scene."""
    from online_joint_depthfusion_and_semantic_amd import mesh
    from online_joint_depthfusion_and_semantic_amd.database import Database
    from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream
    from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config

    cfg = default_config(24, 32, semantics=True)
    st = SyntheticStream(24, 32, 32, 3)
    db = Database(st, database_config(cfg))
    scene = st.scene
    # stand in for a fused state: the ground-truth grid and labels
    db.scenes_est[scene].volume = torch.as_tensor(db.scenes_gt[scene].volume).to('cuda', torch.float16).contiguous()
    n = db.scenes_est[scene].volume.shape[0]
    ids = np.random.default_rng(0).integers(1, 30, size=(n, n, n)).astype(np.uint8)
    db.ids_est[scene].volume = torch.from_numpy(ids).cuda()
    v, f, nrm, rgb = db.get_mesh(scene, semantics=True)
    assert v.shape[0] > 0 and f.max() < v.shape[0] and nrm.shape == v.shape and rgb.shape == v.shape
    res = float(db.resolution[scene])
    assert v.min() >= 0 and v.max() <= (n - 1) * res * (1 + 1e-6)  # the reference's frame: voxel index * voxel size
    db.save(str(tmp_path), save_mode='test', scene_id=scene)
    base = scene.replace('/', '.')
    plain = mesh.load_ply(str(tmp_path / (base + '.ply')))
    sem = mesh.load_ply(str(tmp_path / (base + '_semantic.ply')))
    assert np.array_equal(plain['vertices'], v) and np.array_equal(plain['faces'], f)  # deterministic order
    idx = np.round(sem['vertices'] / np.float32(res)).astype(int)
    frac = np.abs(sem['vertices'] / res - np.floor(sem['vertices'] / res) - 0.5).min(axis=1) < 1e-3
    assert np.array_equal(sem['rgba'][~frac, 3], ids[idx[~frac, 0], idx[~frac, 1], idx[~frac, 2]])  # id in alpha
    db.save(str(tmp_path), save_mode='ply', scene_id=scene)
    with pytest.raises(ValueError):
        db.save(str(tmp_path), save_mode='obj', scene_id=scene)


# ---- reconstruction F-score on the device ---------------------------------------------------------------------------
@pytest.mark.parametrize('n_q,n_p,tau', [(5000, 7000, 0.05), (3000, 100, 0.3), (1, 1, 0.01), (4000, 4000, 1e-4), (2000, 3000, 2.5),
                                         (3000, 3000, 1e-7), (500, 800, 0.0)])
def test_points_within_matches_kdtree(n_q, n_p, tau):
    from scipy.spatial import cKDTree
    from online_joint_depthfusion_and_semantic_amd import mesh
    rng = np.random.default_rng(n_q + n_p)
    p = rng.random((n_p, 3)) * [2.0, 1.0, 3.0] + [-1.0, 5.0, 0.25]
    q = rng.random((n_q, 3)) * [2.4, 1.4, 3.4] + [-1.2, 4.8, 0.05]  # some queries leave the cell grid
    q[: n_q // 10] = p[rng.integers(0, n_p, n_q // 10)]  # exact hits at distance 0
    count, hit = mesh.points_within(dev(q), dev(p), tau)
    d = cKDTree(p).query(q)[0]
    want = d <= tau
    assert np.array_equal(hit.cpu().numpy().astype(bool), want) and count == int(want.sum())
    assert mesh.points_within(dev(q), dev(p[:0]), tau)[0] == 0 and mesh.points_within(dev(q[:0]), dev(p), tau)[0] == 0


def test_device_f_score_equals_host_definition():
    from online_joint_depthfusion_and_semantic_amd import mesh, metrics
    n, res, origin = 48, 0.02, (-0.3, 0.1, 1.0)
    gt = (np.clip(sphere(n, 15.2, dtype=np.float32)[0], -3, 3) * res).astype(np.float16)
    rng = np.random.default_rng(1)
    est = (gt.astype(np.float32) + rng.normal(size=gt.shape).astype(np.float32) * 0.6 * res).astype(np.float16)
    w = (rng.random(gt.shape) > 0.1).astype(np.float16)
    pe_host = metrics.surface_points(est, w > 0, origin, res)
    pe_dev = mesh.surface_points(dev(est), dev(w) > 0, origin, res)
    assert np.array_equal(pe_dev.cpu().numpy(), pe_host)  # same points, same order, f64
    pg_host = metrics.surface_points(gt, w > 0, origin, res)
    for tau in (None, 0.5 * res, 0.1 * res):
        t = 1.5 * res if tau is None else tau
        have = metrics.reconstruction_f_score(dev(est), dev(gt), dev(w), origin, res, tau)  # dispatches to the device
        host = metrics.reconstruction_f_score(est, gt, w, origin, res, tau)
        assert have == host and 0.05 < have['fscore'] <= 1.0, (have, host)  # integer hit counts: exact
    same = metrics.reconstruction_f_score(dev(gt), dev(gt), dev(w), origin, res)
    assert same == {'precision': 1.0, 'recall': 1.0, 'fscore': 1.0}
    none = metrics.reconstruction_f_score(dev(gt), dev(gt), dev(np.zeros_like(w)), origin, res)
    assert none == {'precision': 0.0, 'recall': 0.0, 'fscore': 0.0}


# ---- analytic surfaces from a committed fixture ---------------------------------------------------
def _analytic_sdf(case, pts):
    """Signed distance (negative inside the solid) of the fixture's closed-form solids at points [n, 3]."""
    kind = case['kind']
    if kind == 'torus':
        q = pts - np.asarray(case['centre'])
        ring = np.sqrt(q[:, 0] ** 2 + q[:, 1] ** 2) - case['R']
        return np.sqrt(ring ** 2 + q[:, 2] ** 2) - case['r']
    if kind == 'spheres':
        d = [np.linalg.norm(pts - np.asarray(c), axis=1) - r for c, r in zip(case['centres'], case['radii'])]
        return np.minimum.reduce(d)
    if kind == 'shell':  # solid between two concentric spheres
        rho = np.linalg.norm(pts - np.asarray(case['centre']), axis=1)
        return np.maximum(rho - case['R'], case['r'] - rho)
    raise ValueError(kind)


def _analytic_samples(case, n=6000, seed=3):
    """Points ON the analytic surface (for the surface -> mesh half of the Hausdorff check)."""
    rng = np.random.default_rng(seed)
    kind = case['kind']
    if kind == 'torus':
        u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
        R, r = case['R'], case['r']
        p = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], axis=1)
        return p + np.asarray(case['centre'])
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if kind == 'spheres':
        half = n // 2
        return np.concatenate([np.asarray(case['centres'][0]) + case['radii'][0] * d[:half],
                               np.asarray(case['centres'][1]) + case['radii'][1] * d[half:]])
    half = n // 2
    return np.asarray(case['centre']) + np.concatenate([case['R'] * d[:half], case['r'] * d[half:]])


def _components(n_vertices, faces):
    parent = np.arange(n_vertices)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for a, b in np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]]]):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
    return len({find(i) for i in range(n_vertices)})


def test_analytic_surfaces():
    """tests/golden/mesh_analytic_cases.json: torus (genus 1), two spheres (two components), a spherical shell (two nested
    surfaces of opposite orientation).  Closed 2-manifold, Euler characteristic and component count exact; area and enclosed
    volume against the closed forms; every vertex within the stated distance of the analytic surface and every analytic
    surface sample within one voxel of a vertex (two-sided Hausdorff bound)."""
    import json
    import os
    from scipy.spatial import cKDTree
    from online_joint_depthfusion_and_semantic_amd import mesh
    cases = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'mesh_analytic_cases.json')))['cases']
    n = 64
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), axis=-1).reshape(-1, 3).astype(np.float64)
    for case in cases:
        vol = np.clip(_analytic_sdf(case, g), -4, 4).astype(np.float16).reshape(n, n, n)
        m = mesh.extract_mesh(dev(vol), resolution=1.0)
        v, f = m['vertices'].astype(np.float64), m['faces']
        e = edge_table(f)
        key = e[:, 0].astype(np.int64) * v.shape[0] + e[:, 1]
        rkey = e[:, 1].astype(np.int64) * v.shape[0] + e[:, 0]
        assert np.unique(key).size == key.size and np.array_equal(np.sort(key), np.sort(rkey)), case['name']  # closed, oriented
        assert v.shape[0] - e.shape[0] // 2 + f.shape[0] == case['euler'], case['name']
        assert _components(v.shape[0], f) == case['components'], case['name']
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        fn = np.cross(b - a, c - a)
        area = 0.5 * np.linalg.norm(fn, axis=1).sum()
        volume = (a * fn).sum() / 6.0  # divergence theorem with outward (towards positive SDF) faces
        assert abs(area / case['area'] - 1) < case['rel_tol'], (case['name'], area)
        assert abs(volume / case['volume'] - 1) < case['rel_tol'], (case['name'], volume)
        assert np.abs(_analytic_sdf(case, v)).max() < case['max_vertex_distance'], (case['name'], float(np.abs(_analytic_sdf(case, v)).max()))
        gap = cKDTree(v).query(_analytic_samples(case))[0].max()
        assert gap < case['max_surface_gap'], (case['name'], float(gap))
