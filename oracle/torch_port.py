"""CPU baseline ("port"): an op-for-op torch-CPU restatement of the reference's tensor-op sequence
for one frame step, keeping the SAME materialised intermediates (int64 index tensors, fp64 corner
weights, masked_select / cat compaction, two dense fp32 caches per frame) - i.e. the cost structure
of the reference's own CPU path, unlike the scalar C oracle.

TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/); never imported by the
product package.  The reference's Python cannot travel to the GPU box, so this file is what gets
timed on the node's host cores next to the GPU number.  In the build container it is checked
against the imported reference for identical outputs and comparable wall time
(tests/test_torch_port.py).

Sources restated: modules/extractor.py:24-120,309-345,533-681; modules/pipeline.py:74-171;
modules/integrator.py:15-196.
"""
import torch
import torch.nn.functional as F


def _unproject(depth, E, K):
    # extractor.py:82-120
    b, h, w = depth.shape
    n = h * w
    rows, cols = torch.meshgrid([torch.arange(h, dtype=torch.float), torch.arange(w, dtype=torch.float)], indexing='ij')
    rows = rows.contiguous().view(1, n, 1).repeat((b, 1, 1))
    cols = cols.contiguous().view(1, n, 1).repeat((b, 1, 1))
    z = depth.contiguous().view(b, n, 1)
    pix = torch.cat((cols, rows, z), dim=2).clone()
    Kinv = K.inverse().float()
    ones = torch.ones((b, 1, n))
    pix[:, :, 0] *= z[:, :, 0]
    pix[:, :, 1] *= z[:, :, 0]
    cam = torch.matmul(Kinv, torch.transpose(pix, dim0=1, dim1=2))
    cam = torch.cat((cam, ones), dim=1)
    world = torch.matmul(E[:3], cam)
    return torch.transpose(world, dim0=1, dim1=2)[:, :, :3]


def _ray_points(coords, eye, origin, resolution, half):
    # extractor.py:309-345
    centre = (coords - origin) / resolution
    eye_v = (eye - origin) / resolution
    direction = F.normalize(centre - eye_v, p=2, dim=2)
    pts = [centre]
    for i in range(1, half + 1):
        fwd = centre + i * 1.0 * direction
        back = centre - i * 1.0 * direction
        pts.append(fwd.clone())
        pts.insert(0, back.clone())
    return torch.stack(pts, dim=2)


def _corner_weights(points):
    # extractor.py:533-593
    centre = torch.floor(points) + 0.5 * torch.ones_like(points)
    nb = torch.sign(centre - points)
    idx = torch.floor(points)
    b, m, n, d = idx.shape
    points = points.contiguous().view(b * m * n, d)
    centre = centre.contiguous().view(b * m * n, d)
    idx = idx.view(b * m * n, d)
    nb = nb.view(b * m * n, d)
    alpha = torch.abs(points - centre)
    alpha_inv = 1 - alpha
    ws, ids = [], []
    for i in (0, 1):
        for j in (0, 1):
            for k in (0, 1):
                wx, ix = (alpha_inv[:, 0], idx[:, 0]) if i == 0 else (alpha[:, 0], idx[:, 0] + nb[:, 0])
                wy, iy = (alpha_inv[:, 1], idx[:, 1]) if j == 0 else (alpha[:, 1], idx[:, 1] + nb[:, 1])
                wz, iz = (alpha_inv[:, 2], idx[:, 2]) if k == 0 else (alpha[:, 2], idx[:, 2] + nb[:, 2])
                ws.append((wx * wy * wz).unsqueeze_(1))
                ids.append(torch.cat((ix.unsqueeze_(1), iy.unsqueeze_(1), iz.unsqueeze_(1)), dim=1).unsqueeze_(1))
    return torch.cat(ws, dim=1), torch.cat(ids, dim=1)


def _inside(indices, shape):
    xs, ys, zs = shape
    return ((indices[:, 0] >= 0) & (indices[:, 0] < xs) & (indices[:, 1] >= 0) & (indices[:, 1] < ys)
            & (indices[:, 2] >= 0) & (indices[:, 2] < zs))


def _gather(indices, volume, mask=None):
    if mask is not None:
        x = torch.masked_select(indices[:, 0], mask)
        y = torch.masked_select(indices[:, 1], mask)
        z = torch.masked_select(indices[:, 2], mask)
    else:
        x, y, z = indices[:, 0], indices[:, 1], indices[:, 2]
    return volume[x, y, z]


def _compact(indices, mask):
    x = torch.masked_select(indices[:, 0], mask)
    y = torch.masked_select(indices[:, 1], mask)
    z = torch.masked_select(indices[:, 2], mask)
    return torch.cat((x.unsqueeze_(1), y.unsqueeze_(1), z.unsqueeze_(1)), dim=1)


def extract(depth, extrinsics, intrinsics, tsdf_volume, weights_volume, origin, resolution, n_points=9):
    """Extractor.forward (extractor.py:24-79) + trilinear_interpolation (:640-681)."""
    K = intrinsics.float()
    E = extrinsics.float()
    b, h, w = depth.shape
    coords = _unproject(depth, E, K)
    pts = _ray_points(coords, E[:, :3, 3], origin, resolution, int((n_points - 1) / 2))
    bb, m, n, _ = pts.shape
    cw, idx = _corner_weights(pts)
    n1, n2, n3 = idx.shape
    idx = idx.contiguous().view(n1 * n2, n3).long()
    valid = _inside(idx, tsdf_volume.shape)
    valid_idx = torch.nonzero(valid)[:, 0]
    v_in = _gather(idx, tsdf_volume, valid)
    w_in = _gather(idx, weights_volume, valid)
    vbuf = -0.1 * torch.ones_like(valid).float()
    wbuf = torch.zeros_like(valid).float()
    vbuf[valid_idx] = v_in.float()
    wbuf[valid_idx] = w_in.float()
    vbuf = vbuf.view(cw.shape)
    wbuf = wbuf.view(cw.shape)
    fv = torch.sum(vbuf * cw, dim=1).view(bb, m, n)
    fw = torch.sum(wbuf * cw, dim=1).view(bb, m, n)
    return dict(fusion_values=fv.float(), fusion_weights=fw.float(), points=pts, depth=depth.view(b, h * w),
                indices=idx.view(n1, n2, n3).view(bb, m, n, 8, 3), weights=cw.view(bb, m, n, 8), pcl=coords)


def volume_update(values, tsdf_est, filtered_frame, sem_ids, scores, n_points=9, n_tail=7, init_value=0.1):
    """Pipeline._prepare_volume_update (pipeline.py:137-171)."""
    b, hw = filtered_frame.shape[0], filtered_frame.shape[-2] * filtered_frame.shape[-1]
    valid = (filtered_frame.view(b, hw, 1) != 0.).nonzero()[:, 1]
    pts = values['points'][:, :, :n_points].contiguous()
    upd = dict(points=pts[:, valid, :n_tail, :], indices=values['indices'][:, valid, :n_tail, :, :],
               weights=values['weights'][:, valid, :n_tail, :],
               values=torch.clamp(tsdf_est[:, valid, :n_tail], -init_value, init_value))
    if sem_ids is not None:
        s = sem_ids.view(b, hw, -1).contiguous().unsqueeze(-2).repeat(1, 1, tsdf_est.shape[2], 1)
        upd['semantics'] = s[:, valid, :n_tail, ...]
        c = scores.view(b, hw, -1).contiguous().unsqueeze(-2).repeat(1, 1, tsdf_est.shape[2], 1)
        upd['scores'] = c[:, valid, :n_tail, ...]
    return upd


def integrate(upd, values_volume, weights_volume, scores_volume=None, semantics_volume=None):
    """Integrator.forward (integrator.py:15-126), 'standard' implementation on the CPU."""
    values, indices, weights = upd['values'], upd['indices'], upd['weights']
    xs, ys, zs = values_volume.shape
    n1, n2, n3 = values.shape
    indices = indices.contiguous().view(n1 * n2 * n3, 8, 3).long()
    weights = weights.contiguous().view(n1 * n2 * n3, 8)
    values = values.contiguous().view(n1 * n2 * n3, 1).repeat(1, 8)
    i1, i2, i3 = indices.shape
    indices = indices.contiguous().view(i1 * i2, i3).long()
    weights = weights.contiguous().view(i1 * i2, 1).float()
    values = values.contiguous().view(i1 * i2, 1).float()
    valid = _inside(indices, values_volume.shape)
    valid_idx = torch.nonzero(valid)[:, 0]
    indices = _compact(indices, valid)
    weights = torch.masked_select(weights[:, 0], valid)
    values = torch.masked_select(values[:, 0], valid)
    update = weights * values
    lin = ys * zs * indices[:, 0] + zs * indices[:, 1] + indices[:, 2]
    cache = torch.zeros(xs * ys * zs).float()
    cache.index_add_(0, lin, weights)
    weights = _gather(indices, cache.view(xs, ys, zs))
    cache = torch.zeros(xs * ys * zs).float()
    cache.index_add_(0, lin, update)
    update = _gather(indices, cache.view(xs, ys, zs))
    del cache, lin
    w_old = _gather(indices, weights_volume).float()
    v_old = _gather(indices, values_volume).float()
    w_new = (w_old + weights).half()
    v_new = ((w_old * v_old + update) / (w_old + weights)).half()
    weights_volume[indices[:, 0], indices[:, 1], indices[:, 2]] = w_new
    values_volume[indices[:, 0], indices[:, 1], indices[:, 2]] = v_new
    if semantics_volume is not None:
        ids = upd['semantics'].contiguous().view(n1 * n2 * n3, 1).repeat(1, 8).contiguous().view(i1 * i2)[valid_idx]
        sc = upd['scores'].contiguous().view(n1 * n2 * n3, 1).repeat(1, 8).contiguous().view(i1 * i2)[valid_idx]
        ids_old = _gather(indices, semantics_volume)
        differs = ids_old != ids
        sc_old = _gather(indices, scores_volume).float()
        sc_new = torch.where(sc > sc_old, sc, sc_old).half()
        id_new = torch.where(sc > sc_old, ids, ids_old)[differs]
        di = indices[differs]
        semantics_volume[di[:, 0], di[:, 1], di[:, 2]] = id_new
        scores_volume[indices[:, 0], indices[:, 1], indices[:, 2]] = sc_new
    return values_volume, weights_volume, semantics_volume, scores_volume


def fuse(batch, vols, net, origin, resolution, depth_key='tof_depth', semantics=False, n_classes=30,
         n_points=9, n_tail=7, init_value=0.1):
    """Pipeline.fuse (pipeline.py:173-248) on CPU tensors.  vols: dict(tsdf, wgt[, ids, scores]) of
    torch CPU tensors, updated in place.  ``net`` is a torch module in eval mode."""
    b, _, h, w = batch['image'].shape
    sem_ids = scores = None
    if semantics:
        sem_ids = batch['semantic_gt'].long()
        scores = torch.ones_like(sem_ids).float()
    frame = batch[depth_key]
    filtered = torch.where(batch['mask'], frame, torch.zeros_like(frame))
    values = extract(frame, batch['extrinsics'], batch['intrinsics'], vols['tsdf'], vols['wgt'], origin, resolution, n_points)
    inputs = {'tsdf_values': values['fusion_values'].view(b, h, w, n_points),
              'tsdf_weights': values['fusion_weights'].view(b, h, w, n_points), 'tsdf_frame': frame.unsqueeze(-1)}
    if net.config.use_semantics:
        inputs['semantic_frame'] = (1 + sem_ids.unsqueeze(-1).float()) / n_classes
    inputs = {k: v.permute(0, -1, 1, 2).contiguous() for k, v in inputs.items()}
    est = net.forward(inputs).permute(0, 2, 3, 1)[:, :, :, :n_points].reshape(b, h * w, n_points)
    upd = volume_update(values, est, filtered, sem_ids.type(torch.uint8) if semantics else None, scores,
                        n_points, n_tail, init_value)
    del values
    integrate(upd, vols['tsdf'], vols['wgt'], vols.get('scores') if semantics else None,
              vols.get('ids') if semantics else None)
    return est
