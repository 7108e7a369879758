"""Integrator: drop-in module with the reference's constructor and ``forward`` signature
(modules/integrator.py:15-126).  ``forward(updates, ...)`` consumes the materialised updates dict the
reference's ``Pipeline._prepare_volume_update`` builds (ojf_integrate_entries); ``forward_frame`` is the
lean entry point this engine's own Pipeline uses (indices/weights recomputed from depth + pose,
ojf_integrate), which never materialises those tensors."""
import torch
from torch import nn

from . import ops
from ._lib import MODE_FAST


class Integrator(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.device = getattr(config.SETTINGS, 'device', None)
        self.implementation = getattr(config.SETTINGS, 'implementation', 'efficient')
        self._workspaces = {}

    def forward(self, updates, values_volume, weights_volume, scores_volume, semantics_volume, test=True):
        """Reference signature and return order (values, weights, semantics, scores), integrator.py:15,126.
        updates: values [1,Nv,T] f32, indices [1,Nv,T,8,3] i64, weights [1,Nv,T,8] f64 and, with semantics,
        semantics [1,Nv,T,1] u8 / scores [1,Nv,T,1] f32.  Volumes are updated in place."""
        dev = values_volume.device
        values = updates['values'].to(dev).float().reshape(-1).contiguous()
        R = values.numel()
        indices = updates['indices'].to(dev).long().reshape(R, 8, 3).contiguous()
        weights = updates['weights'].to(dev).double().reshape(R, 8).contiguous()
        sem = bool(self.config.DATA.semantics) and test and semantics_volume is not None
        ids = updates['semantics'].to(dev).to(torch.uint8).reshape(R).contiguous() if sem else None
        scores = updates['scores'].to(dev).float().reshape(R).contiguous() if sem else None
        ws = getattr(self, '_entry_ws', None)
        if ws is None or ws.shape != tuple(values_volume.shape) or ws.max_rows < R or ws.buf.device != dev:
            ws = self._entry_ws = ops.EntryWorkspace(values_volume.shape, max(R, 1), dev)
        ops.integrate_entries(values, indices, weights, values_volume, weights_volume, ws, ids, scores,
                              semantics_volume if sem else None, scores_volume if sem else None)
        return values_volume, weights_volume, semantics_volume, scores_volume

    def forward_frame(self, depth_filtered, extrinsics, intrinsics, origin, resolution, tsdf_est,
                      values_volume, weights_volume, scores_volume=None, semantics_volume=None,
                      sem_ids=None, sem_scores=None, test=True, mode=MODE_FAST):
        """Integrate one frame in place and return (values, weights, semantics, scores) in the
        reference's order (integrator.py:126)."""
        h, w = depth_filtered.shape[-2:]
        fm = self.config.FUSION_MODEL
        key = (tuple(values_volume.shape), h, w, mode)
        if key not in self._workspaces:
            self._workspaces[key] = ops.IntegrateWorkspace(values_volume.shape, h, w, fm.n_tail_points, mode,
                                                           values_volume.device)
        Ki, E = ops.camera_arrays(intrinsics, extrinsics)
        sem = bool(self.config.DATA.semantics) and test and semantics_volume is not None
        ops.integrate(depth_filtered.reshape(h, w).contiguous(), Ki, E, origin, resolution,
                      tsdf_est.reshape(h * w, -1).contiguous(), values_volume, weights_volume,
                      self._workspaces[key], n_points=fm.n_points, n_tail=fm.n_tail_points,
                      trunc=self.config.DATA.init_value,
                      sem_ids=sem_ids if sem else None, sem_scores=sem_scores if sem else None,
                      id_vol=semantics_volume if sem else None, score_vol=scores_volume if sem else None, mode=mode)
        return values_volume, weights_volume, semantics_volume, scores_volume
