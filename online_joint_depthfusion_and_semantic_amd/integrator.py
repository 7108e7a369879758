"""Integrator: module with the reference's constructor; the scatter itself is ojf_integrate
(see Pipeline.fuse).  The reference's ``forward(updates, ...)`` consumed materialised int64
indices / fp64 weights (modules/integrator.py:15-126); those tensors do not exist in this engine,
so the drop-in entry point takes the frame and pose instead."""
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
