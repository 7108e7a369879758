"""Extractor: drop-in wrapper with the reference module's call signature and return dict
(modules/extractor.py:24-79), backed by ojf_extract.  ``Pipeline`` does not go through this
class (it lets the kernel write straight into the net's input rows); it exists so that callers
and tests of the reference's Extractor keep working."""
import torch
from torch import nn

from . import ops


class Extractor(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_points = config.FUSION_MODEL.n_points
        self.mode = 'ray'

    def forward(self, depth, extrinsics, intrinsics, tsdf_volume, weights_volume, origin, resolution,
                full=True):
        """Returns dict(fusion_values [1,N,P] f32, fusion_weights [1,N,P] f32, points [1,N,P,3] f64,
        depth [1,N], indices [1,N,P,8,3] i64, weights [1,N,P,8] f64, pcl [1,N,3] f32); batch size
        1 like the reference.  ``full=False`` skips the large index/weight tensors."""
        b, h, w = depth.shape
        if b != 1:
            raise ValueError('Extractor: batch size 1 only (the reference broadcasts one eye per batch)')
        dev = tsdf_volume.device
        Ki, E = ops.camera_arrays(intrinsics[0], extrinsics[0])
        d = depth[0].to(dev, torch.float32).contiguous()
        out = ops.extract(d, Ki, E, origin, resolution, tsdf_volume, weights_volume,
                          n_points=self.n_points, debug=full)
        n, p = h * w, self.n_points
        values = dict(fusion_values=out['fusion_values'].view(1, n, p),
                      fusion_weights=out['fusion_weights'].view(1, n, p),
                      depth=d.view(1, n))
        if full:
            values.update(points=out['points'].view(1, n, p, 3), indices=out['indices'].view(1, n, p, 8, 3),
                          weights=out['weights'].view(1, n, p, 8), pcl=out['pcl'].view(1, n, 3))
        return values
