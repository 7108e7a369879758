"""Seeded synthetic frame streams with the reference's batch-dict schema.

The reference ships no data; every parity test, the oracle, the
CPU baseline and bench.py share this generator.  A sample has
the keys the reference's loaders produce (dataset/replica.py:211-294):
``image [3,h,w] f32``, ``<input> [h,w] f32`` metric z-depth, ``mask [h,w] bool``,
``extrinsics [3,4] f64`` camera-to-world, ``intrinsics [3,3] f64``,
``semantic_gt [h,w] u8``, ``frame_id 'scene/trajectory/frame'``.

Scene: an axis-aligned box room with a few box solids, analytic ray-cast z-depth,
pinhole camera with 90 deg horizontal field of view on a smooth orbit.
Everything is numpy; nothing here touches the GPU.
"""
import numpy as np

ROOM_MIN = np.array([-2.4, -2.4, -1.4])
ROOM_MAX = np.array([2.4, 2.4, 1.4])
GRID_ORIGIN = np.array([-2.56, -2.56, -2.56])
GRID_EXTENT = 5.12

# (min corner, max corner) of the furniture solids; class ids start at 7
_SOLIDS = [
    (np.array([-2.4, -2.4, -1.4]), np.array([-1.2, -0.9, -0.6])),
    (np.array([1.1, -2.4, -1.4]), np.array([2.4, -1.5, 0.2])),
    (np.array([0.9, 1.2, -1.4]), np.array([1.9, 2.4, -0.4])),
]


def scene_sdf(points):
    """Signed distance (positive in free space) of ``points [...,3]`` to the room surfaces."""
    p = np.asarray(points, dtype=np.float64)
    inside = np.minimum(p - ROOM_MIN, ROOM_MAX - p).min(axis=-1)  # >0 inside the room
    sdf = inside
    for lo, hi in _SOLIDS:
        c = 0.5 * (lo + hi)
        e = 0.5 * (hi - lo)
        q = np.abs(p - c) - e
        outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
        box = outside + np.minimum(q.max(axis=-1), 0.0)  # >0 outside the solid
        sdf = np.minimum(sdf, box)
    return sdf


def grid_spec(grid):
    """(origin f64[3], resolution float, bbox f64[3,2]) of the ``grid``^3 volume."""
    res = GRID_EXTENT / grid
    bbox = np.stack([GRID_ORIGIN, GRID_ORIGIN + GRID_EXTENT], axis=1)
    return GRID_ORIGIN.copy(), res, bbox


def gt_volumes(grid, truncation=0.1, n_classes=30):
    """Ground-truth TSDF (f16) and label (u8) grids sampled at voxel centres."""
    origin, res, _ = grid_spec(grid)
    ax = origin[0] + (np.arange(grid) + 0.5) * res
    tsdf = np.empty((grid, grid, grid), dtype=np.float16)
    labels = np.zeros((grid, grid, grid), dtype=np.uint8)
    yy, zz = np.meshgrid(ax, ax, indexing='ij')
    for i in range(grid):  # slab by slab keeps the peak memory small at 512^3
        pts = np.stack([np.full_like(yy, ax[i]), yy, zz], axis=-1)
        sd = scene_sdf(pts)
        tsdf[i] = np.clip(sd, -truncation, truncation).astype(np.float16)
        near = np.abs(sd) < truncation
        lab = 1 + (np.floor((pts[..., 0] - ROOM_MIN[0]) / 0.8).astype(np.int64) % (n_classes - 1))
        labels[i] = np.where(near, lab, 0).astype(np.uint8)
    return tsdf, labels


def intrinsics(h, w):
    f = w / 2.0  # 90 deg hfov
    return np.array([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]], dtype=np.float64)


def camera_pose(t):
    """Camera-to-world [3,4] f64 at orbit parameter ``t`` (radians)."""
    eye = np.array([0.9 * np.cos(t), 0.9 * np.sin(t), 0.25 * np.sin(2.0 * t)])
    yaw = t + 0.35 * np.sin(0.5 * t)
    pitch = 0.20 * np.sin(1.5 * t)
    fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch)])
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)  # columns: camera x (right), y (down), z (forward)
    return np.concatenate([R, eye[:, None]], axis=1)


def _raycast(E, K, h, w):
    """Analytic z-depth and surface id for every pixel."""
    R, eye = E[:, :3], E[:, 3]
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], axis=-1)
    dw = dc @ R.T
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = 1.0 / dw
        # room: the ray starts inside, take the nearest exit plane
        t_hi = np.where(dw > 0, (ROOM_MAX - eye) * inv, np.where(dw < 0, (ROOM_MIN - eye) * inv, np.inf))
        axis = t_hi.argmin(axis=-1)
        t_best = t_hi.min(axis=-1)
        side = np.take_along_axis(dw, axis[..., None], axis=-1)[..., 0] > 0
        ids = (1 + 2 * axis + side).astype(np.int64)  # 1..6
        for s, (lo, hi) in enumerate(_SOLIDS):
            t0 = (lo - eye) * inv
            t1 = (hi - eye) * inv
            tn = np.minimum(t0, t1).max(axis=-1)
            tf = np.maximum(t0, t1).min(axis=-1)
            hit = (tn < tf) & (tn > 1e-6) & (tn < t_best)
            t_best = np.where(hit, tn, t_best)
            ids = np.where(hit, 7 + s, ids)
    return t_best, ids


class SyntheticStream:
    """Seeded frame stream for one scene.

    ``frame(i)`` returns the un-batched sample dict (numpy); ``batch(i)`` the
    batched torch dict a DataLoader with batch size 1 would hand to ``Pipeline.fuse``.
    """

    def __init__(self, h, w, grid, n_frames, scene='room_0', seed=1911, n_classes=30,
                 depth_key='tof_depth', noise_sigma=0.005, drop_fraction=0.01):
        self.h, self.w, self.grid = h, w, grid
        self.n_frames = n_frames
        self.scene = scene
        self.seed = seed
        self.n_classes = n_classes
        self.depth_key = depth_key
        self.noise_sigma = noise_sigma
        self.drop_fraction = drop_fraction
        self.K = intrinsics(h, w)
        self.origin, self.resolution, self.bbox = grid_spec(grid)
        self.scenes = [scene]

    def __len__(self):
        return self.n_frames

    def frame(self, i):
        rng = np.random.default_rng([self.seed, i])
        t = 2.0 * np.pi * i / max(self.n_frames, 40)
        E = camera_pose(t)
        depth_gt, ids = _raycast(E, self.K, self.h, self.w)
        depth = depth_gt + rng.normal(0.0, self.noise_sigma, size=depth_gt.shape)
        depth[rng.random(depth.shape) < self.drop_fraction] = 0.0
        depth = depth.astype(np.float32)
        mask = (depth > 0.05) & (depth < 5.0)
        scores = rng.uniform(0.5, 1.0, size=depth.shape).astype(np.float32)
        image = rng.standard_normal((3, self.h, self.w)).astype(np.float32)
        return {
            'item_id': i,
            'frame_id': '{}/0/{:06d}'.format(self.scene, i),
            'image': image,
            self.depth_key: depth,
            'depth_gt': depth_gt.astype(np.float32),
            'mask': mask,
            'extrinsics': E,
            'intrinsics': self.K.copy(),
            'semantic_gt': (ids % self.n_classes).astype(np.uint8),
            'semantic_scores': scores,
        }

    def batch(self, i):
        import torch
        out = {}
        for k, v in self.frame(i).items():
            if isinstance(v, np.ndarray):
                out[k] = torch.from_numpy(v).unsqueeze(0)
            elif isinstance(v, str):
                out[k] = [v]
            else:
                out[k] = torch.tensor([v])
        return out

    # the reference's Database pulls its grids from the dataset object
    # (modules/database.py:48-58: dataset.scenes, dataset.get_grid)
    def get_grid(self, scene, truncation, semantic_grid=True):
        from .database import Voxelgrid
        tsdf, labels = gt_volumes(self.grid, truncation, self.n_classes)
        g = Voxelgrid(self.resolution)
        g.from_array(tsdf, self.bbox)
        if semantic_grid:
            s = Voxelgrid(self.resolution)
            s.from_array(labels, self.bbox)
            return (g, s)
        return (g,)


class SyntheticDataset:
    """torch-Dataset-shaped collection of scenes (one SyntheticStream each), frames interleaved
    scene-major like the reference's file lists.  ``__getitem__`` returns the un-batched sample
    dict of tensors a DataLoader(batch_size=1) collates into the batch ``Pipeline.fuse`` expects."""

    def __init__(self, h, w, grid, frames_per_scene, scenes=('room_0',), seed=1911, n_classes=30,
                 depth_key='tof_depth'):
        self.streams = {s: SyntheticStream(h, w, grid, frames_per_scene, scene=s, seed=seed + 17 * i,
                                           n_classes=n_classes, depth_key=depth_key)
                        for i, s in enumerate(scenes)}
        self.scenes = list(scenes)
        self.frames_per_scene = frames_per_scene

    def __len__(self):
        return len(self.scenes) * self.frames_per_scene

    def __getitem__(self, item):
        import torch
        s = self.scenes[item // self.frames_per_scene]
        f = self.streams[s].frame(item % self.frames_per_scene)
        return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in f.items()}

    def get_grid(self, scene, truncation, semantic_grid=True):
        return self.streams[scene].get_grid(scene, truncation, semantic_grid)
