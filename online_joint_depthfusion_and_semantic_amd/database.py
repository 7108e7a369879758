"""Per-scene volume store: counterpart of the reference's ``modules/database.py`` Database.

Same constructor, attributes and methods, with one deliberate difference in
*where* the state lives: the four volumes of every scene (TSDF fp16, weights fp16, semantic ids
u8, semantic scores fp16) are HIP device tensors resident in HBM for the whole stream, the GT
volumes included (the reference re-uploads the GT grid every training frame, extractor.py:47).
Pipeline mutates them in place.  ``to_numpy()`` moves the state to the host exactly like the
reference's (database.py:383-393), after which ``filter`` / ``evaluate`` run on numpy arrays;
while the state is on the device they run as HIP kernels (ojf_volume_*).
"""
import os

import numpy as np
import torch

from . import ops
from .metrics import evaluation, semantic_evaluation, semantic_metrics_from_counts


class Voxelgrid:
    """The 30 lines of deps/graphics' Voxelgrid the hot path uses (voxelgrid.py:157-161)."""

    def __init__(self, resolution):
        self.resolution = resolution
        self.volume = None
        self.bbox = None
        self.origin = None

    def from_array(self, array, bbox):
        self.volume = array
        self.bbox = np.asarray(bbox, dtype=np.float64)
        self.origin = self.bbox[:, 0].copy()

    @property
    def shape(self):
        return self.volume.shape


def _is_dev(v):
    return torch.is_tensor(v) and v.is_cuda


class Database(torch.utils.data.Dataset):

    def __init__(self, dataset, config):
        super().__init__()
        self.device = torch.device(config.device)
        self.implementation = config.implementation
        self.transform = config.transform
        self.initial_value = config.init_value
        self.semantics = config.semantics
        self.semantic_grid = config.semantic_grid
        self.pad = config.pad
        if self.semantics:
            self.n_classes = config.n_classes

        self.scenes = []
        self.state = {}  # True once a scene's grid holds integrated frames
        self.origin, self.resolution = {}, {}
        self.scenes_gt, self.scenes_est, self.fusion_weights = {}, {}, {}
        self.ids_gt, self.ids_est, self.scores = {}, {}, {}

        for s in dataset.scenes:
            self.scenes.append(s)
            try:
                grid = dataset.get_grid(s, self.initial_value, self.semantic_grid)
            except Exception:  # no ground truth available (database.py:52-53)
                grid = dataset.create_grid(s, self.initial_value)
            self.state[s] = False
            self.scenes_gt[s] = grid[0]
            self.origin[s] = np.asarray(grid[0].origin, dtype=np.float64)
            self.resolution[s] = grid[0].resolution
            shape = tuple(grid[0].volume.shape)
            self.scenes_est[s] = Voxelgrid(grid[0].resolution)
            self.scenes_est[s].from_array(np.full(shape, self.initial_value, dtype=np.float16), grid[0].bbox)
            self.fusion_weights[s] = np.zeros(shape, dtype=np.float16)
            if self.semantics:
                if self.semantic_grid:
                    self.ids_gt[s] = grid[1]
                self.ids_est[s] = Voxelgrid(grid[0].resolution)
                self.ids_est[s].from_array(np.zeros(shape, dtype=np.uint8), grid[0].bbox)
                self.scores[s] = Voxelgrid(grid[0].resolution)
                self.scores[s].from_array(np.zeros(shape, dtype=np.float16), grid[0].bbox)
        self.to_torch()

    # ---- access ---------------------------------------------------------------------------
    def __getitem__(self, item):
        sample = {'origin': self.origin[item], 'resolution': self.resolution[item],
                  'gt': self.scenes_gt[item].volume, 'current': self.scenes_est[item].volume,
                  'weights': self.fusion_weights[item]}
        if self.semantics:
            sample['ids_est'] = self.ids_est[item].volume
            sample['scores'] = self.scores[item].volume
            if self.semantic_grid:
                sample['ids_gt'] = self.ids_gt[item].volume
        else:
            sample.update(histograms=None, ids_est=None, ids_gt=None, scores=None)
        return sample

    def __len__(self):
        return len(self.scenes_gt)

    # ---- residency ------------------------------------------------------------------------
    def _dev(self, a):
        if torch.is_tensor(a):
            return a.to(self.device).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def to_torch(self, gt=True, scenes=None):
        scenes = self.scenes if scenes is None else [scenes]
        for s in scenes:
            self.scenes_est[s].volume = self._dev(self.scenes_est[s].volume)
            self.fusion_weights[s] = self._dev(self.fusion_weights[s])
            if gt:
                o = self.origin[s]
                self.origin[s] = o.double().cpu() if torch.is_tensor(o) else torch.from_numpy(np.asarray(o, np.float64))
                self.scenes_gt[s].volume = self._dev(self.scenes_gt[s].volume)
            if self.semantics:
                self.ids_est[s].volume = self._dev(self.ids_est[s].volume)
                self.scores[s].volume = self._dev(self.scores[s].volume)
                if gt and self.semantic_grid:
                    self.ids_gt[s].volume = self._dev(self.ids_gt[s].volume)

    def to_numpy(self):
        def host(a):
            return a.detach().cpu().numpy() if torch.is_tensor(a) else a
        for s in self.scenes:
            self.origin[s] = host(self.origin[s])
            self.scenes_est[s].volume = host(self.scenes_est[s].volume)
            self.scenes_gt[s].volume = host(self.scenes_gt[s].volume)
            self.fusion_weights[s] = host(self.fusion_weights[s])
            if self.semantics:
                self.ids_est[s].volume = host(self.ids_est[s].volume)
                self.scores[s].volume = host(self.scores[s].volume)
                if self.semantic_grid:
                    self.ids_gt[s].volume = host(self.ids_gt[s].volume)

    def reset(self, scene_id=None):
        """database.py:351-370; device volumes are re-filled in place by a HIP kernel instead of
        being re-allocated on the host and uploaded."""
        for s in ([scene_id] if scene_id else self.scenes):
            self.state[s] = False
            if _is_dev(self.scenes_est[s].volume):
                ops.volume_fill(self.scenes_est[s].volume, self.initial_value)
                ops.volume_fill(self.fusion_weights[s], 0.0)
                if self.semantics:
                    ops.volume_fill(self.ids_est[s].volume, 0)
                    ops.volume_fill(self.scores[s].volume, 0.0)
            else:
                shape = self.scenes_est[s].volume.shape
                self.scenes_est[s].volume = np.full(shape, self.initial_value, dtype=np.float16)
                self.fusion_weights[s] = np.zeros(shape, dtype=np.float16)
                if self.semantics:
                    self.ids_est[s].volume = np.zeros(shape, dtype=np.uint8)
                    self.scores[s].volume = np.zeros(shape, dtype=np.float16)
                self.to_torch(gt=False, scenes=s)

    def remove(self, scene_id):
        self.state[scene_id] = False
        self.scenes_est[scene_id] = None
        self.scenes_gt[scene_id] = None
        self.fusion_weights[scene_id] = None
        if self.semantics:
            self.ids_est[scene_id] = None
            self.scores[scene_id] = None
            if self.semantic_grid:
                self.ids_gt[scene_id] = None

    # ---- post-processing ------------------------------------------------------------------
    def filter(self, value=2.):
        """Outlier filter (database.py:108-112): where weights < value: tsdf = init, weights = 0."""
        for s in self.scenes:
            w = self.fusion_weights[s]
            if _is_dev(w):
                ops.volume_filter(self.scenes_est[s].volume, w, value, self.initial_value)
            else:
                low = w < value
                self.scenes_est[s].volume[low] = self.initial_value
                self.fusion_weights[s][low] = 0

    def filter_semantics(self, value=5):
        """database.py:114-116: median filter of the label volume (on device for the reference's size 5)."""
        for s in self.scenes:
            v = self.ids_est[s].volume
            if _is_dev(v) and value == 5:
                self.ids_est[s].volume = ops.volume_median5(v)
            else:
                from scipy.ndimage import median_filter
                host = v.cpu().numpy() if torch.is_tensor(v) else v
                res = median_filter(host, size=value)
                self.ids_est[s].volume = torch.from_numpy(res).to(v.device) if torch.is_tensor(v) else res

    def evaluate(self, mode='train', workspace=None):
        """database.py:265-309 (note: averages over ALL scenes, untouched ones included, :304)."""
        results, per_scene = {}, {}
        for s in self.scenes:
            if not self.state[s]:
                continue
            est, gt, w = self.scenes_est[s].volume, self.scenes_gt[s].volume, self.fusion_weights[s]
            if _is_dev(est):
                r = ops.volume_evaluate(est, gt, w)
            else:
                r = evaluation(est, gt, w > 0)
            per_scene[s] = r
            for k, v in r.items():
                msg = '{} {}'.format(k, v)
                workspace.log(msg, mode) if workspace is not None else print(msg)
                results[k] = results.get(k, 0) + v
        for k in results:
            results[k] /= len(self.scenes_est.keys())
        return (results, per_scene) if mode == 'test' else results

    def evaluate_semantics(self, mode='train', workspace=None):
        """database.py:311-349.  Device-resident volumes are reduced to the C x C confusion counts by one HIP pass
        (ojf_volume_confusion, 5 B/voxel); only those counts come to the host."""
        results, per_scene = {}, {}

        def host(a):
            return a.detach().cpu().numpy() if torch.is_tensor(a) else a
        for s in self.scenes:
            if not self.state[s]:
                continue
            est, gt, w = self.ids_est[s].volume, self.ids_gt[s].volume, self.fusion_weights[s]
            if _is_dev(est) and _is_dev(w):
                gt_dev = gt if _is_dev(gt) else self._device_volume(gt, torch.uint8)
                hist, e_ids, g_ids = ops.volume_confusion(est, gt_dev, w, self.n_classes)
                n = self.n_classes  # np.bincount(np.unique(x), minlength=n): at least n entries, more if a label >= n occurs
                e_ids = e_ids[:max(n, int(np.flatnonzero(e_ids).max(initial=0)) + 1)]
                g_ids = g_ids[:max(n, int(np.flatnonzero(g_ids).max(initial=0)) + 1)]
                k = max(len(e_ids), len(g_ids))
                e_ids, g_ids = np.pad(e_ids, (0, k - len(e_ids))), np.pad(g_ids, (0, k - len(g_ids)))
                r, cls_iou = semantic_metrics_from_counts(hist, e_ids, g_ids)
            else:
                r, cls_iou = semantic_evaluation(host(est), host(gt), host(w) > 0, self.n_classes)
            per_scene[s] = cls_iou
            for k, v in r.items():
                msg = '{} {}'.format(k, v)
                workspace.log(msg, mode) if workspace is not None else print(msg)
                results[k] = results.get(k, 0) + v
        for k in results:
            results[k] /= len(self.scenes_est.keys())
        return results, per_scene

    # ---- IO: volumes as hdf (or npy), meshes as ply from the HIP marching-tetrahedra kernel (mesh.py) ---
    def _device_volume(self, vol, dtype):
        if not torch.is_tensor(vol):
            vol = torch.from_numpy(np.ascontiguousarray(vol))
        return vol.to(device='cuda', dtype=dtype).contiguous()

    def get_mesh(self, scene_id, semantics=False, palette=None):
        """database.py:118-139: (vertices, faces, normals, rgb) of the zero level set of the estimated volume in
        the reference's mesh frame (voxel index * voxel size, no origin).  Marching tetrahedra instead of skimage's
        marching cubes: same surface, different triangulation (mesh.py)."""
        from . import mesh
        ids = self._device_volume(self.ids_est[scene_id].volume, torch.uint8) if semantics else None
        m = mesh.extract_mesh(self._device_volume(self.scenes_est[scene_id].volume, torch.float16), ids=ids,
                              resolution=float(self.resolution[scene_id]), palette=palette)
        return m['vertices'], m['faces'], m['normals'], m['rgb']

    def save_to_workspace(self, workspace, mode, save_mode='ply'):
        """database.py:141-177: every scene that holds integrated frames goes to the workspace's output directory as
        ``<scene>.tsdf_<mode>.hf5`` / ``.weights_<mode>.hf5`` / ``.semantic_<mode>.hf5`` ('tsdf'), ``<scene>_<mode>.ply``
        ('ply'), or all of them ('test').  ``workspace`` offers save_tsdf_data / save_weights_data /
        save_semantic_data / save_ply_data(file, volume) (utils/setup.py:253-267; drivers.Workspace)."""
        if save_mode not in ('tsdf', 'ply', 'test'):
            return  # the reference's if / elif chain falls through silently
        for s in self.scenes:
            if not self.state[s]:
                continue
            base = s.replace('/', '.')
            if save_mode in ('tsdf', 'test'):
                workspace.save_tsdf_data('{}.tsdf_{}.hf5'.format(base, mode), self.scenes_est[s].volume)
                workspace.save_weights_data('{}.weights_{}.hf5'.format(base, mode), self.fusion_weights[s])
                if self.semantics:
                    workspace.save_semantic_data('{}.semantic_{}.hf5'.format(base, mode), self.ids_est[s].volume)
            if save_mode in ('ply', 'test'):
                workspace.save_ply_data('{}_{}.ply'.format(base, mode), self.scenes_est[s].volume)

    def save(self, path, save_mode='ply', scene_id=None, palette=None):
        """database.py:172-261: 'tsdf' (volumes), 'ply' (mesh), 'test' (volumes + mesh + label-coloured mesh whose
        alpha channel carries the label id)."""
        if scene_id is None:
            raise NotImplementedError
        if save_mode not in ('tsdf', 'ply', 'test'):
            raise ValueError('unknown save_mode {!r}'.format(save_mode))
        base = scene_id.replace('/', '.')
        if save_mode in ('ply', 'test'):
            from . import mesh
            sem = self.semantics and save_mode == 'test'
            ids = self._device_volume(self.ids_est[scene_id].volume, torch.uint8) if sem else None
            m = mesh.extract_mesh(self._device_volume(self.scenes_est[scene_id].volume, torch.float16), ids=ids,
                                  resolution=float(self.resolution[scene_id]), palette=palette)
            mesh.save_ply(os.path.join(path, base + '.ply'), m['vertices'], m['faces'], m['normals'])
            if sem:
                table = np.array(mesh.default_palette() if palette is None else palette, dtype=np.uint8)
                rgba = np.concatenate([table[m['labels']], m['labels'][:, None]], axis=1)
                mesh.save_ply(os.path.join(path, base + '_semantic.ply'), m['vertices'], m['faces'], m['normals'], rgba)
            if save_mode == 'ply':
                return

        def host(a):
            return a.detach().cpu().numpy() if torch.is_tensor(a) else a
        arrays = {'tsdf': ('TSDF', host(self.scenes_est[scene_id].volume)),
                  'weights': ('weights', host(self.fusion_weights[scene_id]))}
        if self.semantics:
            arrays['semantics'] = ('semantics', host(self.ids_est[scene_id].volume))
        from .datasets import save_volume_hdf
        for name, (key, arr) in arrays.items():  # same file / dataset names as database.py:184-201 (npz without h5py)
            save_volume_hdf(os.path.join(path, '{}.{}.hf5'.format(base, name)), key, arr)
