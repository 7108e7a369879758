"""Dataset adapters either side of the hot path: Replica and ScanNet frame
streams in the reference's on-disk layouts, producing the batch-dict schema ``Pipeline.fuse`` consumes.

Behaviour follows ``dataset/replica.py`` and ``dataset/scannet.py`` (cited per method).  The reference reads
images with OpenCV and grids with h5py; neither is a dependency here:
  * images: Pillow, with OpenCV's conventions restated - 3-channel images are handed out in **BGR** order
    (``cv2.imread``), ``IMREAD_UNCHANGED`` keeps 16-bit depth, nearest-neighbour resizing uses OpenCV's index rule
    ``src = min(floor(dst * src_size / dst_size), src_size - 1)`` (not Pillow's pixel-centre rule);
  * GT grids: ``*.hdf`` through h5py when it is importable, otherwise a ``*.npz`` sibling with the same content
    (``sdf`` [1|2, X, Y, Z], ``bbox`` [3, 2], ``voxel_size``) - ``export_grid_npz`` converts.
PARITY UNPINNED: OpenCV / h5py are absent from the build image, so the reference classes cannot be imported to
generate golden samples; tests/test_datasets.py checks these adapters against the documented conventions on a
synthetic on-disk scene (round trips of the pose conventions, OpenCV's resize rule, unit conversions).
"""
import csv
import glob
import os
import random

import numpy as np

from .database import Voxelgrid

REPLICA_BGR_MEAN = (179.66761167, 179.55742948, 188.2114891)   # dataset/replica.py:241-242
REPLICA_BGR_STD = (12.46442902, 12.55030275, 13.12021586)
SCANNET_BGR_MEAN = (99.09, 113.94, 126.81)                       # dataset/scannet.py:231-232
SCANNET_BGR_STD = (69.64, 71.31, 73.16)
SCANNET_MAIN_IDS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39)  # utils/mapping.py:227-250


# ---- OpenCV-compatible image IO on Pillow ---------------------------------------------------------
def imread(path, unchanged=False):
    """``cv2.imread(path)`` / ``cv2.imread(path, -1)``: uint8 BGR [h,w,3] by default; with ``unchanged`` the
    file's own depth and channel count (16-bit single-channel depth maps stay uint16, colour stays BGR(A))."""
    from PIL import Image
    with Image.open(path) as im:
        if not unchanged:
            return np.asarray(im.convert('RGB'))[:, :, ::-1].copy()
        if im.mode in ('I;16', 'I;16B', 'I;16L', 'I'):
            return np.asarray(im).astype(np.uint16)
        if im.mode in ('L', 'P', '1'):
            return np.asarray(im.convert('L'))
        a = np.asarray(im)
        if a.ndim == 3 and a.shape[2] >= 3:  # RGB(A) -> BGR(A)
            a = np.concatenate([a[:, :, 2::-1], a[:, :, 3:]], axis=2)
        return a.copy()


def resize_nearest(img, width, height):
    """``cv2.resize(img, (width, height), interpolation=cv2.INTER_NEAREST)``."""
    sh, sw = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(height) * (sh / height)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(width) * (sw / width)).astype(np.int64), sw - 1)
    return img[ys][:, xs]


def _frame_number(path):
    return int(os.path.splitext(os.path.basename(path))[0])


def load_sdf_file(path):
    """(sdf [1|2,X,Y,Z], bbox [3,2], voxel_size) from an hdf grid (dataset/replica.py:297-299) or its npz sibling."""
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None and os.path.exists(path):
        with h5py.File(path, 'r') as f:
            return np.array(f['sdf']), np.array(f.attrs['bbox']), float(f.attrs['voxel_size'])
    alt = os.path.splitext(path)[0] + '.npz'
    if os.path.exists(alt):
        z = np.load(alt)
        return z['sdf'], z['bbox'], float(z['voxel_size'])
    raise FileNotFoundError('{}: reading it needs h5py; without h5py provide {} (export_grid_npz)'.format(path, alt))


def export_grid_npz(path, sdf, bbox, voxel_size, labels=None):
    """Writes the npz sibling of ``path`` (an ``*.hdf`` name): ``sdf[0]`` = TSDF, ``sdf[1]`` = labels."""
    vol = np.asarray(sdf, dtype=np.float32)[None]
    if labels is not None:
        vol = np.concatenate([vol, np.asarray(labels, dtype=np.float32)[None]], axis=0)
    out = os.path.splitext(path)[0] + '.npz'
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, sdf=vol, bbox=np.asarray(bbox, dtype=np.float64), voxel_size=float(voxel_size))
    return out


def save_volume_hdf(path, key, data):
    """utils/saving.py:16-39 / modules/database.py:184-201: one gzip-compressed dataset ``key`` ('TSDF' | 'weights' |
    'semantics') per ``*.hf5`` file.  Without h5py the same array goes to ``<path stem>.npz`` under the same key."""
    import torch
    arr = data.detach().cpu().numpy() if torch.is_tensor(data) else np.asarray(data)
    try:
        import h5py
    except ImportError:
        out = os.path.splitext(path)[0] + '.npz'
        np.savez_compressed(out, **{key: arr})
        return out
    with h5py.File(path, 'w') as f:
        f.create_dataset(key, shape=arr.shape, data=arr, compression='gzip', compression_opts=9)
    return path


class ToTensor:
    """utils/transform.py:5-30: ndarrays -> tensors, the image HWC -> CHW."""

    def __call__(self, sample):
        import torch
        out = {}
        for k, v in sample.items():
            if isinstance(v, np.ndarray):
                out[k] = torch.from_numpy(np.ascontiguousarray(v.transpose(2, 0, 1) if k == 'image' else v))
            else:
                out[k] = v
        return out


class _FrameDataset:
    """Path bookkeeping shared by the two adapters: scene-list files, the 'hybrid' scene interleaving
    (dataset/replica.py:55-133) and per-modality path lists sorted by frame number."""

    scene_field = 0  # which '/'-separated field of a list entry names the scene

    def __init__(self, config_data):
        c = config_data
        self.root_dir = c.root_dir
        self.resolution = (c.resy, c.resx)  # rows, columns
        self.pad = c.get('pad', 0)
        self.augmentations = c.get('augmentations', None)
        self.normalize = c.get('normalize', False)
        self.transform = c.get('transform', None)
        self.frame_ratio = c.get('frame_ratio', 1)
        self.scene_list = c.scene_list
        self.input = c.input
        self.target = c.get('target', None)
        self.semantics = c.get('semantics', None)
        self.mode = c.get('mode', 'test')
        self.truncation_strategy = c.get('truncation_strategy', 'standard')
        self._scenes = []
        self.scenedir = None
        if c.get('data_load_strategy', 'max_depth_diversity') == 'hybrid':
            self._hybrid_order(c.load_scenes_at_once)

    def _list_path(self):
        return self.scene_list if os.path.isabs(self.scene_list) or os.path.exists(self.scene_list) \
            else os.path.join(self.root_dir, self.scene_list)

    def _list_lines(self):
        with open(self._list_path()) as f:
            return [ln.rstrip('\n').split(' ') for ln in f if ln.strip()]

    def _scene_of(self, entry):
        return entry.split('/')[self.scene_field]

    def _from_list(self, position):
        """All files of column ``position`` of every list line, every ``frame_ratio``-th, by frame number
        (dataset/replica.py:136-159)."""
        paths = []
        for cols in self._list_lines():
            scene = self._scene_of(cols[0])
            if scene not in self._scenes:
                self._scenes.append(scene)
            files = sorted(glob.glob(os.path.join(self.root_dir, cols[position], '*')), key=_frame_number)
            paths.extend(files)
        paths = paths[::self.frame_ratio]
        paths.sort(key=_frame_number)  # stable: equal frame numbers keep list order
        return paths

    def _interleave(self, per_key):
        """Round-robin over the per-loader lists (dataset/replica.py:120-131)."""
        out, longest = [], max(len(v) for v in per_key.values())
        for i in range(longest):
            for k in per_key:
                if i < len(per_key[k]):
                    out.append(per_key[k][i])
        return out

    @property
    def scenes(self):
        return self._scenes

    def __len__(self):
        return len(self.color_images)

    def scene_of_item(self, item):
        """Scene of frame ``item`` without decoding it (scene-sharded loaders)."""
        return self.color_images[item].split('/')[-4 if self.scene_field == 0 else -3]

    def _depth_sample(self, path):
        d = imread(path, unchanged=True)
        return (resize_nearest(d, self.resolution[1], self.resolution[0]) / 1000.0)  # millimetres -> metres

    def _grid_from_sdf(self, file, truncation, with_labels):
        """Truncation, free-space label reset, padding and bounding box exactly as dataset/replica.py:297-327 /
        dataset/scannet.py:276-305."""
        sdf, box, voxel_size = load_sdf_file(file)
        voxels = np.array(sdf[0]).astype(np.float16)
        if self.truncation_strategy == 'artificial':
            voxels[np.abs(voxels) >= truncation] = truncation
        elif self.truncation_strategy == 'standard':
            voxels[voxels > truncation] = truncation
            voxels[voxels < -truncation] = -truncation
        labels = None
        if with_labels:
            labels = np.array(sdf[1]).astype(np.uint8)
            labels[voxels > truncation] = 0
            labels[voxels < -truncation] = 0
        voxels = np.pad(voxels, self.pad, 'constant', constant_values=-truncation)
        bbox = np.zeros((3, 2))
        bbox[:, 0] = box[:, 0] - self.pad * voxel_size
        bbox[:, 1] = bbox[:, 0] + voxel_size * np.array(voxels.shape)
        grid = Voxelgrid(voxel_size)
        grid.from_array(voxels, bbox)
        if labels is None:
            return (grid,)
        sem = Voxelgrid(voxel_size)
        sem.from_array(np.pad(labels, self.pad, 'constant', constant_values=0), bbox)
        return (grid, sem)


class Replica(_FrameDataset):
    """dataset/replica.py.  Scene-list columns: ``[0]`` GT depth dir, ``[1]`` noisy (ToF) depth dir, ``[-3]`` colour,
    ``[-2]`` camera matrices, ``[-1]`` semantic images; directories are ``<scene>/<trajectory>/<modality>``."""

    scene_field = 0

    def __init__(self, config_data):
        super().__init__(config_data)
        hy = self.scenedir is not None
        self.color_images = self._hybrid('left_rgb') if hy else self._from_list(-3)
        self.cameras = self._hybrid('left_camera_matrix') if hy else self._from_list(-2)
        if self.input != 'image':
            if self.input == 'tof_depth':
                self.depth_images = self._hybrid('left_depth_noise_5.0') if hy else self._from_list(1)
            elif self.input == 'depth_gt':
                self.depth_images = self._hybrid('left_depth_gt') if hy else self._from_list(0)
            else:
                raise NotImplementedError(self.input)
        if self.target == 'depth_gt':
            self.depth_images_gt = self._hybrid('left_depth_gt') if hy else self._from_list(0)
        if self.semantics == 'class30':
            self.semantic_images_gt = self._hybrid('left_' + self.semantics) if hy else self._from_list(-1)

    def _hybrid_order(self, n_loaders):
        """dataset/replica.py:55-101: scenes are dealt to ``n_loaders`` lists, three trajectories each, shuffled."""
        scenes = []
        for cols in self._list_lines():
            s = self._scene_of(cols[0])
            if s not in scenes:
                scenes.append(s)
        self._scenes = list(scenes)
        if n_loaders > len(scenes):
            raise ValueError('load_scenes_at_once exceeds the number of scenes')
        dealt = {i: [] for i in range(n_loaders)}
        while scenes:
            pick = random.sample(range(len(scenes)), min(n_loaders, len(scenes)))
            for key, idx in enumerate(pick):
                dealt[key].append(scenes[idx])
            scenes = [s for i, s in enumerate(scenes) if i not in pick]
        for key in dealt:
            trajectories = ['{}/{}'.format(s, t + 1) for s in dealt[key] for t in range(3)]
            random.shuffle(trajectories)
            dealt[key] = trajectories
        self.scenedir = dealt

    def _hybrid(self, modality):
        per_key = {}
        for key, trajectories in self.scenedir.items():
            files = []
            for tr in trajectories:
                files.extend(sorted(glob.glob(os.path.join(self.root_dir, tr, modality, '*')), key=_frame_number))
            per_key[key] = files
        paths = self._interleave(per_key)
        return paths[::self.frame_ratio] if self.mode in ('val', 'test') else paths

    def __getitem__(self, item):
        """dataset/replica.py:211-294."""
        sample = {'item_id': item}
        file = self.color_images[item]
        parts = file.split('/')
        sample['frame_id'] = '{}/{}/{}'.format(parts[-4], parts[-3], os.path.splitext(parts[-1])[0])
        rows, cols = self.resolution
        image = resize_nearest(imread(file), cols, rows)
        if self.semantics:
            semantic = resize_nearest(imread(self.semantic_images_gt[item], unchanged=True)[:, :, 0], cols, rows)
            if self.augmentations is not None:
                image, semantic = self.augmentations(image, semantic)
            sample['semantic_gt'] = semantic.astype(np.uint8)
        if self.normalize:
            image = (image - np.array(REPLICA_BGR_MEAN)) / np.array(REPLICA_BGR_STD)
        sample['image'] = image.astype(np.float32)
        if self.input in ('tof_depth', 'depth_gt'):
            depth = self._depth_sample(self.depth_images[item])
            sample[self.input] = depth.astype(np.float32)
            sample['mask'] = (depth > 0.05) & (depth < 5.0)
        if self.target == 'depth_gt':
            sample[self.target] = self._depth_sample(self.depth_images_gt[item]).astype(np.float32)
        sample['extrinsics'] = self.pose_from_file(np.loadtxt(self.cameras[item]))
        # 90 degree field of view; the reference derives BOTH the focal length and the principal point from the
        # row count (replica.py:281-290), i.e. it assumes square frames
        f = rows / 2.0 * (1.0 / np.tan(np.deg2rad(90.0) / 2))
        shift = rows / 2
        sample['intrinsics'] = np.asarray([[f, 0.0, shift], [0.0, f, shift], [0.0, 0.0, 1.0]])
        return self.transform(sample) if self.transform else sample

    @staticmethod
    def pose_from_file(matrix):
        """The camera-to-world [3,4] float64 pose the fusion code expects (z forward, y down, x right) from a
        Replica camera-matrix file, step by step as dataset/replica.py:266-279."""
        ext = np.linalg.inv(matrix).astype(np.float32)
        r_y = np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]], dtype=np.float32)
        r_z = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]], dtype=np.float32)
        r_x90 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float32)
        ext = np.matmul(np.matmul(r_z, r_y), ext[0:3, 0:4])
        ext = np.linalg.inv(np.concatenate((ext, np.array([[0, 0, 0, 1]])), axis=0))
        return np.matmul(r_x90, ext[0:3, 0:4])

    @staticmethod
    def file_from_pose(pose):
        """Inverse of ``pose_from_file`` (writes synthetic scenes in the Replica convention)."""
        r_y = np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]], dtype=np.float64)
        r_z = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]], dtype=np.float64)
        r_x90 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)
        c = np.concatenate((np.linalg.inv(r_x90) @ np.asarray(pose, dtype=np.float64)[0:3, 0:4], [[0, 0, 0, 1]]), axis=0)
        b = np.linalg.inv(c)[0:3, 0:4]
        a = np.concatenate((np.linalg.inv(r_z @ r_y) @ b, [[0, 0, 0, 1]]), axis=0)
        return np.linalg.inv(a)

    def get_grid(self, scene, truncation, semantic_grid=True):
        name = 'semantic_sdf.hdf' if self.semantics else 'sdf.hdf'
        return self._grid_from_sdf(os.path.join(self.root_dir, scene, 'gt_semantic_sdf', name), truncation,
                                   bool(self.semantics))


class ScanNet(_FrameDataset):
    """dataset/scannet.py.  Scene-list columns: ``[0]`` depth dir, ``[1]`` colour, ``[2]`` label-filt, ``[3]`` pose,
    ``[-1]`` the scan directory holding ``intrinsic_depth.txt``; entries look like ``scans/<scene>/<modality>``.
    ``semantics: nyu40 | nyu20`` needs ScanNet's ``scannetv2-labels.combined.tsv`` (``labels_tsv`` in the config,
    default ``<root_dir>/scannetv2-labels.combined.tsv``; the reference hard-codes a cluster path, mapping.py:254)."""

    scene_field = 1

    def __init__(self, config_data):
        super().__init__(config_data)
        hy = self.scenedir is not None
        self.color_images = self._hybrid('color') if hy else self._from_list(1)
        self.cameras = self._hybrid('pose') if hy else self._from_list(3)
        self.intrinsics = {}
        rows, cols = self.resolution
        for line in self._list_lines():  # dataset/scannet.py:169-181
            k_file = np.loadtxt(os.path.join(self.root_dir, line[-1], 'intrinsic_depth.txt'))
            kx, ky = cols / 640, rows / 480
            k = np.array([[kx, 0, kx], [0, ky, ky], [0, 0, 1]]).astype(np.float32)
            self.intrinsics[self._scene_of(line[0])] = np.matmul(k, k_file[0:3, 0:3])
        if self.input != 'image':
            self.depth_images = self._hybrid('depth') if hy else self._from_list(0)
        if self.target == 'depth_gt':
            self.depth_images_gt = self._hybrid('depth') if hy else self._from_list(0)
        self.label_map = None
        if self.semantics in ('nyu40', 'nyu20'):
            tsv = config_data.get('labels_tsv', os.path.join(self.root_dir, 'scannetv2-labels.combined.tsv'))
            self.label_map = self.nyu_label_map(tsv, self.semantics)
            self.semantic_images_gt = self._hybrid('label-filt') if hy else self._from_list(2)

    @staticmethod
    def nyu_label_map(tsv_path, kind):
        """utils/mapping.py:252-280: raw ScanNet id -> NYU40 id (column 4 of the tsv), or its 21-class subset index."""
        with open(tsv_path) as f:
            rows = list(csv.reader(f, delimiter='\t'))[1:]
        m40 = {int(r[0]): int(r[4]) for r in rows}
        m40[0] = 0
        if kind == 'nyu40':
            return m40
        return {k: SCANNET_MAIN_IDS.index(v if v in SCANNET_MAIN_IDS else 0) for k, v in m40.items()}

    def _hybrid_order(self, n_loaders):
        """dataset/scannet.py:69-97: physical rooms (``scene0000``) are dealt to the loaders with all their scans."""
        scenes = []
        for cols in self._list_lines():
            s = self._scene_of(cols[0])
            if s not in scenes:
                scenes.append(s)
        self._scenes = list(scenes)
        rooms = {}
        for s in scenes:
            rooms.setdefault(s.split('_')[0], []).append(s.split('_')[1])
        if n_loaders > len(rooms):
            raise ValueError('load_scenes_at_once exceeds the number of rooms')
        dealt = {k: [] for k in range(n_loaders)}
        while rooms:
            pick = random.sample(range(len(rooms)), min(len(rooms), n_loaders))
            names = list(rooms.keys())
            for key, idx in enumerate(pick):
                dealt[key].extend('{}_{}'.format(names[idx], t) for t in rooms[names[idx]])
            rooms = {k: v for i, (k, v) in enumerate(rooms.items()) if i not in pick}
        self.scenedir = {k: random.sample(v, len(v)) for k, v in dealt.items()}

    def _hybrid(self, modality):
        per_key = {}
        for key, scans in self.scenedir.items():
            files = []
            for scan in scans:
                files.extend(sorted(glob.glob(os.path.join(self.root_dir, scan, modality, '*')), key=_frame_number))
            per_key[key] = files
        return self._interleave(per_key)[::self.frame_ratio]

    def __getitem__(self, item):
        """dataset/scannet.py:194-263."""
        sample = {'item_id': item}
        file = self.color_images[item]
        parts = file.split('/')
        scene = parts[-3]
        sample['frame_id'] = '{}/{}'.format(scene, os.path.splitext(parts[-1])[0])
        rows, cols = self.resolution
        image = resize_nearest(imread(file), cols, rows)
        if self.semantics:
            semantic = resize_nearest(imread(self.semantic_images_gt[item], unchanged=True), cols, rows)
            if self.augmentations is not None:
                image, semantic = self.augmentations(image, semantic)
            semantic = np.array([self.label_map[int(s)] for s in semantic.flatten()]).reshape(self.resolution)
            sample['semantic_gt'] = semantic.astype(np.uint8)
        if self.normalize:
            image = (image - np.array(SCANNET_BGR_MEAN)) / np.array(SCANNET_BGR_STD)
        sample['image'] = image.astype(np.float32)
        if self.input == 'depth_gt':
            depth = self._depth_sample(self.depth_images[item])
            sample[self.input] = depth.astype(np.float32)
            sample['mask'] = depth > 0.01
        if self.target == 'depth_gt':
            sample[self.target] = self._depth_sample(self.depth_images_gt[item]).astype(np.float32)
        sample['extrinsics'] = np.loadtxt(self.cameras[item]).astype(np.float32)  # 4x4 camera-to-world as stored
        sample['intrinsics'] = self.intrinsics[scene]
        return self.transform(sample) if self.transform else sample

    def get_grid(self, scene, truncation, semantic_grid=True):
        file = os.path.join(self.root_dir, 'scans', scene, scene + '_sdf.hdf')
        if not (os.path.exists(file) or os.path.exists(os.path.splitext(file)[0] + '.npz')):
            file = file.replace('scans', 'scans_test', 1)
        return self._grid_from_sdf(file, truncation, bool(semantic_grid))
