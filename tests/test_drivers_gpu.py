"""-m gpu: the build-owned train / test drivers (counterparts of train_fusion.py / test_fusion.py)."""
import os

import numpy as np
import pytest
import torch

from online_joint_depthfusion_and_semantic_amd.config import default_config
from online_joint_depthfusion_and_semantic_amd.drivers import train_fusion, test_fusion as run_test_fusion, _training_defaults
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticDataset

pytestmark = pytest.mark.gpu


def test_train_then_test_round_trip(cuda, tmp_path):
    h, w, grid = 48, 64, 32
    cfg = _training_defaults(default_config(h, w))
    cfg.SETTINGS.device = str(cuda)
    cfg.TRAINING.optimization.accumulation_steps = 2
    cfg.TRAINING.optimizer.lr = 1e-3
    ds = SyntheticDataset(h, w, grid, 8, scenes=['room_0', 'room_1'])
    pipe, db, losses = train_fusion(cfg, ds, cuda, max_steps=12, checkpoint_dir=str(tmp_path), log=lambda *a: None)
    assert len(losses) == 12 and np.all(np.isfinite(losses))
    assert np.mean(losses[-4:]) < np.mean(losses[:4])  # the fusion net learns on the synthetic stream
    ck = torch.load(os.path.join(str(tmp_path), 'last.pth.tar'), map_location='cpu')
    assert set(ck) == {'epoch', 'model_state', 'optimizer_state', 'scheduler_state'}  # train_fusion.py:245-250
    assert any(db.state.values())
    results, per_scene, db2 = run_test_fusion(cfg, ds, cuda, state_dict=ck['model_state'], log=lambda *a: None)
    assert set(results) == {'mse', 'mad', 'iou', 'acc'} and all(np.isfinite(v) for v in results.values())
    assert set(per_scene) == {'room_0', 'room_1'}
