"""-m gpu: full-grid Database passes on device against their host definitions (bit-exact for u8/fp16 work)."""
import numpy as np
import pytest
import torch
from scipy.ndimage import median_filter

from online_joint_depthfusion_and_semantic_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(32, 32, 32), (17, 9, 30), (64, 40, 8), (5, 5, 5), (3, 2, 1)])
def test_median5_matches_scipy(cuda, shape):
    rng = np.random.default_rng(sum(shape))
    # blobs of labels plus salt noise: the regime filter_semantics is used in
    vol = (rng.integers(0, 40, size=[(s + 3) // 4 for s in shape]).repeat(4, 0).repeat(4, 1).repeat(4, 2))[:shape[0], :shape[1], :shape[2]]
    vol = np.where(rng.random(shape) < 0.2, rng.integers(0, 256, size=shape), vol).astype(np.uint8)
    vol = np.ascontiguousarray(vol)
    want = median_filter(vol, size=5)  # modules/database.py:116
    got = ops.volume_median5(torch.from_numpy(vol).to(cuda)).cpu().numpy()
    assert np.array_equal(got, want)


def test_fill_filter_known_answers(cuda):
    t = torch.empty((9, 7, 13), dtype=torch.float16, device=cuda)
    ops.volume_fill(t, 0.1)
    assert torch.all(t == torch.tensor(0.1, dtype=torch.float16))
    w = torch.tensor(np.linspace(0, 4, 9 * 7 * 13).reshape(9, 7, 13).astype(np.float16)).to(cuda)
    w0 = w.clone()
    t.fill_(-0.02)
    ops.volume_filter(t, w, 2.0, 0.1)
    low = w0 < 2.0
    assert torch.all(t[low] == torch.tensor(0.1, dtype=torch.float16)) and torch.all(w[low] == 0)
    assert torch.all(t[~low] == torch.tensor(-0.02, dtype=torch.float16)) and torch.equal(w[~low], w0[~low])
    u = torch.ones((5, 5, 5), dtype=torch.uint8, device=cuda)
    ops.volume_fill(u, 0)
    assert int(u.sum()) == 0
