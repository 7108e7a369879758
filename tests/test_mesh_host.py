"""Host side of the mesh export (mesh.py): PLY writer/reader round trip and the palette; no GPU needed."""
import numpy as np
import pytest

from online_joint_depthfusion_and_semantic_amd import mesh


def test_ply_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    v = rng.normal(size=(17, 3)).astype(np.float32)
    f = rng.integers(0, 17, size=(29, 3)).astype(np.int32)
    n = rng.normal(size=(17, 3)).astype(np.float32)
    c = rng.integers(0, 256, size=(17, 4)).astype(np.uint8)
    for normals, rgba in ((None, None), (n, None), (n, c), (None, c)):
        path = str(tmp_path / 'm.ply')
        mesh.save_ply(path, v, f, normals, rgba)
        back = mesh.load_ply(path)
        assert np.array_equal(back['vertices'], v) and np.array_equal(back['faces'], f)
        assert (back['normals'] is None) == (normals is None) and (back['rgba'] is None) == (rgba is None)
        if normals is not None:
            assert np.array_equal(back['normals'], n)
        if rgba is not None:
            assert np.array_equal(back['rgba'], c)
    head = open(path, 'rb').read(200).decode('latin1')
    assert head.startswith('ply\nformat binary_little_endian 1.0\n') and 'property list uchar int vertex_indices' in open(path, 'rb').read(600).decode('latin1')
    mesh.save_ply(path, np.zeros((0, 3)), np.zeros((0, 3), int))
    assert mesh.load_ply(path)['vertices'].shape == (0, 3)


def test_palette_and_gpu_requirement():
    p = mesh.default_palette()
    assert p.shape == (256, 3) and p.dtype == np.uint8 and (p[0] == 0).all()
    assert np.unique(p, axis=0).shape[0] == 256  # labels stay distinguishable
    import torch
    if not torch.cuda.is_available():  # the product path never falls back to a CPU mesher
        with pytest.raises(RuntimeError):
            mesh.extract_triangles(torch.zeros((4, 4, 4), dtype=torch.float16))
