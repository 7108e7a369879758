"""CPU: the op-for-op torch port timed as cpu_baseline (oracle/torch_port.py) reproduces the golden
volumes of the reference's Pipeline.fuse and agrees bit-for-bit with the C oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_port
from helpers import n_mismatch, golden, net_from_golden, oracle_fuse, fresh_volumes, make_stream


def _tvols(grid):
    v = fresh_volumes(grid, True)
    return {k: torch.from_numpy(a) for k, a in v.items()}


@pytest.mark.parametrize('sem', [True, False])
def test_port_reproduces_reference_pipeline_golden(sem):
    torch.set_num_threads(1)  # golden rule: duplicate-index writes are deterministic single-threaded
    g = golden('pipeline_v3_%s_24x32_g32.npz' % ('sem' if sem else 'nosem'))
    h, w, grid = 24, 32, 32
    net = net_from_golden(g, sem, h, w)
    st = make_stream(h, w, grid)
    vols = _tvols(grid)
    with torch.no_grad():
        for i in range(3):
            torch_port.fuse(st.batch(i), vols, net, torch.from_numpy(st.origin), st.resolution, semantics=True)
            for key in ('tsdf', 'wgt', 'ids', 'scores'):
                assert n_mismatch(vols[key].numpy(), g['f%d_%s' % (i, key)]) == 0, (key, i)


def test_port_equals_c_oracle_config_A():
    torch.set_num_threads(1)
    h, w, grid = 120, 160, 64
    g = golden('pipeline_v3_nosem_24x32_g32.npz')
    net = net_from_golden(g, False, h, w)
    st = make_stream(h, w, grid)
    vols_p, vols_o = _tvols(grid), fresh_volumes(grid, True)
    with torch.no_grad():
        for i in range(2):
            torch_port.fuse(st.batch(i), vols_p, net, torch.from_numpy(st.origin), st.resolution, semantics=True)
            oracle_fuse(st, i, vols_o, net, True)
            for key in ('tsdf', 'wgt', 'ids', 'scores'):
                assert n_mismatch(vols_p[key].numpy(), vols_o[key]) == 0, (key, i)
