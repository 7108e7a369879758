"""Kernel time of ojf_mesh_extract on a synthetic room volume (count pass, emit pass, weld), for DESIGN.md."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time
import numpy as np
import torch
from online_joint_depthfusion_and_semantic_amd import mesh, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.arange(n, device='cuda', dtype=torch.float32)
x, y, z = torch.meshgrid(g, g, g, indexing='ij')
c = (n - 1) / 2 + 0.137
room = torch.minimum(torch.minimum(x - 3.3, n - 4.7 - x), torch.minimum(torch.minimum(y - 3.3, n - 4.7 - y), torch.minimum(z - 3.3, n - 4.7 - z)))
ball = torch.sqrt((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2) - n / 5
vol = (torch.minimum(room, ball).clamp(-4, 4) * 0.01).to(torch.float16).contiguous()
del x, y, z, room, ball
tri, _ = mesh.extract_triangles(vol, resolution=0.01)
lib = _lib.load()
count = torch.zeros(1, dtype=torch.int32, device='cuda')
org = np.zeros(3)
st = _lib.stream_ptr(vol.device)
wsb = lib.ojf_mesh_workspace_bytes(n, n, n)
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
for cap, buf in ((0, None), (tri.shape[0], tri)):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        lib.ojf_mesh_extract(vol.data_ptr(), None, None, n, n, n, 0.0, org.ctypes.data, 0.01, ws.data_ptr(), wsb, None if buf is None else buf.data_ptr(), None, None, cap, count.data_ptr(), st)
    ev[0].record()
    for it in range(10):
        lib.ojf_mesh_extract(vol.data_ptr(), None, None, n, n, n, 0.0, org.ctypes.data, 0.01, ws.data_ptr(), wsb, None if buf is None else buf.data_ptr(), None, None, cap, count.data_ptr(), st)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print('n=%d %s pass: %.3f ms, %.1f GB/s of compulsory volume bytes, %d triangles' % (n, 'count+scan' if cap == 0 else 'count+scan+emit', ms, vol.numel() * 2 / ms / 1e6, int(count.item())))
t0 = time.time()
m = mesh.extract_mesh(vol, resolution=0.01)
torch.cuda.synchronize()
print('extract_mesh end to end (count+emit+weld+normals+D2H): %.1f ms, V=%d F=%d' % ((time.time() - t0) * 1e3, m['vertices'].shape[0], m['faces'].shape[0]))
