"""Profiling helper (not a test): phase stamps of the accumulate kernel's blocks from a build with -DOJF_ACC_STAMPS
(OJF_LIB_PATH=.../libojf_stamps.so): where a block of integrate_accumulate_tiled_kernel spends its time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from online_joint_depthfusion_and_semantic_amd import _lib

def main():
    dev = torch.device('cuda:0')
    h, w, grid = 240, 320, 256
    c = dict(h=h, w=w, grid=grid, semantics=False, mode='fast', n_classes=30, arith='f16x3', strategy='gt', seg_engine='hip')
    case = bench.Case(c, dev, 0, 40)
    with torch.no_grad():
        for i in range(30):
            case.fuse(i)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros((4096, 8), dtype=np.uint64)
    rc = lib.ojf_debug_acc_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    assert rc == 0, rc
    n = 600
    b = buf[:n].astype(np.float64)
    wall = (b[:, 7] - b[:, 6]) * 10e-3  # 100 MHz -> us
    cyc = b[:, 5] - b[:, 0]
    rate = np.median(cyc / np.maximum(wall, 1e-9))  # cycles per us
    names = ['hash clear + ray frames', 'items (corners, hash atomics)', 'number the records', 'head exchanges + record stores', 'first-touch list']
    print('blocks %d: block life %.2f us median (%.2f mean, %.2f max); clock64 %.0f per us' % (n, np.median(wall), wall.mean(), wall.max(), rate))
    for i, nm in enumerate(names):
        d = (b[:, i + 1] - b[:, i]) / rate
        print('  %-34s median %6.2f us  mean %6.2f  p90 %6.2f' % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
    start = (b[:, 6] - b[:, 6].min()) * 10e-3
    end = (b[:, 7] - b[:, 6].min()) * 10e-3
    print('  block starts: median %.2f us, p90 %.2f, max %.2f after the first; last block ends at %.2f us' % (np.median(start), np.percentile(start, 90), start.max(), end.max()))

main()
