"""Profiling helper (not a test): phase stamps of the extract / accumulate / finalize kernels' blocks from a build with -DOJF_EXT_STAMPS /
-DOJF_ACC_STAMPS (OJF_LIB_PATH=.../libojf_stamps.so): where a block of those kernels spends its time."""
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

    def report(fn, n, title, names):
        if not hasattr(lib, fn):
            return
        buf = np.zeros((4096, 8), dtype=np.uint64)
        rc = getattr(lib, fn)(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
        assert rc == 0, rc
        b = buf[:n].astype(np.float64)
        b = b[b[:, 6] > 0]
        wall = (b[:, 7] - b[:, 6]) * 10e-3  # 100 MHz -> us
        last = len(names)
        cyc = b[:, last] - b[:, 0]
        ok = wall > 0.5
        rate = np.median(cyc[ok] / wall[ok]) if ok.any() else 2400.0  # cycles per us
        print('%s: blocks %d: block life %.2f us median (%.2f mean, %.2f max); clock64 %.0f per us' % (title, len(b), np.median(wall), wall.mean(), wall.max(), rate))
        for i, nm in enumerate(names):
            d = (b[:, i + 1] - b[:, i]) / rate
            print('  %-48s median %6.2f us  mean %6.2f  p90 %6.2f' % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
        start = (b[:, 6] - b[:, 6].min()) * 10e-3
        end = (b[:, 7] - b[:, 6].min()) * 10e-3
        print('  block starts: median %.2f us, p90 %.2f, max %.2f after the first; last block ends at %.2f us' % (np.median(start), np.percentile(start, 90), start.max(), end.max()))

    report('ojf_debug_ext_stamps', 1200, 'extract_tile_kernel', ['ray frames (wave 0) + barrier', 'items: corners, gathers (thread 0 = sample 0)', 'barrier', 'transposed stores'])
    report('ojf_debug_acc_stamps', 600, 'integrate_accumulate_tiled_kernel', ['hash clear + ray frames', 'items (corners, hash atomics, head exchanges)', 'barrier', 'record stores', 'tile count'])
    if hasattr(lib, 'ojf_debug_acc_item_stamps'):  # thread 0's two items: loads + ray sample | eight corners | exchange results filed
        buf = np.zeros((4096, 8), dtype=np.uint64)
        assert lib.ojf_debug_acc_item_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
        b = buf[:600].astype(np.float64)
        b = b[(b[:, 0] > 0) & (b[:, 7] > 0)]
        rate = 2330.0
        for it in (0, 1):
            d = [(b[:, 4 * it + j + 1] - b[:, 4 * it + j]) / rate for j in range(3)]
            print('  accumulate thread 0, item %d: loads + ray sample %.2f us | eight corners (hash, claims) %.2f us | exchange results filed %.2f us (medians over %d blocks)'
                  % (it, np.median(d[0]), np.median(d[1]), np.median(d[2]), len(b)))
        print('  between the items: %.2f us' % np.median((b[:, 4] - b[:, 3]) / rate))
    report('ojf_debug_fin_stamps', 600, 'integrate_finalize_kernel', ['first-touch list load', 'head + old values requested', 'record walk', 'volume stores'])


main()
