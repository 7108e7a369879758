"""Probe (not a test): does the 4-frame look-ahead leg of bench.py run faster after other legs ran in the same process?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
sync = lambda: torch.cuda.synchronize(dev)
c = dict(h=240, w=320, grid=256, semantics=True, strategy='predict', seg_engine='hip', mode='fast', arith='f16x3', n_classes=30)
def la(L, tag):
    n_la = (60 + L - 1) // L * L
    case = bench.Case(c, dev, 0, 3 * n_la + 4 * L)
    r = bench.run_lookahead(case, n_la, 2 * L, sync, L, 3)
    pf = case.pipe.__dict__.get('_prefetch'); lib = bench._lib.load() if hasattr(bench, '_lib') else __import__('online_joint_depthfusion_and_semantic_amd._lib', fromlist=['x']).load()
    main = torch.cuda.current_stream(dev)
    print(tag, 'L', L, round(r['value'], 1), 'prefetched %d taken %d side stream %x main %x overlap %d' % (pf['n'], pf.get('hits', 0), pf['stream'].cuda_stream, main.cuda_stream, lib.ojf_streams_overlap(main.cuda_stream, pf['stream'].cuda_stream)), flush=True)
    del case; torch.cuda.empty_cache()
def one(tag):
    case = bench.Case(c, dev, 0, 3 * 60 + 10)
    r = case.run(60, 10, sync, profile_frames=16, repeats=3)
    print(tag, 'frame at a time done', flush=True)
    del case; torch.cuda.empty_cache()
la(4, 'fresh process:')
la(8, 'after L=4:')
la(4, 'after L=8:')
one('')
la(4, 'after the frame-at-a-time leg:')
la(8, 'after that:')
