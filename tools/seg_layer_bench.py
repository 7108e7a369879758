"""Profiling helper (not a test): ONE SegConv layer shape in a loop (python tools/seg_layer_bench.py c_in c_out k h w batch [reps [copies]]);
kernel form / ablations by the OJF_SEG_* switches.  The loop is a replayed device graph of `reps` dependent launches; with copies > 1 the
launches walk through that many separately allocated weight sets (copies x weights > L2 + MALL: every launch starts on cold weights,
as a layer does inside a frame), with copies = 1 they hit the weights the previous launch left in the caches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd.segconv import SegConv, nhwc
cin, cout, k, h, w, B = [int(x) for x in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 64
copies = int(sys.argv[8]) if len(sys.argv) > 8 else 1
torch.manual_seed(0)
conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False).cuda()
ops = [SegConv(conv) for _ in range(copies)]
x = nhwc(cin, h, w, 'cuda', batch=B); x.normal_()
out = nhwc(cout, h, w, 'cuda', batch=B)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for i in range(copies + 2): ops[i % copies](x, out=out, act='relu')
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        for i in range(reps): ops[i % copies](x, out=out, act='relu')
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
us = (time.perf_counter() - t0) / (5 * reps) * 1e6
import hashlib
with torch.no_grad():
    ref = torch.relu(torch.nn.functional.conv2d(x[:, :cin].contiguous().double(), conv.weight.double(), padding=k // 2))
    err = float((out[:, :cout].double() - ref).abs().max() / ref.abs().max())
sha = hashlib.sha256(out.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
print('%4d -> %4d k%d %dx%d B%d: %.1f us per launch, out sha %s rel err %.1e (graph of %d, %d weight set(s) = %.0f MB, %s)' % (
    cin, cout, k, h, w, B, us, sha, err, reps, copies, copies * cin * cout * k * k * 4 / 1e6,
    ' '.join('%s=%s' % (a, b) for a, b in sorted(os.environ.items()) if a.startswith('OJF_SEG'))))
