#!/usr/bin/env python
"""frames/sec fused: the per-frame hot path (extract -> FusionNet_v3 -> integrate) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one Pipeline.fuse call: one 320x240 synthetic depth frame fused into a 256^3 fp16
volume resident in HBM (BASELINE.json configs[1], geometry-only, FusionNet_v3 without semantics).
Frames (depth, mask) are resident in HBM before the timed region; poses stay on the host like any
small per-frame metadata.  With N > 1 every rank owns one scene (its volumes + its own frame stream,
SURVEY.md §8e): no data-path collective, weak scaling, value = N*K / max-over-ranks time.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline      dominant kernel (conv_mfma_kernel, fp32 MFMA): algorithmic flops / live HIP-event time
  roofline_hbm  extract + integrate against the HBM roofline (algorithmic bytes of SURVEY.md §8d)
  cpu_baseline  the op-for-op torch-CPU port of the reference path timed on this node's host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from online_joint_depthfusion_and_semantic_amd.config import default_config, database_config  # noqa: E402
from online_joint_depthfusion_and_semantic_amd.database import Database  # noqa: E402
from online_joint_depthfusion_and_semantic_amd.pipeline import Pipeline  # noqa: E402
from online_joint_depthfusion_and_semantic_amd.synthetic import SyntheticStream  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3  # dense fp32-input MFMA peak (v_mfma_f32_16x16x4_f32)
F16_MFMA_PEAK_TF = 2516.6  # dense fp16 MFMA peak (v_mfma_f32_16x16x32_f16)
# split-fp16 arithmetic issues three fp16 MFMAs per product block: the rate at which it can retire
# fp32-equivalent (useful) flops is a third of the fp16 peak
ARITH = {'f32': ('f32', 'f32-input MFMA (v_mfma_f32_16x16x4_f32)', F32_MFMA_PEAK_TF),
         'f16x3': ('f16x3', 'split-fp16: 3 x v_mfma_f32_16x16x32_f16 per product block, fp32 accumulate, fp32 activations',
                   F16_MFMA_PEAK_TF / 3.0)}


def seeded_weights(pipe, seed=1911):
    torch.manual_seed(seed)
    for m in pipe._fusion_network.modules():  # train_fusion.py:29-31 + non-trivial BN statistics
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.xavier_normal_(m.weight)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)


def cpu_baseline(args, h, w, grid, semantics):
    """Times oracle/torch_port.fuse (the reference's op sequence on torch-CPU) on a bounded sample."""
    from oracle import torch_port  # checker / baseline only - never on the product path
    from online_joint_depthfusion_and_semantic_amd.model import FusionNet_v3
    cfg = default_config(h, w, semantics=semantics)
    cfg.FUSION_MODEL.resx, cfg.FUSION_MODEL.resy = w, h
    net = FusionNet_v3(cfg.FUSION_MODEL).eval()
    st = SyntheticStream(h, w, grid, 40, scene='cpu_scene')
    vols = dict(tsdf=torch.full((grid,) * 3, 0.1, dtype=torch.float16), wgt=torch.zeros((grid,) * 3, dtype=torch.float16))
    if semantics:
        vols.update(ids=torch.zeros((grid,) * 3, dtype=torch.uint8), scores=torch.zeros((grid,) * 3, dtype=torch.float16))
    origin = torch.from_numpy(st.origin)
    times = []
    with torch.no_grad():
        for i in range(args.cpu_frames + 1):
            b = st.batch(i)
            t0 = time.perf_counter()
            torch_port.fuse(b, vols, net, origin, st.resolution, semantics=semantics)
            times.append(time.perf_counter() - t0)
    t = float(np.mean(times[1:]))
    return {'value': 1.0 / t, 'unit': 'frames/sec', 'cores': int(torch.get_num_threads()), 'kind': 'port',
            'sample': '%d frames of the %dx%d -> %d^3 workload after 1 warm-up frame, torch-CPU op-for-op port of '
                      'the reference (oracle/torch_port.py), %.2f s/frame' % (args.cpu_frames, w, h, grid, t)}


def pmc_traffic(h, w, grid, semantics):
    """HBM bytes per frame per kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    collected in separate runs of this bench, tools/pmc_traffic.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950).  Only valid for the
    default workload it was measured on; otherwise None."""
    path = os.path.join(ROOT, 'profiles', 'r01_traffic_pmc.json')
    if (h, w, grid, semantics) != (240, 320, 256, False) or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def algorithmic_bytes(st, frames, n_points, n_tail, semantics):
    """Compulsory HBM bytes of extract + integrate per frame, averaged over a sample."""
    from oracle import oracle  # only to COUNT distinct voxels on the host; not part of the timed path
    N = st.h * st.w
    tot, ug_s, us_s = 0.0, 0, 0
    for i in frames:
        f = st.frame(i)
        depth = f[st.depth_key]
        fd = np.where(f['mask'], depth, np.float32(0)).astype(np.float32)
        Ki, E = oracle.camera_arrays(f['intrinsics'], f['extrinsics'])
        ug, us = oracle.unique_voxels(depth, fd, Ki, E, st.origin, st.resolution, (st.grid,) * 3, n_points, n_tail)
        b = 4 * N + 4 * ug + 8 * n_points * N + 4 * n_tail * N + 8 * us
        if semantics:
            b += 6 * us + 5 * N
        tot += b
        ug_s += ug
        us_s += us
    k = len(frames)
    return tot / k, ug_s / k, us_s / k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--height', type=int, default=240)
    ap.add_argument('--width', type=int, default=320)
    ap.add_argument('--grid', type=int, default=256)
    ap.add_argument('--semantics', action='store_true', help='BASELINE configs[2]-style: gt labels + semantic head')
    ap.add_argument('--semantic-strategy', default='gt', choices=['gt', 'predict'],
                    help="with --semantics: 'predict' runs AdapNet++ (random init) on every frame")
    ap.add_argument('--seg-engine', default='hip', choices=['hip', 'torch'],
                    help="AdapNet++ convolutions: 'hip' = SEGCONV MFMA kernels (default), 'torch' = module forward on MIOpen")
    ap.add_argument('--mode', default='fast', choices=['fast', 'parity'])
    ap.add_argument('--arith', default='f16x3', choices=['f16x3', 'f32'], help='net MFMA arithmetic (include/ojf.h OJF_ARITH_*)')
    ap.add_argument('--cpu-frames', type=int, default=4, help='timed frames of the CPU baseline (0 = skip)')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL, default) | gloo (validation of the N>1 path on a 1-GPU box)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        # RCCL; used only for the barrier + max-reduce of the time (no data-path collective)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    h, w, grid = args.height, args.width, args.grid
    cfg = default_config(h, w, semantics=args.semantics, integrate_mode=args.mode)
    cfg.SETTINGS.device = str(dev)
    cfg.FUSION_MODEL.arithmetic = args.arith
    if args.semantics:
        cfg.DATA.semantic_strategy = args.semantic_strategy
        cfg.SEMANTIC_2D_MODEL.engine = args.seg_engine
    n_frames = args.steps + args.warmup
    st = SyntheticStream(h, w, grid, n_frames, scene='room_%d' % rank, seed=1911 + rank)
    db = Database(st, database_config(cfg))
    pipe = Pipeline(cfg)
    seeded_weights(pipe)
    pipe = pipe.to(dev).eval()
    pipe.profile = 8  # stage events on every 8th frame of the timed region: a record is a marker packet (~4.5 us of idle queue)

    # frames resident in HBM before the clock starts; poses and ids stay host-side metadata
    batches = []
    image = torch.zeros((1, 3, h, w), device=dev)  # only its shape is read on this path
    predict = args.semantics and args.semantic_strategy == 'predict'
    for i in range(n_frames):
        f = st.frame(i)
        b = {'image': torch.from_numpy(f['image']).unsqueeze(0).to(dev) if predict else image, 'frame_id': [f['frame_id']],
             st.depth_key: torch.from_numpy(f[st.depth_key]).unsqueeze(0).to(dev),
             'mask': torch.from_numpy(f['mask']).unsqueeze(0).to(dev),
             'extrinsics': torch.from_numpy(f['extrinsics']).unsqueeze(0),
             'intrinsics': torch.from_numpy(f['intrinsics']).unsqueeze(0)}
        if args.semantics:
            b['semantic_gt'] = torch.from_numpy(f['semantic_gt']).unsqueeze(0).to(dev)
        batches.append(b)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for i in range(args.warmup):
            pipe.fuse(batches[i], db, dev)
        pipe.reset_profile()
        sync()
        t0 = time.perf_counter()
        for i in range(args.warmup, n_frames):
            pipe.fuse(batches[i], db, dev)
        sync()
        elapsed = time.perf_counter() - t0
    pipe.check()  # outside the timed region: raises if the split-fp16 range guard fired on any frame

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    stages = pipe.stage_times_ms()  # live HIP events recorded on the launch stream inside the timed region

    if rank == 0:
        fps = world * args.steps / elapsed
        P, T = cfg.FUSION_MODEL.n_points, cfg.FUSION_MODEL.n_tail_points
        N = h * w
        flops = 2.0 * pipe._engine.macs_per_pixel * N  # useful flops, channel padding excluded
        net_s = stages['net'] / 1e3
        n_conv = pipe._engine.conv_launches
        sample = list(range(args.warmup, n_frames, max(1, args.steps // 4)))[:4]
        bytes_frame, ug, us = algorithmic_bytes(st, sample, P, T, args.semantics)
        ei_s = (stages['extract'] + stages['integrate']) / 1e3
        tr = pmc_traffic(h, w, grid, args.semantics)
        net_traffic = hbm_traffic = None
        if tr:
            net_k = [k for k in tr if k.startswith(('conv_mfma', 'conv_f16x3', 'chain1x1', 'vortex_tail'))]
            net_traffic = sum(tr[k]['fetch_bytes_per_frame_x2'] + tr[k]['write_bytes_per_frame'] for k in net_k) / n_conv
            hbm_traffic = sum(tr[k]['fetch_bytes_per_frame_x2'] + tr[k]['write_bytes_per_frame']
                              for k in tr if 'extract' in k or 'integrate' in k)
        out = {
            'metric': 'frames/sec fused (%dx%d, %d^3 grid)' % (w, h, grid), 'value': fps, 'unit': 'frames/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': ARITH[args.arith][0], 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: geometry-only fusion, %dx%d depth into a %d^3 fp16 TSDF grid, '
                                   'FusionNet_v3%s, one scene per GPU' % (w, h, grid, (' + %s semantics' % ('AdapNet++ (predict, %s convolutions)' % args.seg_engine if predict else 'gt')) if args.semantics else ''),
                       'frame': [h, w], 'grid': grid, 'n_points': P, 'n_tail_points': T, 'integrate_mode': args.mode,
                       'volume_dtype': 'f16', 'net_arithmetic': ARITH[args.arith][1], 'parallelism': 'scene-sharded x%d' % world},
            'stages_ms': stages,
            'roofline': {'bound': 'mfma', 'kernel': 'conv_f16x3_kernel' if args.arith == 'f16x3' else 'conv_mfma_kernel',
                         'launches_per_frame': n_conv,
                         'achieved': flops / net_s / 1e12, 'peak': ARITH[args.arith][2], 'unit': 'TFLOP/s',
                         'frac': flops / net_s / 1e12 / ARITH[args.arith][2], 'traffic': net_traffic,
                         'flops_per_frame': flops, 'avg_launch_us': 1e6 * net_s / n_conv,
                         'frac_of_f32_mfma_peak': flops / net_s / 1e12 / F32_MFMA_PEAK_TF,
                         'note': 'useful (fp32-equivalent, padding excluded) flops of all MFMA launches of one frame / HIP-event '
                                 'time of the net stage; peak = dense MFMA peak of the arithmetic (f16x3: 2516.6/3 TFLOP/s because '
                                 'every product block costs three fp16 MFMAs; f32: 157.3); traffic = PMC HBM bytes per launch '
                                 '(profiles/r01_traffic_pmc.json)'},
            'roofline_hbm': {'bound': 'hbm', 'kernel': 'extract_kernel + integrate_*_kernel',
                             'achieved': bytes_frame / ei_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': bytes_frame / ei_s / 1e9 / HBM_PEAK_GBS, 'traffic': hbm_traffic,
                             'bytes_per_frame': bytes_frame, 'unique_gather_voxels': ug, 'unique_scatter_voxels': us},
        }
        if world == 1 and args.cpu_frames > 0:
            out['cpu_baseline'] = cpu_baseline(args, h, w, grid, args.semantics)
            out['speedup_vs_cpu_baseline'] = fps / out['cpu_baseline']['value']
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
