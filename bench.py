#!/usr/bin/env python
"""frames/sec fused: the per-frame hot path (extract -> FusionNet_v3 -> integrate) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one Pipeline.fuse call: one synthetic depth frame fused into an fp16 volume resident in HBM.  Default
workload = BASELINE.json configs[1] (320x240 depth into 256^3, geometry-only, FusionNet_v3); flags select the other
configurations and `config.workload` says which one ran.  Frames (depth, mask) are resident in HBM before the timed
region; poses stay on the host like any small per-frame metadata.  With N > 1 every rank owns one scene (its volumes
+ its own frame stream): no data-path collective, weak scaling, value = N*K / max-over-ranks time.

Rank 0 prints ONE JSON line with the driver's contract fields plus
  stages_ms            extract / net / integrate from fence-free HIP events on the launch stream, sampled on every 4th
                       frame INSIDE the timed region (an event record is a marker packet that idles the queue ~4.5 us)
  stages_ms_all_frames the same from a separate UNTIMED pass with events on every frame
  kernels              per-kernel table of the fusion net from a profiled forward (ojf_net_profile: one event behind
                       every launch): launches, us per frame, algorithmic GFLOP, TFLOP/s, fraction of the MFMA peak
  roofline             the kernel that takes most of the frame (from that table), against the dense MFMA peak of the
                       arithmetic; roofline_net = all MFMA launches of the frame over the net stage time
  roofline_hbm         extract + integrate against the HBM roofline (algorithmic bytes of SURVEY.md §8d)
  cpu_baseline         the op-for-op torch-CPU port of the reference path timed on this node's host cores
  secondary            (N = 1, default flags only) short runs of the other BASELINE configurations in the same process:
                       fp32-input MFMA arithmetic, + semantics (gt labels), + semantics predicted by AdapNet++,
                       640x480 -> 512^3 (configs[4] size) without / with semantics, and the configs[3] TRAINING frame step

Timing: after W warm-up steps the K-step timed loop (barrier + synchronize on both sides, max over ranks) is run
``--repeats`` times (default 5) back to back; ``value`` / ``ms_per_step`` are the MEDIAN repeat, ``value_min`` / ``value_max``
the slowest / fastest one (a single 9 ms region moves by a percent with one slow launch).

``--train`` times BASELINE configs[3]'s frame step instead (train_fusion.py:145-189): fuse_training + FusionLoss + backward
on every frame, and at every accumulation boundary (8 frames) the flat-gradient all-reduce over RCCL + the RMSprop step;
``allreduce_us`` is the mean duration of that collective from events on the launch stream.

Launching: ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (one
process per GPU, rendezvous on 127.0.0.1) and waits for them; under ``torch.distributed.run`` the ranks are used as given.
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
from online_joint_depthfusion_and_semantic_amd.database import Database, Voxelgrid  # noqa: E402
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
DISTINCT_FRAMES = 64  # distinct synthetic frames kept resident per case (cycled when steps + warmup exceed it)


class BenchStream(SyntheticStream):
    """Synthetic stream whose ground-truth grid is a constant volume: Pipeline.fuse never reads the GT grid, and
    sampling the analytic scene at 512^3 voxel centres would take a minute of host time per case."""

    def get_grid(self, scene, truncation, semantic_grid=True):
        g = Voxelgrid(self.resolution)
        g.from_array(np.full((self.grid,) * 3, truncation, dtype=np.float16), self.bbox)
        if semantic_grid:
            s = Voxelgrid(self.resolution)
            s.from_array(np.zeros((self.grid,) * 3, dtype=np.uint8), self.bbox)
            return (g, s)
        return (g,)


def seeded_weights(pipe, seed=1911):
    torch.manual_seed(seed)
    for m in pipe._fusion_network.modules():  # train_fusion.py:29-31 + non-trivial BN statistics
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.xavier_normal_(m.weight)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)


def workload_name(c):
    sem = ''
    if c['semantics']:
        sem = ' + semantics (%s, %d classes)' % ('AdapNet++ predicted per frame, %s convolutions' % c['seg_engine']
                                                 if c['strategy'] == 'predict' else 'gt labels, two-head net', c['n_classes'])
    base = {(240, 320, 256): 'BASELINE configs[%d]' % (2 if c['semantics'] else 1),
            (480, 640, 512): 'BASELINE configs[4] size',
            (120, 160, 64): 'BASELINE configs[0] size'}.get((c['h'], c['w'], c['grid']), 'custom size')
    return '%s: %dx%d depth into a %d^3 fp16 TSDF grid, FusionNet_v3%s, %s integrate, %s arithmetic, one scene per GPU' % (
        base, c['w'], c['h'], c['grid'], sem if sem else ', geometry-only', c['mode'], c['arith'])


CPU_WARMUP = 2  # untimed frames in front of the CPU baseline's sample (>= 10 frames after 2 warm-up)


def cpu_baseline(args, h, w, grid, semantics):
    """Times oracle/torch_port.fuse (the reference's op sequence on torch-CPU) on a bounded sample, on a stated number
    of threads: min(16, cores) - with all 128 host threads the oneDNN / ATen calls of this size oversubscribe and run
    3-4x SLOWER (7.2 s/frame on 128 threads vs 1.9 s on 8 in the build container)."""
    from oracle import torch_port  # checker / baseline only - never on the product path
    from online_joint_depthfusion_and_semantic_amd.model import FusionNet_v3
    nproc = os.cpu_count() or 1
    before = torch.get_num_threads()
    threads = min(16, nproc)
    torch.set_num_threads(threads)
    try:
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
            for i in range(args.cpu_frames + CPU_WARMUP):
                b = st.batch(i)
                t0 = time.perf_counter()
                torch_port.fuse(b, vols, net, origin, st.resolution, semantics=semantics)
                times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(before)
    t = float(np.mean(times[CPU_WARMUP:]))
    return {'value': 1.0 / t, 'unit': 'frames/sec', 'cores': threads, 'host_cores': nproc, 'torch_threads_default': before,
            'kind': 'port',
            'sample': '%d frames of the %dx%d -> %d^3 workload after 2 warm-up frames (SURVEY.md 8d), torch-CPU op-for-op port of the reference '
                      '(oracle/torch_port.py) on %d threads (torch.set_num_threads; %d host cores), %.2f s/frame'
                      % (args.cpu_frames, w, h, grid, threads, nproc, t)}


def pmc_traffic(c):
    """HBM bytes per frame per kernel REPLAYED from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this bench, tools/pmc_traffic.py; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes) - not
    observed by this run.  Only valid for the workload the passes were taken on; otherwise None."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_*traffic_pmc.json')))
    if not found:
        return None, 'no profiles/rNN_*traffic_pmc.json in this tree (tools/pmc_profile.sh writes one)'
    path = found[-1]  # the latest round's passes
    name = os.path.basename(path)
    if (c['h'], c['w'], c['grid'], c['semantics'], c['arith']) != (240, 320, 256, False, 'f16x3'):
        return None, 'the committed counter passes (profiles/%s) were taken on BASELINE configs[1] with the f16x3 arithmetic, not on this workload' % name
    with open(path) as f:
        return json.load(f), 'profiles/' + name


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


def net_kernel_macs(c_ch=19, gf=5):
    """Algorithmic MAC per pixel of every kernel class of the single-head net (channel padding excluded; c = 2 * n_points
    + 1 = 19); the classes add up to the library's ojf_net_macs_per_pixel (326 876)."""
    out = c_ch * (gf + 1)
    dense = sum(9 * c_ch * (i + 1) * c_ch + 9 * c_ch * c_ch for i in range(gf))
    branches = 8 * 9 * c_ch * c_ch              # per VortexPooling: four branches x two dilated 3x3 (19 -> 19)
    entry = out * 4 * c_ch                      # per VortexPooling: the four branch-entry 1x1 (114 -> 19) stacked
    # closing 1x1s + the final 1x1 over the 5 * 114 concat (the global-average columns are folded into its bias here,
    # but they are per-pixel MACs of the reference's layer)
    tail = 4 * (c_ch * out + out * out) + out * out
    widths = [out] + [c_ch * k for k in range(gf, 0, -1)]
    head = sum(a * b + b * b for a, b in zip(widths[:-1], widths[1:])) + c_ch * 9
    return {'dense': dense, 'branches': branches, 'entry': entry, 'tail': tail, 'head': head}


class Case:
    """One workload: pipeline + volumes + resident frames."""

    def __init__(self, c, dev, rank, n_frames):
        self.c = c
        h, w, grid = c['h'], c['w'], c['grid']
        cfg = default_config(h, w, semantics=c['semantics'], integrate_mode=c['mode'], n_classes=c['n_classes'])
        cfg.SETTINGS.device = str(dev)
        cfg.FUSION_MODEL.arithmetic = c['arith']
        if c['semantics']:
            cfg.DATA.semantic_strategy = c['strategy']
            cfg.SEMANTIC_2D_MODEL.engine = c['seg_engine']
        self.cfg = cfg
        n_distinct = min(n_frames, DISTINCT_FRAMES)
        self.st = BenchStream(h, w, grid, max(n_distinct, 40), scene='room_%d' % rank, seed=1911 + rank, n_classes=c['n_classes'])
        self.db = Database(self.st, database_config(cfg))
        pipe = Pipeline(cfg)
        seeded_weights(pipe)
        self.pipe = pipe.to(dev).eval()
        self.dev = dev
        # frames resident in HBM before the clock starts; poses and ids stay host-side metadata
        self.batches = []
        image = torch.zeros((1, 3, h, w), device=dev)  # only its shape is read on the geometry / gt-label paths
        predict = c['semantics'] and c['strategy'] == 'predict'
        for i in range(n_distinct):
            f = self.st.frame(i)
            b = {'image': torch.from_numpy(f['image']).unsqueeze(0).to(dev) if predict else image, 'frame_id': [f['frame_id']],
                 self.st.depth_key: torch.from_numpy(f[self.st.depth_key]).unsqueeze(0).to(dev),
                 'mask': torch.from_numpy(f['mask']).unsqueeze(0).to(dev),
                 'extrinsics': torch.from_numpy(f['extrinsics']).unsqueeze(0),
                 'intrinsics': torch.from_numpy(f['intrinsics']).unsqueeze(0)}
            if c['semantics']:
                b['semantic_gt'] = torch.from_numpy(f['semantic_gt']).unsqueeze(0).to(dev)
            self.batches.append(b)

    def fuse(self, i):
        self.pipe.fuse(self.batches[i % len(self.batches)], self.db, self.dev)

    def run(self, steps, warmup, sync, profile_frames=32, kernel_reps=5, repeats=1):
        pipe = self.pipe
        with torch.no_grad():
            pipe.profile = False
            for i in range(warmup):
                self.fuse(i)
            pipe.profile = 4  # stage events on every 4th frame of the timed region
            pipe.reset_profile()
            times = []
            at = warmup
            for _ in range(repeats):  # each repeat: EXACTLY `steps` frames between two barrier + synchronize points
                sync()
                t0 = time.perf_counter()
                for i in range(at, at + steps):
                    self.fuse(i)
                sync()
                times.append(time.perf_counter() - t0)
                at += steps
            steps_done = at - warmup
            pipe.check()  # outside the timed region: raises if the split-fp16 range guard fired on any frame
            stages = pipe.stage_times_ms()
            n_samples = len(pipe._marks) // 4
            # separate UNTIMED pass: stage events on every frame
            pipe.profile = True
            pipe.reset_profile()
            for i in range(profile_frames):
                self.fuse(warmup + steps_done + i)
            stages_all = pipe.stage_times_ms() if profile_frames else {}
            pipe.profile = False
            # per-kernel table of the fusion net: profiled forwards on the last frame's packed input
            table = {}
            reps = kernel_reps
            for _ in range(reps):
                for k, (name, us) in enumerate(pipe._engine.profile(pipe._est)):
                    ent = table.setdefault(name, [0, 0.0])
                    ent[0] += 1
                    ent[1] += max(us, 0.0)
            kernels = {name: {'launches_per_frame': n / reps, 'us_per_frame': us / reps} for name, (n, us) in table.items()}
            pipe.check()
        return {'times': times, 'stages': stages, 'stage_samples': n_samples, 'stages_all': stages_all,
                'stages_all_frames': profile_frames, 'kernels': kernels, 'launches': pipe._engine.launches}


def run_lookahead(case, steps, warmup, sync, L, repeats=3, prefetch=True):
    """One scene, Pipeline.fuse_sequence on chunks of L consecutive frames (the 2-D network of the L frames as one batched pass,
    the frame steps in order): frames/s of a recorded stream handed over in chunks, next to the frame-at-a-time predict leg."""
    pipe, n = case.pipe, len(case.batches)
    chunk = lambda i: [case.batches[(i + k) % n] for k in range(L)]
    times = []
    with torch.no_grad():
        for i in range(0, warmup, L):
            pipe.fuse_sequence(chunk(i), case.db, case.dev, prefetch=chunk(i + L) if prefetch else None)
        at = (warmup + L - 1) // L * L
        steps = steps // L * L
        for _ in range(repeats):
            sync()
            t0 = time.perf_counter()
            for i in range(at, at + steps, L):
                pipe.fuse_sequence(chunk(i), case.db, case.dev, prefetch=chunk(i + L) if prefetch else None)
            sync()
            times.append(time.perf_counter() - t0)
            at += steps
        pipe.check()
    times.sort()
    med = times[len(times) // 2]
    return {'workload': workload_name(case.c) + ' - one scene, labels of %d consecutive frames predicted as one batched pass (Pipeline.fuse_sequence), frame steps in order' % L,
            'lookahead_frames': L, 'next_chunk_prefetched_on_side_stream': bool(prefetch), 'value': steps / med, 'unit': 'frames/sec', 'ms_per_step': 1e3 * med / steps, 'steps': steps, 'warmup': warmup,
            'repeats': repeats, 'value_min': steps / times[-1], 'value_max': steps / times[0]}


class ManyScenes:
    """S BenchStreams as one dataset object (Database pulls .scenes / .get_grid from it, modules/database.py:48-58)."""

    def __init__(self, streams):
        self.streams = {st.scene: st for st in streams}
        self.scenes = [st.scene for st in streams]

    def get_grid(self, scene, truncation, semantic_grid=True):
        return self.streams[scene].get_grid(scene, truncation, semantic_grid)


class ManyCase:
    """S scenes of one frame size on ONE GPU, one frame of each per step through Pipeline.fuse_many:
    the aggregate frames/s of the device when the per-frame launch chains of several scenes run side by side."""

    def __init__(self, c, dev, rank, n_frames, S):
        self.c, self.S, self.dev = c, S, dev
        h, w, grid = c['h'], c['w'], c['grid']
        cfg = default_config(h, w, semantics=c['semantics'], integrate_mode=c['mode'], n_classes=c['n_classes'])
        cfg.SETTINGS.device = str(dev)
        cfg.FUSION_MODEL.arithmetic = c['arith']
        if c['semantics']:
            cfg.DATA.semantic_strategy = c['strategy']
            cfg.SEMANTIC_2D_MODEL.engine = c['seg_engine']
        self.cfg = cfg
        n_distinct = min(n_frames, DISTINCT_FRAMES)
        streams = [BenchStream(h, w, grid, max(n_distinct, 40), scene='room_%d_%d' % (rank, k), seed=1911 + rank + 101 * k, n_classes=c['n_classes'])
                   for k in range(S)]
        self.db = Database(ManyScenes(streams), database_config(cfg))
        self.streams0 = streams[0]
        pipe = Pipeline(cfg)
        seeded_weights(pipe)
        self.pipe = pipe.to(dev).eval()
        predict = c['semantics'] and c['strategy'] == 'predict'
        image = torch.zeros((1, 3, h, w), device=dev)
        self.batches = []
        for i in range(n_distinct):
            row = []
            for st in streams:
                f = st.frame(i)
                b = {'image': torch.from_numpy(f['image']).unsqueeze(0).to(dev) if predict else image, 'frame_id': [f['frame_id']],
                     st.depth_key: torch.from_numpy(f[st.depth_key]).unsqueeze(0).to(dev), 'mask': torch.from_numpy(f['mask']).unsqueeze(0).to(dev),
                     'extrinsics': torch.from_numpy(f['extrinsics']).unsqueeze(0), 'intrinsics': torch.from_numpy(f['intrinsics']).unsqueeze(0)}
                if c['semantics']:
                    b['semantic_gt'] = torch.from_numpy(f['semantic_gt']).unsqueeze(0).to(dev)
                row.append(b)
            self.batches.append(row)

    def run(self, steps, warmup, sync, repeats=3):
        times = []
        with torch.no_grad():
            for i in range(warmup):
                self.pipe.fuse_many(self.batches[i % len(self.batches)], self.db, self.dev)
            at = warmup
            host = []
            for _ in range(repeats):
                sync()
                t0 = time.perf_counter()
                for i in range(at, at + steps):
                    self.pipe.fuse_many(self.batches[i % len(self.batches)], self.db, self.dev)
                host.append(time.perf_counter() - t0)  # the host's share: enqueue time of the loop (the queue never blocks it here)
                sync()
                times.append(time.perf_counter() - t0)
                at += steps
            self.pipe.check()
            # stage times of the joint launches (events around extract_many / integrate_many on every call of an untimed pass)
            stages = {}
            try:
                self.pipe.profile = True
                self.pipe.reset_profile()
                for i in range(at, at + 12):
                    self.pipe.fuse_many(self.batches[i % len(self.batches)], self.db, self.dev)
                stages = self.pipe.stage_times_ms()
                self.pipe.profile = False
                self.pipe.check()
            except Exception as e:
                stages = {'error': repr(e)}
        self.host_ms_per_call = 1e3 * sorted(host)[len(host) // 2] / steps
        times.sort()
        med = times[len(times) // 2]
        out = {'workload': workload_name(self.c) + ' - %d scenes per GPU, one frame of each per Pipeline.fuse_many call' % self.S, 'scenes_per_gpu': self.S,
               'value': self.S * steps / med, 'unit': 'frames/sec (all scenes)', 'ms_per_call': 1e3 * med / steps, 'steps': steps, 'warmup': warmup,
               'repeats': repeats, 'value_min': self.S * steps / times[-1], 'value_max': self.S * steps / times[0],
               'host_enqueue_ms_per_call': self.host_ms_per_call}
        if 'extract' in stages:
            # the S scenes' gather and scatter are ONE launch each (ojf_extract_many / ojf_integrate_many): per-frame time = call / S
            cfg = self.cfg
            st0 = self.streams0
            bytes_frame, ug, us = algorithmic_bytes(st0, [warmup % st0.n_frames, (warmup + 7) % st0.n_frames], cfg.FUSION_MODEL.n_points,
                                                    cfg.FUSION_MODEL.n_tail_points, self.c['semantics'])
            ei_us = 1e3 * (stages['extract'] + stages['integrate']) / self.S
            out['stages_ms_per_call'] = stages
            out['extract_integrate_us_per_frame'] = ei_us
            out['roofline_hbm'] = {'bound': 'hbm', 'kernel': 'extract_tile_many_kernel + integrate_*_many_kernel (blockIdx.y = scene)',
                                   'achieved': bytes_frame / (ei_us * 1e-6) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': bytes_frame / (ei_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 'bytes_per_frame': bytes_frame,
                                   'note': 'algorithmic bytes of one frame (scene 0 of the call, SURVEY.md 8d) x S / HIP-event time of the two joint launches'}
        elif stages:
            out['stages_ms_per_call'] = stages
        return out


def kernel_table(kernels, c, N, peak_tf, total_macs):
    """Adds algorithmic GFLOP / TFLOP/s / fraction of the arithmetic's MFMA peak to the profiled per-kernel times
    (geometry-only net; with the semantic head the table carries times only)."""
    flops = {}
    if not c['semantics']:
        m = net_kernel_macs()
        flops = {'dense_pair_kernel': m['dense'], 'dense_chain_kernel': m['dense'], 'conv_f16x3_kernel (grouped)': 2 * m['branches'], 'vortex_branch_kernel': 2 * m['branches'], 'conv_mfma_kernel': None,
                 'entry1x1_kernel': m['entry'], 'vortex_tail_kernel (+ next entry GEMM)': m['tail'] + m['entry'],
                 'vortex_tail_kernel (+ prediction head)': m['tail'] + m['head']}
        assert m['dense'] + 2 * m['branches'] + 2 * m['entry'] + 2 * m['tail'] + m['head'] == total_macs, (m, total_macs)
    rows = []
    for name, k in sorted(kernels.items(), key=lambda kv: -kv[1]['us_per_frame']):
        row = {'kernel': name, 'launches_per_frame': k['launches_per_frame'], 'us_per_frame': round(k['us_per_frame'], 2)}
        if flops.get(name) and k['us_per_frame'] > 0:
            gf = 2.0 * flops[name] * N / 1e9
            row['gflop'] = round(gf, 3)
            row['tflops'] = round(gf / (k['us_per_frame'] * 1e-6) / 1e3, 2)
            row['frac_of_mfma_peak'] = round(row['tflops'] / peak_tf, 4)
        rows.append(row)
    return rows


def adapnet_gflop(net, h, w):
    """Useful GFLOP of one AdapNet++ forward (convolutions and transposed convolutions that feed output[0]: the auxiliary
    heads the engine skips are left out), counted by running a meta-device copy of the module tree with hooks."""
    import copy
    m = copy.deepcopy(net).to('meta')
    macs = [0]

    def hook(mod, inp, out):
        k = mod.kernel_size[0] * mod.kernel_size[1]
        hw = inp[0].shape[2] * inp[0].shape[3] if isinstance(mod, torch.nn.ConvTranspose2d) else out.shape[2] * out.shape[3]
        macs[0] += hw * mod.in_channels * mod.out_channels * k // mod.groups
    for name, mm in m.named_modules():
        if isinstance(mm, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and 'aux' not in name:
            mm.register_forward_hook(hook)
    x = torch.zeros(1, 3, h, w, device='meta')
    with torch.no_grad():
        m(x, x) if getattr(net, 'fusion', False) else m(x)
    return 2e-9 * macs[0]


def segmentation_report(case, seg_ms, c):
    """The AdapNet++ share of a predict-strategy frame: its time (HIP events around Pipeline._frame_semantics on the sampled
    frames: graph replay + the host work up to the extract launch), useful TFLOP/s against the split-fp16 MFMA peak, and the
    MFMA-pipe-busy time of its convolution kernels per frame replayed from the round's SQ-counter pass."""
    out = {'ms_per_frame': seg_ms, 'engine': c['seg_engine']}
    try:
        gf = adapnet_gflop(case.pipe._semantic_2d_network, c['h'], c['w'])
        out.update(gflop=gf, tflops=gf / seg_ms, frac_of_mfma_peak=gf / seg_ms / ARITH['f16x3'][2], frac_of_raw_fp16_peak=gf / seg_ms / F16_MFMA_PEAK_TF,
                   note='useful flops of the convolutions feeding output[0] / segmentation stage time; peak = split-fp16 (838.9 TFLOP/s)')
    except Exception as e:  # a reporting extra must not fail the leg
        out['gflop_error'] = repr(e)
    import glob
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r[0-9][0-9]_*sq_counters_predict.json')))
    if found and (c['h'], c['w']) == (240, 320) and c['seg_engine'] == 'hip':
        try:
            path = found[-1]  # the latest round's counter pass
            sq = json.load(open(path))
            rows = {k: v for k, v in sq.items() if k.startswith('segconv') and isinstance(v, dict)}
            frames = float(sq.get('_frames', 22.0))  # frames of that counter pass (bench.py --steps 20 --warmup 2 --lean unless the file says otherwise)
            out['replayed_not_measured_in_this_run'] = {
                'mfma_pipe_busy_us_per_frame': sum(v['mfma_busy_us_per_simd_per_dispatch'] * v['dispatches'] for v in rows.values()) / frames,
                'segconv_dispatches_per_frame': sum(v['dispatches'] for v in rows.values()) / frames,
                'source': 'profiles/%s (segconv* kernels, per SIMD; a rocprofv3 --pmc pass of an earlier run of this workload)' % os.path.basename(path)}
        except Exception as e:
            out['mfma_error'] = repr(e)
    return out


def report(case, res, steps, warmup, world, args_cpu_frames=0, full=True):
    c, cfg, st = case.c, case.cfg, case.st
    h, w, grid = c['h'], c['w'], c['grid']
    times = sorted(res['times'])
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    fps = world * steps / med
    N = h * w
    peak = ARITH[c['arith']][2]
    stages = res['stages']
    out = {'workload': workload_name(c), 'value': fps, 'unit': 'frames/sec', 'ms_per_step': 1e3 * med / steps,
           'steps': steps, 'warmup': warmup, 'repeats': len(times), 'value_min': world * steps / times[-1],
           'value_max': world * steps / times[0], 'stages_ms': stages, 'stage_samples_in_timed_region': res['stage_samples'],
           'stages_ms_all_frames': res['stages_all'], 'stages_all_frames_pass': '%d untimed frames, events on every frame' % res['stages_all_frames'],
           'net_launches_per_frame': res['launches']}
    if c['semantics'] and c['strategy'] == 'predict' and 'segmentation' in stages:
        out['segmentation'] = segmentation_report(case, stages['segmentation'], c)
    if not full:
        return out
    P, T = cfg.FUSION_MODEL.n_points, cfg.FUSION_MODEL.n_tail_points
    flops = 2.0 * case.pipe._engine.macs_per_pixel * N  # useful flops, channel padding excluded
    net_s = stages['net'] / 1e3
    rows = kernel_table(res['kernels'], c, N, peak, case.pipe._engine.macs_per_pixel)
    out['kernels'] = rows
    dom = next((r for r in rows if 'tflops' in r), None)
    tr, tr_src = pmc_traffic(c)
    net_traffic = hbm_traffic = None
    if tr:
        net_k = [k for k in tr if k.startswith(('conv_mfma', 'conv_f16x3', 'chain1x1', 'vortex_tail', 'vortex_branch', 'dense_pair', 'dense_chain', 'entry1x1'))]
        net_traffic = sum(tr[k]['fetch_bytes_per_frame_x2'] + tr[k]['write_bytes_per_frame'] for k in net_k)
        hbm_traffic = sum(tr[k]['fetch_bytes_per_frame_x2'] + tr[k]['write_bytes_per_frame'] for k in tr if 'extract' in k or 'integrate' in k)
    if dom:
        per_launch_us = dom['us_per_frame'] / dom['launches_per_frame']
        dom_traffic = None
        if tr:
            key = [k for k in tr if k.startswith(dom['kernel'].split(' ')[0])]
            if key:
                dom_traffic = sum(tr[k]['fetch_bytes_per_frame_x2'] + tr[k]['write_bytes_per_frame'] for k in key) / dom['launches_per_frame']
        out['roofline'] = {'bound': 'mfma', 'kernel': dom['kernel'], 'launches_per_frame': dom['launches_per_frame'],
                           'avg_launch_us': per_launch_us, 'flops_per_launch': dom['gflop'] * 1e9 / dom['launches_per_frame'],
                           'achieved': dom['tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': dom['tflops'] / peak,
                           'frac_of_raw_fp16_peak': dom['tflops'] / F16_MFMA_PEAK_TF,
                           'traffic': dom_traffic, 'traffic_source': ('replayed from ' + tr_src) if dom_traffic is not None else ('none: ' + str(tr_src)),
                           'note': 'the net kernel with the largest share of the frame (kernels[] lists all): algorithmic '
                                   '(fp32-equivalent, padding excluded) flops per launch / its launch duration from HIP events behind '
                                   'every launch on the launch stream (ojf_net_profile); peak = dense MFMA peak of the arithmetic '
                                   '(f16x3: 2516.6/3 TFLOP/s because every product block costs three fp16 MFMAs; f32: 157.3)'}
    out['roofline_net'] = {'bound': 'mfma', 'kernel': 'all MFMA launches of the frame', 'achieved': flops / net_s / 1e12, 'peak': peak,
                           'unit': 'TFLOP/s', 'frac': flops / net_s / 1e12 / peak, 'flops_per_frame': flops,
                           'frac_of_f32_mfma_peak': flops / net_s / 1e12 / F32_MFMA_PEAK_TF,
                           'frac_of_raw_fp16_peak': flops / net_s / 1e12 / F16_MFMA_PEAK_TF, 'traffic': net_traffic,
                           'traffic_source': ('replayed from ' + tr_src) if net_traffic is not None else ('none: ' + str(tr_src)),
                           'note': 'useful flops of the whole net / HIP-event time of the net stage inside the timed region'}
    n_b = len(case.batches)
    sample = [(warmup + k * max(1, steps // 4)) % n_b for k in range(4)]
    bytes_frame, ug, us = algorithmic_bytes(st, sample, P, T, c['semantics'])
    ei_s = (stages['extract'] + stages['integrate']) / 1e3
    out['roofline_hbm'] = {'bound': 'hbm', 'kernel': 'extract_tile_kernel + integrate_*_kernel',
                           'achieved': bytes_frame / ei_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': bytes_frame / ei_s / 1e9 / HBM_PEAK_GBS, 'traffic': hbm_traffic,
                           'traffic_source': ('replayed from ' + tr_src) if hbm_traffic is not None else ('none: ' + str(tr_src)),
                           'bytes_per_frame': bytes_frame, 'unique_gather_voxels': ug, 'unique_scatter_voxels': us}
    return out


class TrainCase:
    """BASELINE configs[3] on this rank: one scene, Pipeline.fuse_training + FusionLoss + backward per frame, and at every
    accumulation boundary the flat-gradient all-reduce (RCCL when world > 1) + RMSprop step (train_fusion.py:145-189;
    drivers.train_fusion is the full driver, this is its inner loop without logging / validation / host reads of the loss)."""

    def __init__(self, h, w, grid, dev, rank, n_frames, accum=8):
        from online_joint_depthfusion_and_semantic_amd.distributed import FlatGradientAllReduce
        from online_joint_depthfusion_and_semantic_amd.drivers import _training_defaults
        from online_joint_depthfusion_and_semantic_amd.loss import FusionLoss
        cfg = _training_defaults(default_config(h, w))
        cfg.SETTINGS.device = str(dev)
        # like drivers.train_fusion: the backward pass of a frame beside the next frame's forward stage (step() keeps the gradient
        # work inside pipe.gradients()); OJF_BENCH_TRAIN_OVERLAP=0: the serial loop (A/B switch)
        cfg.FUSION_MODEL.train_overlap = os.environ.get('OJF_BENCH_TRAIN_OVERLAP', '1') not in ('0',)
        cfg.FUSION_MODEL.train_overlap_thread = os.environ.get('OJF_BENCH_TRAIN_THREAD', '0') not in ('0',)
        self.cfg, self.dev, self.accum = cfg, dev, accum
        n_distinct = min(n_frames, DISTINCT_FRAMES)
        self.st = SyntheticStream(h, w, grid, max(n_distinct, 40), scene='room_%d' % rank, seed=1911 + rank)
        self.db = Database(self.st, database_config(cfg))
        pipe = Pipeline(cfg)
        seeded_weights(pipe)  # the same on every rank (a replica), like train_fusion's broadcast initial state
        self.pipe = pipe.to(dev).train()
        self.crit = FusionLoss(w_l1=cfg.TRAINING.loss.w_l1, w_l2=cfg.TRAINING.loss.w_l2, w_cos=cfg.TRAINING.loss.w_cos)
        self.grads = FlatGradientAllReduce(self.pipe._fusion_network)
        o = cfg.TRAINING.optimizer
        self.opt = torch.optim.RMSprop(self.pipe._fusion_network.parameters(), lr=o['lr'], momentum=o['momentum'],
                                       weight_decay=o['weight_decay'], eps=o['eps'])
        self.batches = []
        image = torch.zeros((1, 3, h, w), device=dev)
        for i in range(n_distinct):
            f = self.st.frame(i)
            self.batches.append({'image': image, 'frame_id': [f['frame_id']],
                                 self.st.depth_key: torch.from_numpy(f[self.st.depth_key]).unsqueeze(0).to(dev),
                                 'mask': torch.from_numpy(f['mask']).unsqueeze(0).to(dev),
                                 'extrinsics': torch.from_numpy(f['extrinsics']).unsqueeze(0),
                                 'intrinsics': torch.from_numpy(f['intrinsics']).unsqueeze(0)})
        self.reduce_events = []
        self.loss_sum = None
        # A/B switch.  Measured (round 6, profiles/r06_train_host.txt): announcing does not move the step on the boxes seen (211 against 214
        # frames/s): the host's enqueue time under a busy queue, not the wait for the count, is what paces it - off by default
        self.announce = os.environ.get('OJF_BENCH_ANNOUNCE', '0') not in ('', '0')

    def step(self, i):
        # (what drivers.train_fusion does: the next frame is announced, so that its valid-ray count - the frame step's one host
        # read - is requested a frame ahead and the host is never tied to the device inside a frame)
        if self.announce:
            self.pipe.announce_training_frame(self.batches[(i + 1) % len(self.batches)], self.dev)
        out = self.pipe.fuse_training(self.batches[i % len(self.batches)], self.db, self.dev)
        if out['tsdf_fused'].shape[1]:
            loss = self.crit.forward(out['tsdf_fused'], out['tsdf_target'])
            loss.backward()
            # train_fusion.py:172 adds loss.item() to a window that is read every log_freq frames; summed on the device
            # and read once per timed loop here (the same numbers, without a host round trip per frame)
            self.loss_sum = loss.detach() if self.loss_sum is None else self.loss_sum + loss.detach()
        boundary = (i + 1) % self.accum == 0

        def gradient_step():  # (FUSION_MODEL.train_overlap: on the gradient stream, behind this frame's backward pass)
            if self.cfg.TRAINING.optimization.clipping:  # train_fusion.py:182-183: on the accumulated gradients, every frame
                self.grads.clip_(1.0)  # (= clip_grad_norm_(parameters, 1., 2): the gradients are views into the flat buffer)
            if boundary:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.grads.reduce()  # the only collective of the training path
                b.record()
                self.reduce_events.append((a, b))
                self.opt.step()
                self.grads.zero()
        self.pipe.gradient_work(gradient_step, join=boundary)  # like drivers.train_fusion: this thread waits only where the weights change

    def run(self, steps, warmup, sync, repeats):
        for i in range(warmup):
            self.step(i)
        self.pipe.join_gradients()
        self.reduce_events = []
        times, host, at = [], [], warmup
        for _ in range(repeats):
            sync()
            t0 = time.perf_counter()
            self.loss_sum = None
            for i in range(at, at + steps):
                self.step(i)
            self.pipe.join_gradients()  # (the gradient thread has enqueued everything it was handed: the loop's launches are all in the queues)
            host.append(time.perf_counter() - t0)  # the host's share: Python + enqueue time of the loop (incl. its waits for the valid-ray counts)
            mean_loss = float(self.loss_sum) / steps if self.loss_sum is not None else float('nan')  # the log window's read
            sync()
            times.append(time.perf_counter() - t0)
            at += steps
        self.host_ms_per_frame = 1e3 * sorted(host)[len(host) // 2] / steps
        tn = self.pipe.__dict__.get('_hip_train')
        self.net_launches_per_pass = int(tn.launches) if tn is not None and hasattr(tn, 'launches') else None
        loss_ok = bool(torch.isfinite(self.grads.flat).all()) and mean_loss == mean_loss
        ar = [a.elapsed_time(b) * 1e3 for a, b in self.reduce_events]
        # (world 1: the bracket is empty, ~5 us of event overhead - reported as None)
        return {'times': times, 'allreduce_us': float(np.mean(ar)) if ar else None, 'allreduce_calls': len(ar),
                'gradient_bytes': self.grads.nbytes, 'finite': loss_ok, 'mean_loss': mean_loss}


def train_report(case, res, steps, warmup, world, h, w, grid, grouped=False):
    times = sorted(res['times'])
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {'workload': 'BASELINE configs[3]: training frame step (fuse_training + FusionLoss + backward per frame; flat-gradient '
                        'all-reduce + RMSprop step every %d frames, clip_grad_norm_ every frame), %dx%d depth into a %d^3 grid, FusionNet_v3 train() mode '
                        '(batch statistics, dropout), split-fp16 forward / backward-data convolutions and weight gradients, one scene per GPU%s' % (
                            case.accum, w, h, grid, '; a frame\'s backward pass runs on the gradient stream beside the next frame\'s forward stage '
                            '(FUSION_MODEL.train_overlap, the training driver\'s default: same bits as the serial loop)' if case.cfg.FUSION_MODEL.get('train_overlap', False) else ''),
            'value': world * steps / med, 'unit': 'frames/sec', 'ms_per_step': 1e3 * med / steps, 'steps': steps, 'warmup': warmup,
            'repeats': len(times), 'value_min': world * steps / times[-1], 'value_max': world * steps / times[0],
            'allreduce_us': res['allreduce_us'] if (world > 1 or grouped) else None, 'allreduce_calls_in_timed_region': res['allreduce_calls'],
            'allreduce_backend': ((('rccl' if torch.distributed.get_backend() == 'nccl' else 'gloo (dry run: gradient buffer staged through the host)')
                                   + (' (a group of one rank: --force-group)' if world == 1 else ''))
                                  if (world > 1 or grouped) else 'none (one rank)'), 'gradient_bytes': res['gradient_bytes'],
            'gradients_finite': res['finite'], 'mean_loss_last_repeat': res['mean_loss'],
            # host- or device-bound?  the loop's enqueue time per frame (the host waits ~1.7 ms of it for the frame's valid-ray count,
            # i.e. for the device) against the frame time; launches of the net's forward + backward pass as the executor counts them
            'host_loop_ms_per_frame': getattr(case, 'host_ms_per_frame', None), 'net_launches_per_pass': getattr(case, 'net_launches_per_pass', None),
            # < 0.8: the device paces the step; ~1: the host does (the number then moves with the host CPU, not the GPU)
            'host_bound_ratio': (getattr(case, 'host_ms_per_frame', 0.0) or 0.0) / (1e3 * med / steps)}


def rccl_world1_probe(timeout=240):
    """The 8-GPU path's collective on a one-GPU box: ``bench.py --train --force-group`` in a process of its own - RCCL is loaded,
    a communicator of one rank set up and the flat 1.44-MB gradient buffer all-reduced at every accumulation boundary of a short
    training loop.  A failure is reported, never fatal to the headline line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--train', '--force-group', '--steps', '32', '--warmup', '16', '--repeats', '2', '--cpu-frames', '0']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    name = 'RCCL on one rank: training frame step with the flat-gradient all-reduce through a process group of ONE rank (nccl backend)'
    try:
        pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
        if pr.returncode != 0 or not line:
            return {'workload': name, 'error': 'rc %d: %s' % (pr.returncode, (pr.stderr or pr.stdout)[-400:])}
        j = json.loads(line[-1])
        return {'workload': name, 'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'allreduce_us': j.get('allreduce_us'),
                'allreduce_backend': j.get('allreduce_backend'), 'allreduce_calls_in_timed_region': j.get('allreduce_calls_in_timed_region'),
                'gradient_bytes': j.get('gradient_bytes'), 'gradients_finite': j.get('gradients_finite')}
    except Exception as e:
        return {'workload': name, 'error': repr(e)}


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script (one process per GPU, rendezvous on
    127.0.0.1) and wait for them.  Rank 0 prints the JSON line on the shared stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    codes = []
    try:
        for pr in procs:
            codes.append(pr.wait())
    finally:
        for pr in procs:  # a rank that died must not leave the others waiting in a collective
            if pr.poll() is None:
                pr.kill()
    bad = [c for c in codes if c != 0]
    if bad or len(codes) != n:
        raise SystemExit('bench.py: rank exit codes %r' % codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=5, help='how many times the K-step timed loop runs; value = the median repeat')
    ap.add_argument('--height', type=int, default=240)
    ap.add_argument('--width', type=int, default=320)
    ap.add_argument('--grid', type=int, default=256)
    ap.add_argument('--train', action='store_true', help='BASELINE configs[3]: time the training frame step (incl. the gradient all-reduce)')
    ap.add_argument('--semantics', action='store_true', help='BASELINE configs[2]-style: gt labels + semantic head')
    ap.add_argument('--semantic-strategy', default='gt', choices=['gt', 'predict'],
                    help="with --semantics: 'predict' runs AdapNet++ (random init) on every frame")
    ap.add_argument('--n-classes', type=int, default=30)
    ap.add_argument('--seg-engine', default='hip', choices=['hip', 'torch'],
                    help="AdapNet++ convolutions: 'hip' = SEGCONV MFMA kernels (default), 'torch' = module forward on MIOpen")
    ap.add_argument('--mode', default='fast', choices=['fast', 'parity'])
    ap.add_argument('--arith', default='f16x3', choices=['f16x3', 'f32'], help='net MFMA arithmetic (include/ojf.h OJF_ARITH_*)')
    ap.add_argument('--no-prefetch', action='store_true', help='with --lookahead: the next chunk is NOT announced to fuse_sequence (its 2-D pass then runs in front of its frame steps instead of beside the previous chunk\'s)')
    ap.add_argument('--lookahead', type=int, default=0, help='L > 1 (with --semantics --semantic-strategy predict): time Pipeline.fuse_sequence on chunks of L consecutive frames (prints its own line)')
    ap.add_argument('--scenes', type=int, default=0, help='S > 1: time Pipeline.fuse_many over S scenes on this GPU instead (prints its own line)')
    ap.add_argument('--cpu-frames', type=int, default=10, help='timed frames of the CPU baseline (0 = skip)')
    ap.add_argument('--secondary', type=int, default=None,
                    help='steps of each secondary workload (default: 60 when the headline runs with default flags on 1 GPU, else 0)')
    ap.add_argument('--lean', action='store_true',
                    help='counter-collection runs (tools/final_profile.sh): only warm-up + timed frames are fused - no stage-mark '
                         'pass, no profiled forwards, no secondary / CPU legs - so every kernel runs exactly steps + warmup times per launch site')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL, default) | gloo (validation of the N>1 path on a 1-GPU box)')
    ap.add_argument('--force-group', action='store_true',
                    help='N = 1: create a process group of ONE rank anyway (with --train the flat-gradient all-reduce then goes through RCCL: '
                         'library load, communicator set-up and the collective launch of the 8-GPU path on a one-GPU box)')
    ap.add_argument('--no-pin', action='store_true', help='N > 1: do not pin the ranks to disjoint core sets')
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.repeats < 1:
        raise SystemExit('bench.py: --gpus, --steps and --repeats must be positive')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no launcher: this process becomes the launcher of N ranks of itself
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
        if args.dist_backend == 'nccl' and torch.cuda.device_count() < args.gpus:
            raise SystemExit('bench.py: --gpus %d but only %d HIP device(s) visible (RCCL needs one device per rank; '
                             '--dist-backend gloo lets several ranks share a device for validation)' % (args.gpus, torch.cuda.device_count()))
        launch_ranks(args.gpus, sys.argv[1:])
        return
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    affinity = ''
    if world > 1 and not args.no_pin:
        # one disjoint core slice per rank (cores / ranks): the training leg is host-paced, unpinned ranks migrate
        from online_joint_depthfusion_and_semantic_amd.distributed import pin_rank_to_cores
        affinity = pin_rank_to_cores(rank, world)
    grouped = world > 1 or args.force_group
    if grouped:
        import torch.distributed as dist
        # RCCL: barrier + max-reduce of the times; with --train also the gradient all-reduce (the path's only collective)
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    def sync():
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(times):
        if grouped:
            t = torch.tensor(times, dtype=torch.float64, device=dev if args.dist_backend == 'nccl' else 'cpu')
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return [float(x) for x in t.tolist()]
        return times

    def per_rank(value):
        """[value of rank 0, .., value of rank N-1] on every rank (what a first scaling curve is read with: which rank is slow)."""
        if not grouped:
            return [float(value)]
        t = torch.zeros(world, dtype=torch.float64, device=dev if args.dist_backend == 'nccl' else 'cpu')
        t[rank] = float(value)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    head = dict(h=args.height, w=args.width, grid=args.grid, semantics=args.semantics, strategy=args.semantic_strategy,
                seg_engine=args.seg_engine, mode=args.mode, arith=args.arith, n_classes=args.n_classes)
    default_flags = (not args.train) and head == dict(h=240, w=320, grid=256, semantics=False, strategy='gt', seg_engine='hip',
                                                       mode='fast', arith='f16x3', n_classes=30)
    metric = 'frames/sec fused (%dx%d, %d^3 grid)' % (args.width, args.height, args.grid)
    total = args.warmup + args.steps * args.repeats
    if args.train:
        # (--dist-backend gloo: a dry run of the training schedule on one device - launch, barrier, accumulation boundary,
        # max over ranks; FlatGradientAllReduce stages the gradient buffer through the host for gloo, 'allreduce_backend' says so)
        tc = TrainCase(args.height, args.width, args.grid, dev, rank, total)
        res = tc.run(args.steps, args.warmup, sync, args.repeats)
        own = sorted(res['times'])[len(res['times']) // 2]
        rank_ms = [1e3 * t / args.steps for t in per_rank(own)]      # each rank's own median loop time per frame
        rank_host = per_rank(tc.host_ms_per_frame)
        rank_ar = per_rank(res['allreduce_us'] or 0.0)
        res['times'] = max_over_ranks(res['times'])
        if rank == 0:
            r = train_report(tc, res, args.steps, args.warmup, world, args.height, args.width, args.grid, grouped=grouped)
            r['per_rank'] = {'ms_per_step': rank_ms, 'host_loop_ms_per_frame': rank_host, 'allreduce_us': rank_ar,
                             'ms_per_step_min': min(rank_ms), 'ms_per_step_max': max(rank_ms)}
            print(json.dumps({'metric': metric + ', training frame step', 'value': r['value'], 'unit': 'frames/sec', 'n_gpus': world,
                              'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_step'], 'higher_is_better': True,
                              'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16x3', 'data': 'synthetic',
                              'config': {'workload': r['workload'], 'frame': [args.height, args.width], 'grid': args.grid,
                                         'parallelism': 'scene-sharded x%d, gradient all-reduce every %d frames' % (world, tc.accum),
                                         'cpu_affinity': affinity or 'not pinned'},
                              **{k: r[k] for k in ('repeats', 'value_min', 'value_max', 'allreduce_us', 'allreduce_calls_in_timed_region',
                                                   'allreduce_backend', 'gradient_bytes', 'gradients_finite', 'host_loop_ms_per_frame',
                                                   'host_bound_ratio', 'net_launches_per_pass', 'per_rank')}}))
        if grouped:
            torch.distributed.destroy_process_group()
        return
    if args.scenes > 1:  # aggregate of S scenes on one GPU (a secondary measurement: the headline stays one scene per GPU)
        mc = ManyCase(head, dev, rank, total, args.scenes)
        r = mc.run(args.steps, args.warmup, sync, args.repeats)
        if rank == 0:
            print(json.dumps(dict(r, metric=metric + ', %d scenes per GPU (fuse_many)' % args.scenes, n_gpus=world)))
        if grouped:
            torch.distributed.destroy_process_group()
        return
    if args.lookahead > 1:
        case = Case(head, dev, rank, total + args.lookahead)
        r = run_lookahead(case, args.steps, args.warmup, sync, args.lookahead, args.repeats, prefetch=not args.no_prefetch)
        if rank == 0:
            print(json.dumps(dict(r, metric=metric + ', %d-frame look-ahead of the 2-D network (fuse_sequence)' % args.lookahead, n_gpus=world)))
        if grouped:
            torch.distributed.destroy_process_group()
        return
    case = Case(head, dev, rank, total)
    if args.lean:
        res = case.run(args.steps, args.warmup, sync, profile_frames=0, kernel_reps=0, repeats=1)
        if rank == 0:
            r = report(case, res, args.steps, args.warmup, world, full=False)
            print(json.dumps({'metric': metric, 'value': r['value'],
                              'unit': 'frames/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_step'],
                              'lean': True, 'frames_fused': args.steps + args.warmup, 'stages_ms': r['stages_ms']}))
        if grouped:
            torch.distributed.destroy_process_group()
        return
    res = case.run(args.steps, args.warmup, sync, repeats=args.repeats)
    res['times'] = max_over_ranks(res['times'])

    if rank == 0:
        r = report(case, res, args.steps, args.warmup, world)
        cfg = case.cfg
        out = {
            'metric': metric, 'value': r['value'], 'unit': 'frames/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': ARITH[args.arith][0], 'data': 'synthetic',
            'config': {'workload': r['workload'], 'frame': [args.height, args.width], 'grid': args.grid,
                       'n_points': cfg.FUSION_MODEL.n_points, 'n_tail_points': cfg.FUSION_MODEL.n_tail_points,
                       'integrate_mode': args.mode, 'volume_dtype': 'f16', 'net_arithmetic': ARITH[args.arith][1],
                       'parallelism': 'scene-sharded x%d' % world, 'cpu_affinity': affinity or 'not pinned'},
            'timing': 'median of %d back-to-back repeats of the %d-step timed loop (each bracketed by barrier + synchronize, max over ranks)'
                      % (args.repeats, args.steps),
        }
        for k in ('repeats', 'value_min', 'value_max', 'stages_ms', 'stage_samples_in_timed_region', 'stages_ms_all_frames',
                  'stages_all_frames_pass', 'net_launches_per_frame', 'kernels', 'segmentation', 'roofline', 'roofline_net', 'roofline_hbm'):
            if k in r:
                out[k] = r[k]
        del case
        torch.cuda.empty_cache()
        n_sec = args.secondary if args.secondary is not None else (60 if (world == 1 and default_flags) else 0)
        if world == 1 and n_sec > 0:
            secondary = []
            for extra in (dict(arith='f32'),
                          dict(semantics=True),
                          dict(semantics=True, strategy='predict'),
                          dict(h=480, w=640, grid=512),
                          dict(h=480, w=640, grid=512, semantics=True, n_classes=40)):
                c2 = dict(head, **extra)
                try:
                    case2 = Case(c2, dev, rank, 3 * n_sec + 10)
                    r2 = report(case2, case2.run(n_sec, 10, sync, profile_frames=16, repeats=3), n_sec, 10, 1, full=False)
                    del case2
                except Exception as e:  # a secondary workload must not take the headline line down with it
                    r2 = {'workload': workload_name(c2), 'error': repr(e)}
                torch.cuda.empty_cache()
                secondary.append(r2)
            for L in (4, 8):  # configs[2] with predicted labels, the 2-D network L frames ahead of the frame steps (one scene)
                try:
                    c2 = dict(head, semantics=True, strategy='predict')
                    n_la = (n_sec + L - 1) // L * L
                    case2 = Case(c2, dev, rank, 3 * n_la + 4 * L)
                    secondary.append(run_lookahead(case2, n_la, 2 * L, sync, L, 3))
                    del case2
                except Exception as e:
                    secondary.append({'workload': 'configs[2] predicted labels, %d-frame look-ahead' % L, 'error': repr(e)})
                torch.cuda.empty_cache()
            for extra, S in ((dict(), 2), (dict(), 4), (dict(semantics=True, strategy='predict'), 4)):
                c2 = dict(head, **extra)  # several scenes per GPU (fuse_many): aggregate frames/s, next to the S = 1 legs above
                try:
                    mc = ManyCase(c2, dev, rank, 3 * n_sec + 10, S)
                    r2 = mc.run(n_sec, 10, sync, 3)
                    del mc
                except Exception as e:
                    r2 = {'workload': workload_name(c2) + ' - %d scenes per GPU' % S, 'error': repr(e)}
                torch.cuda.empty_cache()
                secondary.append(r2)
            try:  # BASELINE configs[3]: the training frame step on this GPU (no collective at N = 1)
                # the same schedule `bench.py --train --steps 48 --warmup 16 --repeats 3` runs: 16 warm-up frames = two optimizer
                # steps (the executor's second-pass arithmetic, packed weights and allocator pools settle inside them: with 8
                # warm-up frames this leg read 14 % below --train in round 4), 48 timed frames, median of 3
                n_tr, w_tr = max(48, n_sec // 8 * 8), 16
                tc = TrainCase(240, 320, 256, dev, rank, w_tr + 3 * n_tr)
                secondary.append(train_report(tc, tc.run(n_tr, w_tr, sync, 3), n_tr, w_tr, 1, 240, 320, 256))
                del tc
            except Exception as e:
                secondary.append({'workload': 'BASELINE configs[3]: training frame step', 'error': repr(e)})
            torch.cuda.empty_cache()
            secondary.append(rccl_world1_probe())
            out['secondary'] = secondary
        if world == 1 and args.cpu_frames > 0:
            out['cpu_baseline'] = cpu_baseline(args, args.height, args.width, args.grid, args.semantics)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if grouped:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
