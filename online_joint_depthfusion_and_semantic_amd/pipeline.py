"""Pipeline: per-frame orchestration with the reference's API (modules/pipeline.py:12-363).

``fuse`` / ``fuse_training`` keep the reference's signatures, batch-dict keys and side effects on
``Database``; the work is done by three HIP stages on the current stream with no host round trip
in between:

    ojf_extract  -> fusion_values / fusion_weights rows [N,9] (kept: they are part of the API)
    ojf_net_*    -> fp32-MFMA FusionNet (eval mode, BN folded)            [inference]
                    torch autograd on the same device                     [training, needs grads]
    ojf_integrate-> recomputes indices/weights, resolves colliding voxel writes, updates the
                    fp16 volumes in place (and the semantic ids/scores in test mode)

The reference's intermediate tensors (int64 indices [N,9,8,3], fp64 weights, NCHW permutes,
fancy-index slices of valid pixels: pipeline.py:74-171) never exist here.
"""
import torch

from . import ops
from . import _lib
from ._lib import MODE_FAST, MODE_PARITY
from .engine import FusionNetEngine
from .extractor import Extractor
from .integrator import Integrator
from .model import FusionNet_v2, FusionNet_v3


class Pipeline(torch.nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_points = config.FUSION_MODEL.n_points
        if config.DATA.semantics:
            self.n_classes = config.SEMANTIC_2D_MODEL.n_classes
        config.FUSION_MODEL.resx = config.DATA.resx  # pipeline.py:24-25
        config.FUSION_MODEL.resy = config.DATA.resy

        name = config.FUSION_MODEL.name
        if name == 'v2':
            self._fusion_network = FusionNet_v2(config.FUSION_MODEL)
        elif name == 'v3':
            self._fusion_network = FusionNet_v3(config.FUSION_MODEL)
        else:  # the reference's v1 cannot even be constructed (model.py:58)
            raise ValueError('FUSION_MODEL.name must be "v2" or "v3"')

        if config.DATA.semantics and config.DATA.semantic_strategy == 'predict':
            from .adapnet import AdapNet
            self._semantic_2d_network = AdapNet(config.SEMANTIC_2D_MODEL)
        else:
            self._semantic_2d_network = None

        self._extractor = Extractor(config)
        self._integrator = Integrator(config)
        # a load_state_dict into the fusion net (any flavour, assign=True included) drops the cached tensor list of the
        # engine-staleness check at once
        def _drop_fingerprint(module, incompatible):
            self.__dict__.pop('_fp_cache', None)  # (a post hook must return None)
        self._fusion_network.register_load_state_dict_post_hook(_drop_fingerprint)
        mode = getattr(config.SETTINGS, 'integrate_mode', 'fast')
        self._integrate_mode = MODE_PARITY if mode == 'parity' else MODE_FAST
        # per-slot device state (engine with its activation buffers, est rows, sample planes, integrate workspaces): slot 0
        # is the frame step of fuse() / fuse_training(); fuse_many() runs scene i of a call on slot i and its own stream
        self._slots = {}
        self._workspaces = {}
        self.profile = False  # True / N: fuse() brackets its three stages with HIP events on every (N-th) frame
        self._frames_fused = 0
        self._marks = []

    # ---- live stage timing (HIP events on the launch stream; bench.py) ---------------------------
    def _mark(self, first=False):
        """Even a fence-free event record is a marker packet in the queue (~4.5 us of idle queue each, four per frame):
        ``profile = N`` samples every N-th frame so that the timing does not change what it times."""
        if first:
            self._frames_fused += 1
        if self.profile and (self.profile is True or self._frames_fused % int(self.profile) == 0):
            e = _lib.TimingEvent()  # no system-scope fence: a default event costs ~5 us of idle queue per record
            e.record(torch.cuda.current_stream(self.device).cuda_stream)
            self._marks.append(e)

    def _mark_segmentation(self):
        """Event in front of the 2-D segmentation of a frame the stage sampling will pick (predict strategy only)."""
        if not self.profile or not self.config.DATA.semantics or self.config.DATA.semantic_strategy != 'predict':
            return None
        if self.profile is not True and (self._frames_fused + 1) % int(self.profile) != 0:
            return None
        e = _lib.TimingEvent()
        e.record(torch.cuda.current_stream(self.device).cuda_stream)
        return e

    def reset_profile(self):
        self._marks = []
        self._seg_marks = []

    def stage_times_ms(self):
        """Mean milliseconds per frame of extract / net / integrate over the frames fused since
        reset_profile() (4 events per frame)."""
        torch.cuda.synchronize(self.device)
        ev = self._marks
        n = len(ev) // 4
        if n == 0:
            return {}
        acc = [0.0, 0.0, 0.0]
        for f in range(n):
            for j in range(3):
                acc[j] += ev[4 * f + j].elapsed_time(ev[4 * f + j + 1])
        out = {'extract': acc[0] / n, 'net': acc[1] / n, 'integrate': acc[2] / n}
        seg = self.__dict__.get('_seg_marks') or []
        if seg:  # AdapNet++ prediction of the frame's labels (+ the host work between it and the extract launch)
            out['segmentation'] = sum(a.elapsed_time(b) for a, b in seg) / len(seg)
        return out

    def check(self):
        """Synchronise and raise OjfError if the split-fp16 range guard fired since the last check (the frames
        fused in between are invalid then; set FUSION_MODEL.arithmetic = 'f32').  Cheap: call per scene / epoch."""
        if self._engine is not None:
            if self._guard_policy() == 'f32':
                self._guard_recover()  # (a tripped frame at the end of a scene: nothing after it polled the flag)
            torch.cuda.synchronize(self.device)  # (every slot's stream)
            self._engine.check()
        elif torch.device(getattr(self, 'device', 'cpu')).type == 'cuda':
            # no inference engine yet (training only, or only the 2-D engine ran): the executors share the one flag
            torch.cuda.synchronize(self.device)
            _lib.check(_lib.load().ojf_net_check(_lib.stream_ptr(self.device)), 'ojf_net_check')

    # ---- cached device objects ----------------------------------------------------------------
    def _weights_fingerprint(self):
        """In-place weight updates (optimizer steps, load_state_dict) bump the tensors' version counters.  Walking
        the module tree costs ~0.3 ms, so the tensor list is cached and re-collected every 64 frames."""
        net = self._fusion_network
        cache = self.__dict__.get('_fp_cache')
        if cache is None or cache[0] != id(net) or cache[2] >= 64:
            cache = [id(net), list(net.parameters()) + list(net.buffers()), 0]
            self.__dict__['_fp_cache'] = cache
        cache[2] += 1
        # version counters see in-place updates (optimizer steps, load_state_dict); storage addresses see
        # ``p.data = ...``; replaced Parameter objects (load_state_dict(assign=True), swapped sub-modules) are seen by
        # the load_state_dict hook below at once and by the re-collection within 64 frames otherwise
        return (len(cache[1]), sum(t._version for t in cache[1]), sum(t.data_ptr() for t in cache[1]) & 0xffffffffffff)

    class _Slot:
        engine = key = est = fv = fw = None

    # slot 0 under the names the rest of the package (and bench.py) knows
    _engine = property(lambda self: self._slots[0].engine if 0 in self._slots else None)
    _est = property(lambda self: self._slots[0].est if 0 in self._slots else None)
    _fv = property(lambda self: self._slots[0].fv if 0 in self._slots else None)
    _fw = property(lambda self: self._slots[0].fw if 0 in self._slots else None)

    def _get_slot(self, h, w, device, slot=0, fingerprint=None):
        arith = self.config.FUSION_MODEL.get('arithmetic', 'f16x3')
        key = (h, w, str(device), arith, fingerprint if fingerprint is not None else self._weights_fingerprint())
        sl = self._slots.get(slot)
        if sl is None:
            sl = self._slots[slot] = Pipeline._Slot()
        if sl.engine is None or sl.key != key:
            if sl.engine is not None:
                sl.engine.close()
            sl.engine = FusionNetEngine(self._fusion_network, h, w, device, arithmetic=arith)
            sl.key = key
            sl.est = torch.empty((h * w, self.n_points), dtype=torch.float32, device=device)
            sl.fv = torch.empty((self.n_points, h * w), dtype=torch.float32, device=device)  # sample planes
            sl.fw = torch.empty((self.n_points, h * w), dtype=torch.float32, device=device)
        return sl

    def _get_engine(self, h, w, device):
        return self._get_slot(h, w, device).engine

    def _get_workspace(self, shape, h, w, device, slot=0):
        key = (tuple(shape), h, w, self._integrate_mode, slot)
        if key not in self._workspaces:
            self._workspaces[key] = ops.IntegrateWorkspace(shape, h, w, self.config.FUSION_MODEL.n_tail_points,
                                                           self._integrate_mode, device)
        return self._workspaces[key]

    # ---- semantics front-end (pipeline.py:42-60, 181-193) --------------------------------------
    def _segmentation(self, data):
        inputs = {'image': (data['image'] / 255.0).to(self.device).float()}
        in_ = self.config.DATA.input
        if in_ != 'image':
            inputs[in_] = data[in_].repeat(1, 3, 1, 1).to(self.device).float()
        net = self._seg_engine(inputs['image'].shape) or self._semantic_2d_network
        if self.config.SEMANTIC_2D_MODEL.stage == 1:
            output = net(inputs[in_])
        else:
            output = net(inputs['image'], inputs[in_])
        logits = output if torch.is_tensor(output) else output[0]  # the engine returns the main head only
        return torch.softmax(logits, dim=1).permute(0, 2, 3, 1)

    def _segment_max(self, data):
        """``_segmentation(data).max(dim=-1)`` -> (scores, ids).  With the HIP engine the whole chain - image / 255,
        depth x 3, AdapNet++, softmax, max - is libojf launches (SegEngine.predict); ids come back as uint8."""
        image = data['image']
        engine = self._seg_engine(image.shape)
        if engine is None:
            return self._segmentation(data).max(dim=-1)
        in_ = self.config.DATA.input
        depth = data[in_].to(self.device).float() if in_ != 'image' else None
        h, w = image.shape[-2:]
        scores, ids = engine.predict(image.to(self.device).float(), depth)
        return scores.view(1, h, w), ids.view(1, h, w)

    def _seg_engine(self, shape):
        """The AdapNet++ convolutions on the SEGCONV HIP kernels (adapnet_engine.SegEngine) when the front-end runs
        inference on the GPU: ``SEMANTIC_2D_MODEL.engine: hip`` (default) | ``torch`` (module forward, MIOpen).
        Rebuilt when the parameters change (version counters, checked like the fusion net's)."""
        net = self._semantic_2d_network
        if (self.config.SEMANTIC_2D_MODEL.get('engine', 'hip') != 'hip' or net.training or torch.is_grad_enabled()
                or torch.device(self.device).type != 'cuda' or shape[0] != 1):
            return None
        if shape[-2] % 16 or shape[-1] % 16:
            if not self.__dict__.get('_warned_seg_fallback'):
                import warnings
                self.__dict__['_warned_seg_fallback'] = True
                warnings.warn('SEMANTIC_2D_MODEL.engine = hip needs frame sides that are multiples of 16 (the 1/16-resolution '
                              'stage of AdapNet++): %dx%d frames run on the torch module forward instead' % (shape[-2], shape[-1]),
                              RuntimeWarning)
            return None
        cache = self.__dict__.get('_seg_cache')
        if cache is None or cache['net'] != id(net) or cache['age'] >= 64:
            tensors = list(net.parameters()) + list(net.buffers())
            cache = {'net': id(net), 'tensors': tensors, 'age': 0, 'key': None, 'engine': None} if cache is None or cache['net'] != id(net) \
                else dict(cache, tensors=tensors, age=0)
            self.__dict__['_seg_cache'] = cache
        cache['age'] += 1
        key = (len(cache['tensors']), sum(t._version for t in cache['tensors']), str(self.device))
        if cache['engine'] is None or cache['key'] != key:
            from .adapnet_engine import SegEngine
            cache['engine'], cache['key'] = SegEngine(net), key
        return cache['engine']

    def _segmentation_graph(self, data):
        """Inference-time replay of ``_segmentation(...).max(-1)``: captured once per frame shape (and engine) into a
        device graph.  320x240: torch module 772 launches, 9.1 ms eager / 4.5 ms replayed; SEGCONV engine 215 launches,
        2.6 ms eager / 2.3 ms replayed.  Parameters are read in place, so ``load_state_dict`` /
        optimizer steps are seen; ``SEMANTIC_2D_MODEL.graph: False`` or a failed capture falls back to eager."""
        image = data['image']
        in_ = self.config.DATA.input
        depth = data[in_] if in_ != 'image' else None
        with torch.no_grad():
            engine = self._seg_engine(image.shape)  # a rebuilt engine owns new weight buffers: recapture
        key = (tuple(image.shape), str(self.device), id(self._semantic_2d_network), id(engine))
        st = self.__dict__.get('_seg_graph')
        if st is None or st['key'] != key:
            st = {'key': key, 'graph': None}
            self.__dict__['_seg_graph'] = st
            try:
                st['image'] = torch.zeros(image.shape, dtype=image.dtype, device=self.device)
                st['depth'] = torch.zeros(depth.shape, dtype=depth.dtype, device=self.device) if depth is not None else None
                batch = {'image': st['image']}
                if depth is not None:
                    batch[in_] = st['depth']
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):  # library workspaces and autotuning must be settled before the capture
                    for _ in range(3):
                        self._segment_max(batch)
                torch.cuda.current_stream(self.device).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    st['scores'], st['ids'] = self._segment_max(batch)
                st['graph'] = graph
            except Exception:  # capture is an optimisation only
                st['graph'] = None
        if st['graph'] is None:
            return self._segment_max(data)
        st['image'].copy_(image, non_blocking=True)
        if depth is not None:
            st['depth'].copy_(depth.reshape(st['depth'].shape), non_blocking=True)
        st['graph'].replay()
        return st['scores'], st['ids']

    def _frame_semantics(self, batch):
        if not self.config.DATA.semantics:
            return None, None
        strategy = self.config.DATA.semantic_strategy
        if strategy == 'predict':
            with torch.no_grad():
                use_graph = (self.config.SEMANTIC_2D_MODEL.get('graph', True) and not self._semantic_2d_network.training
                             and torch.device(self.device).type == 'cuda')
                if use_graph:
                    scores, sem_ids = self._segmentation_graph(batch)
                else:
                    scores, sem_ids = self._segment_max(batch)
        elif strategy == 'gt':
            sem_ids = batch['semantic_gt']
            scores = torch.ones_like(sem_ids, dtype=torch.float32)
        else:
            raise ValueError('Valid values for DATA.semantic_strategy are "gt" or "predict".')
        h, w = sem_ids.shape[-2:]
        sem_ids = sem_ids.to(self.device).to(torch.uint8).reshape(h * w).contiguous()
        scores = scores.to(self.device).float().reshape(h * w).contiguous()
        return sem_ids, scores

    def fuse_sequence(self, batches, database, device, prefetch=None):
        """``for b in batches: fuse(b, database, device)`` for CONSECUTIVE frames (of one scene, or any mix of scenes in stream
        order).  The frame steps stay strictly in order - frame t+1 is extracted from the volume frame t was integrated into -
        but the 2-D network does not depend on the volumes: with ``semantic_strategy: predict`` the labels of all the frames
        are predicted first, as ONE batched AdapNet++ pass (SegEngine.predict_many: 0.62 / 0.48 ms per frame at four / eight
        frames per pass against 1.20 ms one at a time).  Same volumes as the separate calls, with the batched
        pass's rounding of the scores (include/ojf.h, ojf_segconv_forward_batch).  The reference predicts and fuses one frame
        at a time (test_fusion.py:68-80); a driver that reads a recorded stream can hand over chunks (drivers.test_fusion does).

        ``prefetch``: the chunk the NEXT call will bring (the same batch objects).  Its batched 2-D pass is enqueued on a
        side stream before this chunk's frame steps and runs beside them; the next call finds the labels ready.  Bits do not
        depend on it (the same graph replay, another stream)."""
        self.device = torch.device(device)
        sems = self._take_prefetched(batches)
        if sems is None:
            sems = self._frame_semantics_many(batches)
        if prefetch and self._batched_2d_pass_applies(prefetch):
            if sems and sems[0][0] is not None and not self.__dict__.get('_prefetch', {}).get('taken'):
                sems = [(i.clone(), sc.clone()) for i, sc in sems]  # (this chunk's labels sit in the graph's output buffers, which the prefetched pass overwrites)
            self._prefetch_semantics(prefetch)
        fp = self._weights_fingerprint()
        recover = self._guard_policy() == 'f32'
        for b, sem in zip(batches, sems):
            self._fuse_guarded(b, database, sem, fp, recover)

    def _fuse_guarded(self, batch, database, sem=None, fingerprint=None, recover=False):
        """One frame step of fuse_sequence / fuse_many's sequential paths under FUSION_MODEL.guard_policy (what ``fuse`` does
        for its one frame): with 'f32' a fired range guard switches the arithmetic, fuses the skipped frames again and goes
        on with this one; the frame joins the ring of remembered batches.  ``sem`` (the frame's labels from a batched 2-D
        pass) does not depend on the fusion net's arithmetic and is used either way; the re-fused frames predict theirs
        again, one frame at a time."""
        if not recover:
            return self._fuse_frame(batch, database, 0, sem, fingerprint)
        try:
            self._fuse_frame(batch, database, 0, sem, fingerprint)
        except _lib.OjfError as err:
            if 'fp16 range' not in str(err):
                raise
            self._guard_recover(err)
            self._fuse_frame(batch, database, 0, sem)  # (fp32 engine: a new fingerprint)
            return
        self._guard_remember(batch, database)

    def _batched_2d_pass_applies(self, batches):
        cfg = self.config
        return bool(cfg.DATA.semantics and cfg.DATA.semantic_strategy == 'predict' and len(batches) > 1
                    and not self._semantic_2d_network.training and cfg.SEMANTIC_2D_MODEL.get('stage', 2) != 1)

    def _side_stream(self, others):
        """A torch stream whose kernels really run beside those of ``others``: the runtime backs all streams by a small
        round-robin pool of hardware queues, and two streams on one queue take turns (measured: the look-ahead pass beside
        the frame steps 933 frames/s on two queues, 775 on one).  Up to eight candidates are tried (ojf_streams_overlap, a
        set-up cost of ~0.5 ms each); the rejected ones stay referenced until the choice is made, so that the next candidate
        is another pool member."""
        lib = _lib.load()
        rejected, best = [], None
        for _ in range(8):
            cand = torch.cuda.Stream(device=self.device)
            if all(lib.ojf_streams_overlap(o if isinstance(o, int) else o.cuda_stream, cand.cuda_stream) == 1 for o in others):
                return cand
            best = best or cand
            rejected.append(cand)
        return best

    def _measured_side_stream(self, cur, frames):
        """The look-ahead stream by measurement: which hardware queue a new stream gets, and whose packets it then waits
        behind, is the runtime's business (a stream that passes ojf_streams_overlap against every stream we know of still ran
        the pass 20 % slower than its neighbour in the pool: 780 against 940 frames/s at four frames per chunk).  Six candidate
        streams each replay the batched 2-D pass beside ``frames`` forward passes of the fusion net on the caller's stream
        (the executor recomputes its last frame's estimate: no volume is touched); the fastest is kept.  ~30 ms, once."""
        st = self.__dict__.get('_seg_graph_many')
        eng, est = getattr(self, '_engine', None), getattr(self, '_est', None)
        if not st or st.get('graph') is None or eng is None or est is None:
            return None
        best, best_ms, cands = None, None, []
        with torch.no_grad():
            for _ in range(6):
                cand = torch.cuda.Stream(device=self.device)
                cands.append(cand)  # (kept alive: the next candidate is another member of the pool)
                ms = []
                for _rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cand.wait_stream(cur)
                    with torch.cuda.stream(cand):
                        st['graph'].replay()
                    for _f in range(frames):
                        eng.forward(est)
                    cur.wait_stream(cand)
                    e1.record(cur)
                    e1.synchronize()
                    ms.append(e0.elapsed_time(e1))
                t = min(ms[1:])
                if best is None or t < best_ms:
                    best, best_ms = cand, t
        return best

    def _lookahead_stream_choice(self):
        """SETTINGS.lookahead_stream (env OJF_LOOKAHEAD_STREAM for A/B runs): 'measured' - six candidate streams timed once against
        the real work (round 5) | 'priority' - one high-priority stream (its own hardware-queue pool) | 'probe' - the first stream that
        passes ojf_streams_overlap."""
        import os
        return os.environ.get('OJF_LOOKAHEAD_STREAM') or self.config.SETTINGS.get('lookahead_stream', 'measured')

    def _net_side_streams(self):
        """Raw handles of the fusion net's own side streams (the second head of the two-head net runs on one): the look-ahead
        pass must not share a hardware queue with them either (measured: 775 frames/s when it does, 930-1070 when not)."""
        eng = getattr(self, '_engine', None)
        try:
            return eng.side_streams() if eng is not None and hasattr(eng, 'side_streams') else []
        except Exception:
            return []

    def _prefetch_semantics(self, batches):
        """The batched 2-D pass of ``batches`` on the side stream; the labels land in one of two persistent result slots
        (the graph's own output buffers belong to the next replay).  Slot k % 2 was last read by the frame steps of the
        chunk two calls ago: they are in front of the side stream's wait on the caller's stream."""
        if not self._batched_2d_pass_applies(batches):
            return
        cur = torch.cuda.current_stream(self.device)
        pf = self.__dict__.get('_prefetch')
        how = self._lookahead_stream_choice()
        if pf is None:
            net_streams = self._net_side_streams()
            if how == 'priority':
                # a stream of ANOTHER priority class: the runtime keeps one pool of hardware queues per priority, so this stream never
                # shares a queue with the caller's (normal-priority) stream or the net's side streams - by construction, not by probing
                side = torch.cuda.Stream(device=self.device, priority=-1)
            else:
                side = self._side_stream([cur] + net_streams)
            pf = self.__dict__['_prefetch'] = {'stream': side, 'n': 0, 'slots': [None, None], 'ready': None, 'taken': False, 'measured': how != 'measured',
                                                'net_seen': bool(net_streams) or getattr(self, '_engine', None) is not None}
        elif not pf.get('measured') and getattr(self, '_engine', None) is not None and self.__dict__.get('_seg_graph_many', {}).get('graph') is not None:
            # (once, when both networks' launch sequences exist: the stream is chosen by what it does to the real work)
            old = pf['stream']
            pf['stream'] = self._measured_side_stream(cur, len(batches)) or old
            pf['stream'].wait_stream(old)
            pf['measured'] = True
        side = pf['stream']
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            sems = self._frame_semantics_many(batches)
            if sems[0][0] is None:
                return
            k = pf['n'] % 2
            slot = pf['slots'][k]
            if slot is None or len(slot) != len(sems) or slot[0][0].shape != sems[0][0].shape:
                side.synchronize(); cur.synchronize()  # (first use / another chunk shape: allocate the slot once, outside any overlap)
                slot = pf['slots'][k] = [(torch.empty_like(i), torch.empty_like(sc)) for i, sc in sems]
            for (di, dsc), (i, sc) in zip(slot, sems):
                di.copy_(i, non_blocking=True)
                dsc.copy_(sc, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        pf['n'] += 1
        # the announced batch OBJECTS are held (and compared with `is`): an id() of a dict the caller has dropped can be handed to a
        # new dict, and the stale labels of other frames would be fused silently
        pf['ready'] = {'batches': list(batches), 'sems': slot, 'event': ev}

    def _take_prefetched(self, batches):
        pf = self.__dict__.get('_prefetch')
        if pf:
            pf['taken'] = False
        if not pf or not pf['ready']:
            return None
        ready, pf['ready'] = pf['ready'], None
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready['event'])  # (also orders this call's own 2-D pass, if the chunk is another one, behind the side stream's use of the graph buffers)
        pf['taken'] = len(ready['batches']) == len(batches) and all(a is b for a, b in zip(ready['batches'], batches))
        pf['hits'] = pf.get('hits', 0) + int(pf['taken'])
        return ready['sems'] if pf['taken'] else None

    def _frame_semantics_many(self, batches):
        """``[_frame_semantics(b) for b in batches]``; with ``semantic_strategy: predict`` on the HIP engine the S frames go
        through the 2-D network as ONE batched pass (SegEngine.predict_many: every layer's weights fetched once for all
        frames), replayed from a device graph per (S, frame shape)."""
        cfg = self.config
        if (not cfg.DATA.semantics or cfg.DATA.semantic_strategy != 'predict' or len(batches) == 1
                or self._semantic_2d_network.training or cfg.SEMANTIC_2D_MODEL.get('stage', 2) == 1):
            return [self._frame_semantics(b) for b in batches]
        with torch.no_grad():
            engine = self._seg_engine(batches[0]['image'].shape)
            if engine is None:
                return [self._frame_semantics(b) for b in batches]
            in_ = cfg.DATA.input
            images = [b['image'].to(self.device).float() for b in batches]
            depths = [b[in_].to(self.device).float() for b in batches] if in_ != 'image' else None
            S = len(batches)
            h, w = images[0].shape[-2:]
            key = (S, tuple(images[0].shape), None if depths is None else tuple(depths[0].shape), str(self.device), id(engine))
            st = self.__dict__.get('_seg_graph_many')
            if st is None or st['key'] != key:
                st = self.__dict__['_seg_graph_many'] = {'key': key, 'graph': None}
                if cfg.SEMANTIC_2D_MODEL.get('graph', True):
                    try:
                        st['images'] = [torch.zeros_like(im) for im in images]
                        st['depths'] = [torch.zeros_like(d) for d in depths] if depths is not None else None
                        side = torch.cuda.Stream(device=self.device)
                        side.wait_stream(torch.cuda.current_stream(self.device))
                        with torch.cuda.stream(side):
                            for _ in range(2):
                                engine.predict_many(st['images'], st['depths'])
                        torch.cuda.current_stream(self.device).wait_stream(side)
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            st['scores'], st['ids'] = engine.predict_many(st['images'], st['depths'])
                        st['graph'] = graph
                    except Exception:  # capture is an optimisation only
                        st['graph'] = None
            if st['graph'] is None:
                scores, ids = engine.predict_many(images, depths)
            else:
                for dst, src in zip(st['images'], images):
                    dst.copy_(src, non_blocking=True)
                if depths is not None:
                    for dst, src in zip(st['depths'], depths):
                        dst.copy_(src.reshape(dst.shape), non_blocking=True)
                st['graph'].replay()
                scores, ids = st['scores'], st['ids']
            return [(ids[i].contiguous(), scores[i].contiguous()) for i in range(S)]

    def _frames(self, batch, filtered=True):
        """(frame, filtered frame) of pipeline.py:194-199; with ``filtered=False`` the second item is the bool mask
        instead and the integrate kernels apply it themselves (ojf_integrate_masked)."""
        frame = batch[self.config.DATA.input]
        frame = frame.reshape(frame.shape[0], frame.shape[-2], frame.shape[-1])  # squeeze_(1) of [b,1,h,w]
        if frame.shape[0] != 1:
            raise ValueError('Pipeline: batch size 1 only (one scene per frame, pipeline.py:199)')
        frame = frame.to(self.device).float().contiguous()
        mask = batch['mask'].to(self.device).reshape(frame.shape)
        if not filtered:
            if mask.dtype != torch.bool:
                mask = mask != 0
            return frame[0], mask[0].contiguous()
        zero = self.__dict__.get('_zero')
        if zero is None or zero.device != frame.device:
            zero = self.__dict__['_zero'] = torch.zeros((), dtype=torch.float32, device=frame.device)
        filtered = torch.where(mask, frame, zero)  # pipeline.py:196; one launch (a python scalar costs a fill kernel)
        return frame[0], filtered[0]

    # ---- range guard of the split-fp16 net: no frame of a tripped net reaches a volume --------------
    # (include/ojf.h ojf_net_check: while the flag is set the integrate calls skip, so nothing is corrupted; what is left
    # to decide is what happens to the skipped frames.)  FUSION_MODEL.guard_policy:
    #   'raise' (default)  the next fuse() / check() raises OjfError; the volumes hold every frame in front of the event
    #                      and nothing after it;
    #   'f32'              the pipeline switches this network to the fp32-input MFMA arithmetic for good (a warning says
    #                      so) and fuses the skipped frames again, in order: the stream comes out as if the net had run in
    #                      fp32 from the event on.  The batches of the last <= 48 frames stay alive for that (an event
    #                      every 16 frames bounds how far the host may run ahead of the device).
    _GUARD_EVERY, _GUARD_KEEP = 16, 48

    def _guard_policy(self):
        if self.config.FUSION_MODEL.get('arithmetic', 'f16x3') != 'f16x3':
            return 'raise'
        return self.config.FUSION_MODEL.get('guard_policy', 'raise')

    def _guard_remember(self, batch, database):
        ring = self.__dict__.setdefault('_guard_ring', [])
        marks = self.__dict__.setdefault('_guard_marks', [])
        ring.append((batch, database))
        if len(ring) % self._GUARD_EVERY == 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            marks.append([len(ring), ev])
        if len(ring) > self._GUARD_KEEP:
            upto, ev = marks.pop(0)
            ev.synchronize()  # (long done unless the host is > 32 frames ahead of the device)
            if not _lib.load().ojf_guard_poll():  # the frames in front of that event were fused with the guard silent
                del ring[:upto]
                for m in marks:
                    m[0] -= upto

    def _guard_recover(self, err=None):
        """guard_policy 'f32': if the guard fired - report + clear the event, switch the arithmetic, fuse the skipped frames
        again.  Synchronises.  Returns True when it did."""
        import ctypes
        import warnings
        lib = _lib.load()
        flag, skipped = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(lib.ojf_guard_status(_lib.stream_ptr(self.device), ctypes.byref(flag), ctypes.byref(skipped)), 'ojf_guard_status')
        if flag.value != 1:  # nothing pending, or an internal error: not this policy's to absorb
            if err is not None:
                raise err
            return False
        lib.ojf_net_check(_lib.stream_ptr(self.device))  # (reports what `flag` said and clears it)
        ring = self.__dict__.get('_guard_ring', [])
        n = skipped.value
        if n > len(ring):
            raise _lib.OjfError('range guard: %d frames were skipped but only %d are remembered' % (n, len(ring)))
        redo = ring[len(ring) - n:] if n else []
        self.__dict__['_guard_ring'], self.__dict__['_guard_marks'] = [], []
        self.config.FUSION_MODEL.arithmetic = 'f32'
        self.guard_events = self.__dict__.get('guard_events', 0) + 1
        warnings.warn('split-fp16 range guard fired: FUSION_MODEL.arithmetic switched to f32 for this network, '
                      '%d skipped frame(s) fused again' % n, RuntimeWarning)
        for b, db in redo:
            self._fuse_frame(b, db)
        return True

    def fuse(self, batch, database, device):
        self.device = torch.device(device)
        # (a frame the forward call's poll refuses was raised BEFORE its net / integrate were enqueued: it is not among the
        # skipped ones, it follows them - _fuse_guarded)
        return self._fuse_guarded(batch, database, recover=self._guard_policy() == 'f32')

    def _fuse_frame(self, batch, database, slot=0, semantics=None, fingerprint=None):
        """One frame step on the current stream with the device state of ``slot``; ``semantics`` = (sem_ids, scores) computed
        by the caller (fuse_many) instead of here."""
        self._shape = batch['image'].shape
        # The range-guard flag is process-wide and raised by EVERY split-fp16 executor (the 2-D engine's kernels always are,
        # whatever FUSION_MODEL.arithmetic says), while the integrate calls skip as long as it is set: a host read of the mapped
        # word here, in front of anything this frame enqueues, keeps a tripped 2-D pass from dropping frames silently when the
        # fusion net itself runs fp32 (whose forward call does not poll).
        if self.device.type == 'cuda' and _lib.load().ojf_guard_poll():
            raise _lib.OjfError('Pipeline.fuse: the split-fp16 range guard is set (fp16 range exceeded in a network pass since '
                                'the last check()): the integrate calls skip until check() has reported it')
        profiled = slot == 0 and semantics is None
        seg0 = self._mark_segmentation() if profiled else None
        sem_ids, scores = self._frame_semantics(batch) if semantics is None else semantics
        frame, mask = self._frames(batch, filtered=False)
        h, w = frame.shape

        scene_id = batch['frame_id'][0].split('/')[0]
        volume = database[scene_id]
        tsdf, weights = volume['current'], volume['weights']
        Ki, E = ops.camera_arrays(batch['intrinsics'][0], batch['extrinsics'][0])

        sl = self._get_slot(h, w, self.device, slot, fingerprint)
        eng = sl.engine
        P = self.n_points
        mark = self._mark if profiled else (lambda first=False: None)
        mark(first=True)
        if seg0 is not None and self._marks:
            self.__dict__.setdefault('_seg_marks', []).append((seg0, self._marks[-1]))
        use_sem = self.config.FUSION_MODEL.use_semantics
        if eng.fused_input:
            # geometry-only net: the extractor writes the net's input planes itself (one launch less, no sample planes)
            ops.extract_to_net(frame, Ki, E, volume['origin'], volume['resolution'], tsdf, weights, eng)
            mark()
        else:
            ops.extract(frame, Ki, E, volume['origin'], volume['resolution'], tsdf, weights, n_points=P,
                        out_values=sl.fv, out_weights=sl.fw, out_stride=h * w, planes=True)
            mark()
            eng.prepare_input(sl.fv, sl.fw, frame, sem_ids if use_sem else None, self.n_classes if use_sem else 0,
                              planes=True)
        eng.forward(sl.est)
        mark()

        sem = bool(self.config.DATA.semantics)
        ws = self._get_workspace(tsdf.shape, h, w, self.device, slot)
        ops.integrate(frame, Ki, E, volume['origin'], volume['resolution'], sl.est, tsdf, weights, ws,
                      n_points=P, n_tail=self.config.FUSION_MODEL.n_tail_points,
                      trunc=self.config.DATA.init_value,
                      sem_ids=sem_ids if sem else None, sem_scores=scores if sem else None,
                      id_vol=volume['ids_est'] if sem else None, score_vol=volume['scores'] if sem else None,
                      mode=self._integrate_mode, mask=mask)  # filtered frame of pipeline.py:196 formed in the kernels
        mark()

        database.state[scene_id] = True  # volumes were updated in place (pipeline.py:239-244)
        database.scenes_est[scene_id].volume = tsdf
        database.fusion_weights[scene_id] = weights
        return

    # ---- several scenes per call (SURVEY.md §7 "hard parts") ------------------------------------------
    def fuse_many(self, batches, database, device):
        """``for b in batches: fuse(b, database, device)`` for frames of DISTINCT scenes (one frame each, one frame size),
        with the frame steps running side by side on the device: scene i of the call uses slot i - its own engine
        (activation buffers), est rows and integrate workspace - on its own stream, forked from and joined to the current
        stream.  The scenes' volumes do not overlap, every slot runs the kernels ``fuse`` runs with the same arguments, so the
        volumes come out bit for bit as from the separate calls; what changes is the schedule - the launches of one frame are
        latency-bound chains of 600-block kernels, and several of them fill the chip better (the two heads of a semantic net
        already run this way).  With ``semantic_strategy: predict`` the 2-D network of the S frames runs first, on the current
        stream (one engine).  The reference has no counterpart: its drivers fuse one frame at a time
        (test_fusion.py:68-80)."""
        self.device = torch.device(device)
        ids = [b['frame_id'][0].split('/')[0] for b in batches]
        if len(set(ids)) != len(ids):
            raise ValueError('Pipeline.fuse_many: one frame per scene (the frames of ONE scene depend on each other)')
        recover = self._guard_policy() == 'f32'
        if len(batches) == 1 or self.device.type != 'cuda':
            for b in batches:
                self._fuse_guarded(b, database, recover=recover)
            return
        if len(batches) > _lib.MAX_SCENES:  # (the launches take up to MAX_SCENES scenes: larger calls go in groups)
            for i in range(0, len(batches), _lib.MAX_SCENES):
                self.fuse_many(batches[i:i + _lib.MAX_SCENES], database, device)
            return
        main = torch.cuda.current_stream(self.device)
        sems = self._frame_semantics_many(batches)  # (None, None) without semantics; predict: ONE batched pass, this stream
        fp = self._weights_fingerprint()
        streams = self.__dict__.setdefault('_slot_streams', [])
        while len(streams) < len(batches) - 1:
            streams.append(self._side_stream([main] + streams))  # (a stream that runs beside the caller's and the other slots')
        enqueued = 0
        try:
            import os
            # 'auto' (default): the joint launches for two scenes, per-slot launches from three on - measured on one box (round 6,
            # profiles/r06_many_scene_launches.txt): joint 2246 against 2220 frames/s at S = 2, 2650 against 2747 at S = 4 (the two joins -
            # every net waits for the gather of ALL scenes, the scatter for ALL nets - cost more than the shared launches save)
            how = os.environ.get('OJF_FUSE_MANY') or self.config.SETTINGS.get('fuse_many_launches', 'auto')  # (env: A/B runs)
            if how == 'auto':
                how = 'joint' if len(batches) <= 2 else 'slots'
            if self._integrate_mode == MODE_FAST and how == 'joint':
                self._fuse_many_joint(batches, database, sems, fp, main, streams)
                enqueued = len(batches)
            else:  # PARITY integrate (its sort is per scene) / A-B switch: every slot runs fuse()'s own launches on its stream
                for i, (b, sem) in enumerate(zip(batches, sems)):
                    if i == 0:
                        self._fuse_frame(b, database, 0, sem, fp)
                    else:
                        st = streams[i - 1]
                        st.wait_stream(main)  # (the frame tensors and the labels were produced on the current stream)
                        with torch.cuda.stream(st):
                            self._fuse_frame(b, database, i, sem, fp)
                    enqueued = i + 1
        except _lib.OjfError as err:
            for st in streams[:len(batches) - 1]:
                main.wait_stream(st)
            if not recover or 'fp16 range' not in str(err):
                raise
            # guard_policy 'f32': the frames of this call that were enqueued behind the event are among the skipped ones
            # (remember them first); the one the poll refused and those after it follow, one at a time on the fp32 engine
            for b in batches[:enqueued]:
                self._guard_remember(b, database)
            self._guard_recover(err)
            for b, sem in zip(batches[enqueued:], sems[enqueued:]):
                self._fuse_guarded(b, database, sem, None, True)
            return
        for st in streams[:len(batches) - 1]:
            main.wait_stream(st)
        if recover:
            for b in batches:
                self._guard_remember(b, database)

    def _fuse_many_joint(self, batches, database, sems, fp, main, streams):
        """fuse_many's frame steps with the gather and the scatter of ALL scenes as single launches (round 6):

            ops.extract_many     one launch, blockIdx.y = scene       caller's stream
            S fusion nets        slot i on its own stream, forked from / joined to the caller's (as before)
            ops.integrate_many   accumulate + finalize of all scenes  caller's stream

        extract / accumulate / finalize are one round of blocks that each walk a chain of dependent steps (DESIGN.md 5.0c:
        0.017 of the HBM roofline, 70 % parked): S scenes in ONE launch share the ramp and keep the CUs busy while other
        blocks wait, which S launches on S streams do not.  Per scene the same blocks run the same code as in ``fuse``: the
        volumes come out bit for bit the same.  Raises (the poll below, a net's forward call) before ANY integrate launch of
        the call has been enqueued: none of the call's frames is among the skipped ones."""
        if _lib.load().ojf_guard_poll():
            raise _lib.OjfError('Pipeline.fuse_many: the split-fp16 range guard is set (fp16 range exceeded in a network pass since '
                                'the last check()): the integrate calls skip until check() has reported it')
        P, n_tail = self.n_points, self.config.FUSION_MODEL.n_tail_points
        use_sem, sem = self.config.FUSION_MODEL.use_semantics, bool(self.config.DATA.semantics)
        prep = []
        for i, (b, (sem_ids, scores)) in enumerate(zip(batches, sems)):
            self._shape = b['image'].shape
            frame, mask = self._frames(b, filtered=False)
            h, w = frame.shape
            scene_id = b['frame_id'][0].split('/')[0]
            volume = database[scene_id]
            Ki, E = ops.camera_arrays(b['intrinsics'][0], b['extrinsics'][0])
            sl = self._get_slot(h, w, self.device, i, fp)
            ws = self._get_workspace(volume['current'].shape, h, w, self.device, i)
            prep.append(dict(scene=scene_id, frame=frame, mask=mask, volume=volume, Ki=Ki, E=E, slot=sl, ws=ws, sem_ids=sem_ids, scores=scores))
        if len({(p['frame'].shape, p['volume']['current'].shape) for p in prep}) != 1:
            raise ValueError('Pipeline.fuse_many: one frame size and one grid size per call')
        fused = prep[0]['slot'].engine.fused_input
        self._mark(first=True)  # (profile: four events per CALL - gather of all scenes | nets | scatter of all scenes)
        ops.extract_many([dict(depth=p['frame'], Ki=p['Ki'], E=p['E'], origin=p['volume']['origin'], resolution=p['volume']['resolution'],
                               tsdf=p['volume']['current'], weights=p['volume']['weights'],
                               **(dict(engine=p['slot'].engine) if fused else dict(out_values=p['slot'].fv, out_weights=p['slot'].fw)))
                          for p in prep], n_points=P)
        self._mark()
        for i, p in enumerate(prep):
            sl = p['slot']
            if i:
                streams[i - 1].wait_stream(main)
            with torch.cuda.stream(streams[i - 1] if i else main):
                if not fused:
                    sl.engine.prepare_input(sl.fv, sl.fw, p['frame'], p['sem_ids'] if use_sem else None, self.n_classes if use_sem else 0, planes=True)
                sl.engine.forward(sl.est)
        for st in streams[:len(prep) - 1]:
            main.wait_stream(st)
        self._mark()
        ops.integrate_many([dict(depth=p['frame'], mask=p['mask'], Ki=p['Ki'], E=p['E'], origin=p['volume']['origin'],
                                 resolution=p['volume']['resolution'], est=p['slot'].est, tsdf=p['volume']['current'],
                                 weights=p['volume']['weights'], workspace=p['ws'],
                                 **(dict(sem_ids=p['sem_ids'], sem_scores=p['scores'], id_vol=p['volume']['ids_est'],
                                         score_vol=p['volume']['scores']) if sem else {}))
                            for p in prep], n_points=P, n_tail=n_tail, trunc=self.config.DATA.init_value)
        self._mark()
        self._frames_fused += len(prep) - 1  # (the first mark counted one)
        for p in prep:
            database.state[p['scene']] = True  # volumes were updated in place (pipeline.py:239-244)
            database.scenes_est[p['scene']].volume = p['volume']['current']
            database.fusion_weights[p['scene']] = p['volume']['weights']

    def _training_forward(self, inputs):
        """The net forward of pipeline.py:322 with a graph for ``loss.backward()``: on the libojf training kernels
        (train.HipTrainNet: ``FUSION_MODEL.train_engine: hip``, default on a GPU) or on torch autograd (``torch``)."""
        if self.config.FUSION_MODEL.get('train_engine', 'hip') == 'hip' and torch.device(self.device).type == 'cuda':
            tn = self.__dict__.get('_hip_train')
            if tn is None or tn.net is not self._fusion_network:
                from .train import HipTrainNet
                tn = self.__dict__['_hip_train'] = HipTrainNet(self._fusion_network, graph=self.config.FUSION_MODEL.get('train_graph', False),
                                                               inplace_grads=True, arithmetic=self.config.FUSION_MODEL.get('train_arithmetic', 'f16x3'),
                                                               backward_arithmetic=self.config.FUSION_MODEL.get('train_arithmetic_bwd', None),
                                                               replay=self.config.FUSION_MODEL.get('train_replay', None),
                                                               overlap=self.config.FUSION_MODEL.get('train_overlap', False),
                                                               overlap_thread=self.config.FUSION_MODEL.get('train_overlap_thread', False))
            return tn(inputs)
        return self._fusion_network.forward(inputs)

    def gradients(self):
        """``with pipeline.gradients():`` around everything between ``loss.backward()`` and the next ``fuse_training`` that touches the
        fusion net's gradients or writes its parameters (train_fusion.py:182-189: clip_grad_norm_, optimizer.step(), zero_grad()).  With
        ``FUSION_MODEL.train_overlap`` the backward pass of a frame runs on a stream of its own beside the next frame's forward stage and
        this context puts its body on that stream; without it (the default) the context does nothing."""
        tn = self.__dict__.get('_hip_train')
        if tn is not None:
            return tn.gradients()
        import contextlib
        return contextlib.nullcontext()

    def gradient_work(self, fn, join=True):
        """``fn()`` behind the backward passes enqueued so far - the function form of ``with pipeline.gradients(): ...`` that a host thread
        of the pipeline's own can run (``FUSION_MODEL.train_overlap_thread``, default off: measured neutral): the caller does not wait
        for the backward pass's launch loop.  ``join=False`` only for an ``fn`` that writes no parameter (train.HipTrainNet.gradient_work)."""
        tn = self.__dict__.get('_hip_train')
        if tn is not None:
            return tn.gradient_work(fn, join=join)
        fn()

    def join_gradients(self):
        """The current stream waits for the gradient stream (before checkpoints, validation, host reads of gradients or freshly stepped
        weights); a no-op without ``FUSION_MODEL.train_overlap``.  The next ``fuse_training`` joins by itself when a weight changed."""
        tn = self.__dict__.get('_hip_train')
        if tn is not None:
            tn.join_gradients()

    def _async_count(self, mask_flat):
        """Number of set elements of a device bool vector, on its way to the host: (pinned int32 buffer, event)."""
        if not mask_flat.is_cuda:
            return int(mask_flat.sum())
        slots = self.__dict__.setdefault('_count_slots', [])
        at = self.__dict__.get('_count_at', 0)
        if len(slots) < 4:  # a small ring: a slot is reused three frames later, long after its value was read
            slots.append((torch.empty(1, dtype=torch.int32).pin_memory(), torch.cuda.Event()))
        buf, ev = slots[at % len(slots)]
        self.__dict__['_count_at'] = at + 1
        buf.copy_(mask_flat.sum(dtype=torch.int32).reshape(1), non_blocking=True)
        ev.record(torch.cuda.current_stream(mask_flat.device))
        return buf, ev

    def _valid_index(self, mask_flat, count):
        """``mask.nonzero()[:, 0]`` without draining the stream: the size comes from _async_count."""
        if isinstance(count, int):
            return mask_flat.nonzero()[:, 0]
        buf, ev = count
        ev.synchronize()
        nv = int(buf[0])
        try:
            return torch.nonzero_static(mask_flat, size=nv)[:, 0]
        except (RuntimeError, NotImplementedError):  # builds without the device kernel: the blocking form
            return mask_flat.nonzero()[:, 0]

    # ---- training frame step (pipeline.py:251-363) -----------------------------------------------
    def announce_training_frame(self, batch, device=None):
        """Tells the pipeline which batch the NEXT ``fuse_training`` call will bring (the same object).  The part of that frame
        step that depends on the batch alone - the filtered frame of pipeline.py:196 and the number of valid rays, the one value
        the host must read to shape ``tsdf_fused`` / ``tsdf_target`` [1, Nv, 9] - is enqueued NOW, in front of the current
        frame's work, so that its answer is on the host long before the next call asks for it.  Without the announcement the count
        is requested at the start of its own frame: the read then waits until the device has drained the previous frame, which
        ties the host to the device once per frame - on a host that needs longer to enqueue a backward pass than the device
        needs for the forward pass, the device starves (round 6: 3.2 ms of host work and 4.25 ms of device work per frame took
        4.8-5.5 ms).  With it the host may run up to one frame ahead.  Optional and bit-neutral: the same launches, earlier.  The
        reference has no counterpart (its ``nonzero`` drains the queue in the middle of every frame, pipeline.py:125-131)."""
        if batch is None:
            self.__dict__['_announced'] = None
            return
        self.device = torch.device(device if device is not None else getattr(self, 'device', 'cpu'))
        frame, filtered = self._frames(batch)
        valid_mask = filtered.reshape(frame.numel()) != 0
        self.__dict__['_announced'] = (batch, frame, filtered, valid_mask, self._async_count(valid_mask))

    def fuse_training(self, batch, database, device):
        self.device = torch.device(device)
        self._shape = batch['image'].shape
        sem_ids, scores = self._frame_semantics(batch)
        ann = self.__dict__.pop('_announced', None)
        if ann is not None and ann[0] is batch and ann[1].device == self.device:
            _, frame, filtered, valid_mask, count = ann  # (enqueued a frame ago: the count is on the host by now)
        else:
            frame, filtered = self._frames(batch)
            valid_mask = count = None
        h, w = frame.shape
        n, P = h * w, self.n_points

        scene_id = batch['frame_id'][0].split('/')[0]
        volume = database[scene_id]
        tsdf, weights = volume['current'], volume['weights']
        Ki, E = ops.camera_arrays(batch['intrinsics'][0], batch['extrinsics'][0])

        # The shapes [1, Nv, P] of the masked outputs (pipeline.py:125-131) need the number of valid rays on the host: the one
        # device -> host read of the frame step.  The count is REQUESTED here (one reduction + an asynchronous copy into
        # pinned memory, behind an event) and awaited only after extract and the net's forward pass have been enqueued: by
        # then the copy is long done, the host never drains the queue, and it runs ahead of the device through loss and
        # backward.  (Read with a blocking ``nonzero`` after the net forward - the reference's order - the queue drained in
        # the middle of every frame and the device idled while Python enqueued the loss: 2.7 of 10.4 ms per 320x240 frame.)
        if valid_mask is None:
            valid_mask = filtered.reshape(n) != 0
            count = self._async_count(valid_mask)

        # sample planes [P, n] are NCHW [1, P, h, w] as they stand: no permute / contiguous copies in front of the net
        cur = ops.extract(frame, Ki, E, volume['origin'], volume['resolution'], tsdf, weights, n_points=P, planes=True)
        gt = ops.extract(frame, Ki, E, volume['origin'], volume['resolution'], volume['gt'], weights, n_points=P)
        fv, fw = cur['fusion_values'], cur['fusion_weights']
        inputs = {'tsdf_values': fv.view(1, P, h, w), 'tsdf_weights': fw.view(1, P, h, w), 'tsdf_frame': frame.view(1, 1, h, w)}
        if self.config.FUSION_MODEL.use_semantics:
            inputs['semantic_frame'] = ((1 + sem_ids.float()) / self.n_classes).view(1, 1, h, w)
        try:
            est_pn = self._training_forward(inputs)[:, :P].reshape(1, P, n)  # differentiable; BN / dropout follow the modules' modes
        except _lib.OjfError as err:
            # the executor's forward refuses to run while the shared range-guard flag is set and leaves it set (include/ojf.h):
            # report + clear it here, so that a training loop that catches the error and carries on (another arithmetic, the
            # next scene) does not have every later integrate call skip silently
            if 'fp16 range' in str(err) and self.device.type == 'cuda':
                _lib.load().ojf_net_check(_lib.stream_ptr(self.device))
            raise
        valid = self._valid_index(valid_mask, count)

        # pipeline.py:104-135 in the plane layout, one launch each way; the [1, n, P] tensor of the API is a transposed view
        init = self.config.DATA.init_value
        from .train import FuseOutput
        tsdf_fused = FuseOutput.apply(est_pn, fv.view(1, P, n), fw.view(1, P, n), valid, init)
        output = {'tsdf_est': est_pn.transpose(1, 2), 'tsdf_fused': tsdf_fused,
                  'tsdf_target': gt['fusion_values'].view(1, n, P)[:, valid, :]}

        ws = self._get_workspace(tsdf.shape, h, w, self.device)
        est_rows = est_pn.detach()[0].t().contiguous()
        ops.integrate(filtered, Ki, E, volume['origin'], volume['resolution'], est_rows, tsdf, weights, ws,
                      n_points=P, n_tail=self.config.FUSION_MODEL.n_tail_points, trunc=init,
                      mode=self._integrate_mode)  # test=False: no semantic update (pipeline.py:357)
        database.state[scene_id] = True
        database.scenes_est[scene_id].volume = tsdf
        database.fusion_weights[scene_id] = weights
        return output
