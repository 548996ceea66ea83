"""hipGraph execution of the head for fixed shapes.

At small batch the head is launch-bound (a few hundred kernels of a few microseconds each per frame), so the
whole ``forward`` + ``get_bboxes_padded`` sequence is captured once into a HIP graph (through
``torch.cuda.CUDAGraph``, which is hipGraph on ROCm) and replayed: every kernel of the path - vendor GEMMs /
convs and the hand-written ones, which enqueue on the capturing stream through the C ABI - becomes a graph
node; inputs are copied into static buffers, outputs are read from static buffers.  Nothing on the path
allocates outside the capture pool or synchronises with the host (see ff3d.h conventions).

Synchronisation (round 6: no discipline left).  Rounds 2-5 had to keep callers from the sequence [replay, eager launch,
torch.cuda.synchronize(), replay]: on ROCm 7.2 / torch 2.10 the replay after it died with a GPU memory fault.  Root cause
(profiles/r06_a_graph_fault_bisect.txt, tools/bisect_graph_fault.py): of ten launch families captured on their own only the one
whose capture held ``hipMemsetAsync`` calls - MEMSET NODES, the histogram / counter zero-fills of csrc/heatmap.hip - faults; the
faulting address lies outside every segment of torch's allocator (a buffer of the HIP runtime's own), and
``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` (the runtime's pre-built AQL packets for graph nodes switched off) makes the same graph safe.
The package now zero-fills with a kernel (``zero_u32``), so its graphs hold kernel nodes only (+ the all-gather's), and the head is
callable between arbitrary eager ops and host synchronisations, as the reference's is (focalformer3d.py:306-319):
tools/stress_replay_sync.py runs 100 x [replay, eager launch, torch.cuda.synchronize()] on GraphedHead / PipelinedHead at the bench
shape and on the captured neck + head (tests/test_round6_gpu.py).  The guard class of rounds 3-5 (``host_synced()``) is gone.  Advice
for code captured TOGETHER with the head into one graph: keep ``hipMemsetAsync`` out of the captured region on this ROCm (torch's own
``zero_()`` / ``fill_()`` are kernels and fine), or run with ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``.
``pack=True`` puts the detection packing into the graph too, so a serving step is one replay.
"""
import os

import torch


class NeckAndHead(torch.nn.Module):
    """``FocalEncoder`` -> ``FocalDecoder`` as ONE capturable unit (BASELINE configs[2]: camera maps + LiDAR BEV -> fusion neck ->
    head): ``unit([img_feats, [pts_feats]], None, None)`` = ``head(neck(img_feats, pts_feats, img_metas)[1], None, img_metas)``, so
    GraphedHead / PipelinedHead capture neck + head + get_bboxes + packing in one graph (rounds 1-4 ran the neck eagerly: 400
    dispatches per step).  The image metas (camera matrices) are fixed at construction, as the weights are: the projection
    sampler keeps the matrices on the device and re-uploads only when their values change (i2p.py), so a replay copies nothing
    from the host."""

    def __init__(self, neck, head, img_metas):
        super().__init__()
        self.neck, self.head, self.img_metas = neck, head, list(img_metas)

    def forward(self, inputs, img_inputs=None, img_metas=None):
        pts = inputs[1][0] if isinstance(inputs[1], (list, tuple)) else inputs[1]
        return self.head(self.neck(inputs[0], pts, self.img_metas)[1], None, self.img_metas)

    def get_bboxes_padded(self, preds, max_out=200):
        return self.head.get_bboxes_padded(preds, max_out=max_out)

    def num_frames(self, inputs):
        """Frames per batch (the camera maps come as (frames * cameras, C, H, W): their first dimension is not it)."""
        return len(self.img_metas)

    def get_bboxes(self, preds, img_metas=None, **kw):
        return self.head.get_bboxes(preds, self.img_metas if img_metas is None else img_metas, **kw)

    @property
    def dense_mode(self):
        return self.head.dense_mode

    @property
    def gemm_dtype(self):
        return getattr(self.head, 'gemm_dtype', torch.float32)

    def invalidate_cache(self):
        self.head.invalidate_cache()


class GraphedHead:
    """Capture ``head(pts_inputs) -> padded detections`` for one input shape.

    >>> g = GraphedHead(head, example_inputs)      # warm-up + capture
    >>> boxes, scores, labels, count = g(inputs)   # replay (outputs are static buffers, overwritten per call)
    """

    def __init__(self, head, example_inputs, warmup=3, pack=False, max_out=200):
        """pack: also capture ``dist.pack_detections`` -> ``self.packed`` (B, max_out + 1, 11), the fixed-shape record that is
        all-gathered / handed to the host, so that a serving step is the replay and nothing else."""
        assert not head.training
        self.head, self.max_out = head, max_out
        self.packed = None
        if pack:
            from .dist import DET_COLS
            frames = head.num_frames(example_inputs) if hasattr(head, 'num_frames') else example_inputs[0].shape[0]
            self.packed = torch.empty(frames, max_out + 1, DET_COLS, device=example_inputs[0].device)
        self.static_in = [example_inputs[0].clone(),
                          [t.clone() for t in example_inputs[1]] if isinstance(example_inputs[1], (list, tuple))
                          else example_inputs[1].clone()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                    # warm-up on a side stream: caches, MIOpen/hipBLASLt heuristics
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._run()
        self.preds = self._preds
        self.done = torch.cuda.Event()

    def wait(self):
        """Block the host until the last replay has finished (an event wait: nothing else on the device is waited for)."""
        self.done.synchronize()

    def _run(self):
        self._preds = self.head(self.static_in, None, None)
        dets = self.head.get_bboxes_padded(self._preds, max_out=self.max_out)
        if self.packed is not None:
            from .dist import pack_detections
            pack_detections(*dets, out=self.packed)
        return dets

    def __call__(self, inputs=None):
        if inputs is not None:
            self.static_in[0].copy_(inputs[0], non_blocking=True)
            if isinstance(self.static_in[1], list):
                for d, s_ in zip(self.static_in[1], inputs[1]):
                    d.copy_(s_, non_blocking=True)
            else:
                self.static_in[1].copy_(inputs[1], non_blocking=True)
        self.graph.replay()
        self.done.record()
        return self.static_out


class PipelinedHead:
    """Several batches in flight on one GPU: ``slots`` captured graphs, each with its own replica of the head, static input /
    output buffers and HIP stream, replayed round-robin.  Consecutive batches then overlap on the device: the ~100 short
    launches of one batch (selection, projections, attention: tens of workgroups each) run beside the other batch's convolutions
    instead of leaving most of the 256 CUs idle.  Measured (profiles/r04_a_batches_in_flight_ab.txt, one MI355X, 180 x 180 x 256):
    1 frame per batch 459 -> 706 frames/s, 2: 693 -> 944, 4: 930 -> 1135 (4 slots: 1161), 8: 1077 -> 1190, 16: 1158 -> 1232.

    >>> p = PipelinedHead(head, [inputs_a, inputs_b])      # one example input per slot: warm-up + capture
    >>> s = p.submit(batch)                                 # copy into slot s's static buffers + replay, on slot s's stream
    >>> p.wait(s); boxes, scores, labels, count = p.dets[s] # outputs of slot s stay valid until the next submit into slot s

    * Every slot owns a deep copy of the head: the derived caches of a head carry per-forward device state (the exponent hints
      the guarded NCHW -> NHWC-pair conversion reads and writes, the per-forward split memo) that two overlapping replays must
      not share.  Weights are therefore frozen at construction, as they are for any captured graph.
    * ``collective`` (a list of one process group per slot, or None): the RCCL all-gather of the packed detections
      (tools/test.py:229-233's counterpart) is captured INSIDE each graph - a step stays one replay, and the exchange of one slot
      overlaps the other slots' work.  One communicator per slot: collectives of
      one communicator must not run concurrently on two streams.  The capture uses ``capture_error_mode='thread_local'`` so
      that the process group's watchdog thread (which polls events of earlier work) does not invalidate it.
    * Waiting: ``wait`` / ``result`` wait on the slot's event; host synchronisations and eager launches between submits are fine.
    * Overlapping replays are refused (``allow_vendor_overlap=False``) when the step hands ANY dense layer to the vendor
      libraries: with hipBLASLt's bf16 GEMMs in the step two concurrent replays hang the GPU
      (profiles/r04_d_waymo_two_slots_hang.txt) - a kernel that spin-waits on workgroups of its own grid (stream-K) deadlocks
      when another graph's kernels hold the CUs those need.  None of this package's kernels waits on another workgroup.  The
      decision is made on what RAN: every vendor fallback of the path reports to ``ops.note_vendor`` during the first slot's
      warm-up (``self.vendor_calls``); round 5 moved the last two vendor GEMMs of the fp32-class step (prediction heads'
      second layer) and the whole bf16 mode (configs[4]) onto own kernels, so both run with batches in flight.
    """

    def __init__(self, head, example_inputs, slots=2, warmup=2, pack=True, max_out=200, collective=None,
                 allow_vendor_overlap=False):
        import copy
        assert not head.training and slots >= 1
        if slots > 1 and warmup < 1:
            # the decision whether replays may overlap is made on what the warm-up handed to the vendor libraries (ops.note_vendor):
            # without a warm-up step the trace is empty and says nothing
            raise ValueError('PipelinedHead: slots > 1 needs warmup >= 1 (the vendor-call trace of the warm-up decides whether '
                             'overlapping replays are safe)')
        if slots > 1 and not allow_vendor_overlap and getattr(head, 'dense_mode', 'f16x3') != 'f16x3':
            raise ValueError("PipelinedHead: more than one batch in flight needs the own kernels (dense mode 'f16x3'); vendor GEMMs in "
                             'overlapping replays can deadlock the GPU - use slots=1')
        from .dist import DET_COLS, pack_detections
        if not isinstance(example_inputs[0], (list, tuple)):          # one example: every slot starts from a copy of it
            example_inputs = [example_inputs] * slots
        assert len(example_inputs) == slots
        dev = example_inputs[0][0].device
        self.slots, self.max_out = slots, max_out
        if slots > 1 and hasattr(head, 'invalidate_cache'):
            head.invalidate_cache()                        # replicas start from the weights alone, not from cached device state
        self.heads = [head] + [copy.deepcopy(head) for _ in range(slots - 1)]
        self.static_in = [[ex[0].clone(), [t.clone() for t in ex[1]] if isinstance(ex[1], (list, tuple)) else ex[1].clone()]
                          for ex in example_inputs]
        B = head.num_frames(example_inputs[0]) if hasattr(head, 'num_frames') else example_inputs[0][0].shape[0]
        self.packed = [torch.empty(B, max_out + 1, DET_COLS, device=dev) if pack else None for _ in range(slots)]
        self.groups = collective
        self.gathered = [None] * slots
        if collective is not None:
            import torch.distributed as dist
            assert pack and len(collective) == slots
            world = dist.get_world_size(collective[0])
            self.gathered = [torch.empty(world * B, max_out + 1, DET_COLS, device=dev) for _ in range(slots)]
        # FF3D_SLOT_PRIORITY=staggered: slot 0 on a high-priority stream, the others on normal ones (A/B hook, measured level with the
        # default: equal priorities, profiles/r05_w_*)
        stag = os.environ.get('FF3D_SLOT_PRIORITY', '') == 'staggered'
        self.streams = [torch.cuda.Stream(device=dev, priority=(-1 if stag and i == 0 else 0)) for i in range(slots)]
        self.done = [torch.cuda.Event() for _ in range(slots)]
        self.graphs, self.dets = [], []

        def run(s):
            h = self.heads[s]
            dets = h.get_bboxes_padded(h(self.static_in[s], None, None), max_out=max_out)
            if pack:
                pack_detections(*dets, out=self.packed[s])
            if collective is not None:
                import torch.distributed as dist
                dist.all_gather_into_tensor(self.gathered[s], self.packed[s], group=collective[s])
            return dets
        # Overlapping replays keep the decoder's projections on this package's own kernels whatever the row count, also when
        # FF3D_LIN_MIN_ROWS asks eager steps to hand small row counts to hipBLASLt (the default is 0 since round 6: own kernels
        # everywhere): no vendor GEMM of any size runs beside another batch's kernels (see the class note on spin-waiting kernels).
        from . import transformer as _tr
        min_rows = _tr.LIN_F16X3_MIN_ROWS
        if slots > 1:
            _tr.LIN_F16X3_MIN_ROWS = 0
        from . import ops as _ops
        try:
            for s in range(slots):
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream())
                if s == 0:
                    _ops.VENDOR_CALLS = []                # trace of the dense layers this step hands to hipBLASLt / MIOpen
                try:
                    with torch.cuda.stream(side):         # warm-up on a side stream: caches, vendor heuristics, lazy RCCL init
                        for _ in range(warmup):
                            run(s)
                finally:
                    if s == 0:
                        self.vendor_calls, _ops.VENDOR_CALLS = sorted(set(_ops.VENDOR_CALLS)), None
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if s == 0 and slots > 1 and self.vendor_calls and not allow_vendor_overlap:
                    # decided on what actually RAN in the warm-up, not on the configuration: a vendor GEMM that
                    # spin-waits on its own grid (stream-K) deadlocks beside another graph's kernels
                    raise ValueError('PipelinedHead: the step hands dense layers to the vendor libraries '
                                     f'{self.vendor_calls[:6]}; vendor GEMMs in overlapping replays can deadlock the GPU '
                                     '(profiles/r04_d_waymo_two_slots_hang.txt) - use slots=1')
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local' if collective is not None else 'global'):
                    self.dets.append(run(s))
                self.graphs.append(g)
        finally:
            _tr.LIN_F16X3_MIN_ROWS = min_rows
        torch.cuda.synchronize()                           # the last device-wide wait: no replay has run yet
        self.i = -1

    def input_buffers(self, slot):
        """The static input tensors of ``slot`` in the head's own input structure ([pts_feat_conv, [stage maps]]; NeckAndHead:
        [camera maps, [LiDAR BEV map]]): what the slot's graph reads.  A producer that writes its output INTO these (the neck's
        last kernels, a data loader's device copy) hands a fresh batch over without any copy - the reference hands its features
        to the head by reference too (necks/focal_encoder.py:212-220 -> FD:522).  Use ``begin_fill`` for the stream ordering."""
        return self.static_in[slot]

    def begin_fill(self, slot=None):
        """-> (slot, input buffers) of the slot the next ``submit`` will use, after making the CURRENT stream wait (event wait,
        no host block) until that slot's previous replay has finished reading them.  Write the next batch into the buffers on
        the current stream, then ``submit(filled=True)``."""
        s = (self.i + 1) % self.slots if slot is None else slot
        torch.cuda.current_stream().wait_event(self.done[s])
        return s, self.static_in[s]

    def submit(self, inputs=None, filled=False):
        """Next slot: (copy ``inputs`` into its static buffers and) replay its graph on its stream.  Returns the slot index.
        ``filled=True``: the caller wrote the slot's ``input_buffers`` in place on the current stream (``begin_fill``) - the
        replay is ordered after those writes, nothing is copied."""
        self.i = s = (self.i + 1) % self.slots
        if filled:
            assert inputs is None, 'submit(filled=True): the batch is already in the slot\'s input buffers'
            self.streams[s].wait_stream(torch.cuda.current_stream())
        if inputs is not None:                             # produced on the caller's stream: join by an event (safe between replays)
            maps = [inputs[0]] + (list(inputs[1]) if isinstance(inputs[1], (list, tuple)) else [inputs[1]])
            mine = [self.static_in[s][0]] + (self.static_in[s][1] if isinstance(self.static_in[s][1], list) else [self.static_in[s][1]])
            if len(maps) != len(mine) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(maps, mine)):
                raise ValueError('PipelinedHead.submit: the captured graphs are for inputs of shapes '
                                 f'{[tuple(t.shape) for t in mine]}; got {[tuple(t.shape) for t in maps]}')
            self.streams[s].wait_stream(torch.cuda.current_stream())
            for t in maps:                                 # the copies below READ these on the slot's stream: the caching allocator
                t.record_stream(self.streams[s])           # must not hand their blocks to the caller's next allocation before that
        with torch.cuda.stream(self.streams[s]):
            if inputs is not None:
                self.static_in[s][0].copy_(inputs[0], non_blocking=True)
                if isinstance(self.static_in[s][1], list):
                    for d, src in zip(self.static_in[s][1], inputs[1]):
                        d.copy_(src, non_blocking=True)
                else:
                    self.static_in[s][1].copy_(inputs[1], non_blocking=True)
            self.graphs[s].replay()
            self.done[s].record()
        return s

    def eager_reference(self, slot):
        """The packed detections of slot ``slot``'s CURRENT static inputs from eager launches of the same head replica with the
        kernel routing its capture used (overlapping replays keep every projection on the own kernels) - what the slot's replay
        must reproduce bit for bit (bench.py's ``verified`` record, tests/test_small_batch_gpu.py).  Eager launches on the caller's
        stream; the pipeline may be replayed again afterwards (round 6; tools/stress_replay_sync.py does exactly that 100 times)."""
        from . import transformer as _tr
        from .dist import pack_detections
        self.wait(slot)
        h = self.heads[slot]
        min_rows = _tr.LIN_F16X3_MIN_ROWS
        if self.slots > 1:
            _tr.LIN_F16X3_MIN_ROWS = 0
        try:
            dets = h.get_bboxes_padded(h(self.static_in[slot], None, None), max_out=self.max_out)
            return pack_detections(*dets)
        finally:
            _tr.LIN_F16X3_MIN_ROWS = min_rows

    def wait(self, slot=None):
        """Block the host (EVENT wait) until slot ``slot`` (default: every slot) has finished its last replay."""
        for s in (range(self.slots) if slot is None else (slot,)):
            self.done[s].synchronize()

    def result(self, slot=None):
        """The all-gathered (or, without a collective, the packed) detections of ``slot`` (default: the last submitted one),
        after waiting for it."""
        s = self.i if slot is None else slot
        self.wait(s)
        return self.gathered[s] if self.gathered[s] is not None else self.packed[s]
