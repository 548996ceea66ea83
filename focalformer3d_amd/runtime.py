"""hipGraph execution of the head for fixed shapes.

At small batch the head is launch-bound (a few hundred kernels of a few microseconds each per frame), so the
whole ``forward`` + ``get_bboxes_padded`` sequence is captured once into a HIP graph (through
``torch.cuda.CUDAGraph``, which is hipGraph on ROCm) and replayed: every kernel of the path - vendor GEMMs /
convs and the hand-written ones, which enqueue on the capturing stream through the C ABI - becomes a graph
node; inputs are copied into static buffers, outputs are read from static buffers.  Nothing on the path
allocates outside the capture pool or synchronises with the host (see ff3d.h conventions).

Synchronisation discipline on ROCm 7.2 / torch 2.10 (tools/debug_graph3.py, debug_graph4.py, profiles/r02_f_graph_*.txt):
a hipDeviceSynchronize / hipStreamSynchronize that follows replays (with or without eager launches in between) makes the
NEXT replay die with a GPU memory fault - also with nothing but torch ops, so it is the runtime's.  What works between
replays: waiting with an EVENT (``torch.cuda.Event.synchronize``), a host read (``tensor.cpu()``), eager work on another
stream joined by events.  ``pack=True`` puts the detection packing into the graph too, so a serving step is one replay.
"""
import torch

# Guard for the discipline above: once a graph has been replayed, a host-blocking torch.cuda.synchronize() /
# Stream.synchronize() poisons further replays of it on this runtime (the next one faults the GPU).  The wrappers below count
# such calls; a GraphedHead whose last replay is older than the last counted call refuses to replay - a Python exception
# instead of a dead device.  BEST EFFORT: the flag is per GraphedHead (a sync elsewhere in the process does not condemn graphs
# that were captured, or re-captured, after it), and the wrappers only see calls that go through ``torch.cuda.synchronize`` /
# ``torch.cuda.Stream.synchronize`` looked up after the first capture - an alias bound earlier (``from torch.cuda import
# synchronize``), ``dist.barrier()``, a hipDeviceSynchronize issued by another library or ``Event.synchronize`` on an event
# recorded before a replay are not seen.  Code that synchronises by such a route calls ``GraphedHead.mark_synced()`` itself.
_STATE = {'syncs': 0, 'installed': False}


def note_host_sync():
    """Record that the host blocked on the device / a stream (what the patched torch entry points call)."""
    _STATE['syncs'] += 1


def _install_sync_guard():
    if _STATE['installed']:
        return
    _STATE['installed'] = True
    dev_sync, stream_sync = torch.cuda.synchronize, torch.cuda.Stream.synchronize

    def synchronize(device=None):
        note_host_sync()
        return dev_sync(device)

    def stream_synchronize(self):
        note_host_sync()
        return stream_sync(self)
    torch.cuda.synchronize = synchronize
    torch.cuda.Stream.synchronize = stream_synchronize


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
            self.packed = torch.empty(example_inputs[0].shape[0], max_out + 1, DET_COLS, device=example_inputs[0].device)
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
        _install_sync_guard()
        self._replayed_at = None                      # value of the sync counter at this graph's last replay (None: never replayed)

    def mark_synced(self):
        """Tell the guard that the host has synchronised with the device by a route the wrappers cannot see (see the module
        note); the next replay of this graph then raises instead of faulting."""
        note_host_sync()

    @property
    def poisoned(self):
        return self._replayed_at is not None and _STATE['syncs'] != self._replayed_at

    def wait(self):
        """Block the host until the last replay has finished - with an EVENT (the safe way to wait between replays)."""
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
        if self.poisoned:
            raise RuntimeError(
                'GraphedHead: torch.cuda.synchronize() / Stream.synchronize() was called after a graph replay; on this ROCm 7.2 / '
                'torch 2.10 runtime the next replay would fault the GPU (focalformer3d_amd/runtime.py).  Wait with '
                'GraphedHead.wait() / torch.cuda.Event.synchronize() or read an output instead, run the head eagerly, or capture '
                'a new GraphedHead (the flag is per captured graph).')
        self.graph.replay()
        self.done.record()
        self._replayed_at = _STATE['syncs']
        return self.static_out
