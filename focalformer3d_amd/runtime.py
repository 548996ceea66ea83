"""hipGraph execution of the head for fixed shapes.

At small batch the head is launch-bound (a few hundred kernels of a few microseconds each per frame), so the
whole ``forward`` + ``get_bboxes_padded`` sequence is captured once into a HIP graph (through
``torch.cuda.CUDAGraph``, which is hipGraph on ROCm) and replayed: every kernel of the path - vendor GEMMs /
convs and the hand-written ones, which enqueue on the capturing stream through the C ABI - becomes a graph
node; inputs are copied into static buffers, outputs are read from static buffers.  Nothing on the path
allocates outside the capture pool or synchronises with the host (see ff3d.h conventions).
"""
import torch


class GraphedHead:
    """Capture ``head(pts_inputs) -> padded detections`` for one input shape.

    >>> g = GraphedHead(head, example_inputs)      # warm-up + capture
    >>> boxes, scores, labels, count = g(inputs)   # replay (outputs are static buffers, overwritten per call)
    """

    def __init__(self, head, example_inputs, warmup=3):
        assert not head.training
        self.head = head
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

    def _run(self):
        self._preds = self.head(self.static_in, None, None)
        return self.head.get_bboxes_padded(self._preds)

    def __call__(self, inputs=None):
        if inputs is not None:
            self.static_in[0].copy_(inputs[0], non_blocking=True)
            if isinstance(self.static_in[1], list):
                for d, s_ in zip(self.static_in[1], inputs[1]):
                    d.copy_(s_, non_blocking=True)
            else:
                self.static_in[1].copy_(inputs[1], non_blocking=True)
        self.graph.replay()
        return self.static_out
