"""Multi-GPU: frames are independent units - shard them over ranks, one process per
GPU, no data-path collective; the only exchange is one fixed-shape all-gather of the padded detections
(RCCL over xGMI when the backend is "nccl"), replacing mmdet ``multi_gpu_test``'s pickled-bytes gather
(tools/test.py:233).  ~8.8 KB per frame: latency-bound, one collective per batch.
"""
import torch
import torch.distributed as dist

DET_COLS = 11   # 9 box values (7 when there is no velocity, zero padded) + score + label


def shard_range(num_frames, rank, world_size):
    """Contiguous chunk [lo, hi) of ``num_frames`` owned by ``rank`` (remainder spread over the first ranks)."""
    base, rem = divmod(num_frames, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _parse_cpulist(text):
    """'0-3,8,10-11' (the kernel's cpulist format) -> sorted list of ints."""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def rank_cpu_slice(allowed, local_rank, local_world, node_cpus=None, ranks_on_node=None):
    """Host cores of one rank (pure function, tested on CPU).  ``allowed``: the cores this process may use.  With the NUMA
    information of the rank's GPU (``node_cpus``: cores of the GPU's node, ``ranks_on_node``: (position of this rank among the
    ranks whose GPU sits on that node, their number)) the rank takes its share of THAT node's allowed cores - host launches and
    the runtime's threads then stay next to the GPU's PCIe root; without it, a contiguous 1 / local_world share of ``allowed``.
    Never returns an empty set: shares smaller than one core fall back to round-robin single cores."""
    allowed = sorted(allowed)
    pool, pos, n = allowed, local_rank, local_world
    if node_cpus:
        mine = [c for c in allowed if c in set(node_cpus)]
        if mine and ranks_on_node:
            pool, (pos, n) = mine, ranks_on_node
    if not pool:
        return []
    if len(pool) < n:
        return [pool[pos % len(pool)]]
    per = len(pool) // n
    return pool[pos * per:(pos + 1) * per]


def _gpu_numa_node(pci_bus_id):
    try:
        with open(f'/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node') as f:
            node = int(f.read().strip())
        if node < 0:
            return None, None
        with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
            return node, _parse_cpulist(f.read())
    except (OSError, ValueError, AttributeError):
        return None, None


def gpu_pci_address(device=None, props=None):
    """sysfs address 'dddd:bb:dd.0' of a GPU from torch's device properties (``pci_domain_id`` / ``pci_bus_id`` / ``pci_device_id``
    are INTEGERS in torch - round 5 tested ``isinstance(bus, str)`` on them, which never held, so the NUMA branch below was dead
    code: ADVICE r05), or None when the properties do not carry them."""
    try:
        pr = props if props is not None else torch.cuda.get_device_properties(device)
        return '%04x:%02x:%02x.0' % (int(getattr(pr, 'pci_domain_id', 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
    except (AttributeError, TypeError, ValueError, RuntimeError, AssertionError):
        return None


def pin_host_threads(local_rank, local_world, device=None, all_pci_bus_ids=None):
    """One process per GPU: keep a rank's host threads (its launch loop, RCCL's proxy threads, the HIP runtime's workers) on its
    own cores instead of letting N ranks migrate over each other - ``os.sched_setaffinity`` by local rank, next to the GPU's NUMA
    node when sysfs knows it (tools/dist_test.sh:9-11 leaves this to the launcher).  ``all_pci_bus_ids``: the sysfs PCI address
    (``gpu_pci_address``) of every local rank's GPU in local-rank order (lets ranks that share a node split it).  The NUMA-aware
    split is used only when EVERY local rank's node is known - every rank evaluates the same list against the same sysfs, so all of
    them take the same branch and the shares never overlap; otherwise all ranks take plain contiguous shares.  Returns a record for the bench line:
    {'cpus': n, 'first': c0, 'last': c1, 'numa_node': k | None}; {} where the platform has no affinity call."""
    import os
    if not hasattr(os, 'sched_setaffinity') or local_world <= 1:
        return {}
    allowed = sorted(os.sched_getaffinity(0))
    node = node_cpus = on_node = None
    if device is not None and torch.cuda.is_available():
        bus = gpu_pci_address(device)
        if isinstance(bus, str) and all_pci_bus_ids and len(all_pci_bus_ids) == local_world:
            nodes = [_gpu_numa_node(b)[0] if isinstance(b, str) else None for b in all_pci_bus_ids]
            if all(n is not None for n in nodes):           # all known, or nobody uses the node information
                node, node_cpus = _gpu_numa_node(bus)
                peers = [r for r, n in enumerate(nodes) if n == node]
                if node is not None and local_rank in peers:
                    on_node = (peers.index(local_rank), len(peers))
    if node_cpus and on_node is None:
        node_cpus = None                                    # peers unknown: plain contiguous shares (never overlapping)
    cpus = rank_cpu_slice(allowed, local_rank, local_world, node_cpus, on_node)
    if not cpus:
        return {}
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return {}
    torch.set_num_threads(max(1, min(len(cpus), 8)))        # the host side of a rank is a launch loop, not a compute pool
    return {'cpus': len(cpus), 'first': cpus[0], 'last': cpus[-1], 'numa_node': node if node_cpus else None}


def pack_detections(boxes, scores, labels, count, out=None):
    """(B,M,7|9), (B,M), (B,M) int32, (B,) int32 -> (B, M+1, 11) fp32; row 0 carries the count
    (exact in fp32 for M < 2^24), rows 1.. the zero-padded detections.  Device tensors: one HIP launch
    (ff3d_pack_detections); host tensors (the gloo rehearsal / CPU tests of the collective plumbing): index ops."""
    B, M, D = boxes.shape
    if out is None:
        out = boxes.new_empty(B, M + 1, DET_COLS)
    if boxes.is_cuda:
        from . import ops
        return ops.pack_detections(boxes, scores, labels, count, out)
    out.zero_()
    out[:, 0, 0] = count.to(out.dtype)
    out[:, 0, 1] = float(D)
    out[:, 1:, :D] = boxes
    out[:, 1:, 9] = scores
    out[:, 1:, 10] = labels.to(out.dtype)
    return out


def unpack_detections(packed):
    """Inverse of pack_detections -> list of (boxes (n,D), scores (n,), labels int32 (n,)) per frame (host sync)."""
    packed = packed.cpu()
    res = []
    for f in packed:
        n, D = int(f[0, 0]), int(f[0, 1])
        res.append((f[1:1 + n, :D], f[1:1 + n, 9], f[1:1 + n, 10].to(torch.int32)))
    return res


def gather_detections(boxes, scores, labels, count, group=None):
    """All-gather the padded detections of every rank: returns (world*B, M+1, 11) on every rank, frames in
    rank order.  Every rank must pass the same B and M (fixed shapes: no pickling, graph friendly)."""
    packed = pack_detections(boxes, scores, labels, count)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return packed
    world = dist.get_world_size(group)
    out = packed.new_empty((world * packed.shape[0],) + tuple(packed.shape[1:]))
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    return out


class AsyncDetectionGather:
    """The per-batch exchange of the sharded decoder, off the critical path: the packed detections of batch i are
    all-gathered on a side stream (RCCL over xGMI) while the main stream already decodes batch i+1.

    >>> g = AsyncDetectionGather(B, M, device)
    >>> for batch in batches:
    ...     dets = head.get_bboxes_padded(head(batch, None, metas))
    ...     g.submit(*dets)               # pack on the main stream (1 launch), all-gather on the side stream
    >>> packed = g.result()               # (world*B, M+1, 11) of the LAST submitted batch, main stream synchronised with it

    Two slots of pack / gather buffers alternate so that a gather in flight never races the next pack.  Without an
    initialised process group (or world size 1) submit() only packs."""

    def __init__(self, B, M, device, group=None, slots=2, force_collective=False):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # force_collective: issue the all-gather even in a 1-rank group (rehearsal of the RCCL + side-stream path on one GPU)
        self.collective = self.world > 1 or (force_collective and dist.is_available() and dist.is_initialized())
        self.cuda = torch.device(device).type == 'cuda'
        self.packed = [torch.empty(B, M + 1, DET_COLS, device=device) for _ in range(slots)]
        self.out = [torch.empty(self.world * B, M + 1, DET_COLS, device=device) if self.collective else None
                    for _ in range(slots)]
        self.side = torch.cuda.Stream(device=device) if (self.cuda and self.collective) else None
        self.done = [None] * slots           # event: gather of this slot finished (side stream)
        self.i = -1

    def submit(self, boxes, scores, labels, count):
        self.i = (self.i + 1) % len(self.packed)
        s = self.i
        if self.side is not None and self.done[s] is not None:
            torch.cuda.current_stream().wait_event(self.done[s])       # slot reuse: its previous gather must be over
        pack_detections(boxes, scores, labels, count, self.packed[s])
        if not self.collective:
            return
        if self.side is None:                                          # host tensors (gloo): synchronous
            dist.all_gather_into_tensor(self.out[s], self.packed[s], group=self.group)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            dist.all_gather_into_tensor(self.out[s], self.packed[s], group=self.group)
            ev = torch.cuda.Event()
            ev.record()
        self.done[s] = ev

    def adopt(self, packed):
        """Graph replay: the packing ran inside the captured graph (runtime.GraphedHead(pack=True)) into its static buffer.
        Without a collective there is nothing to launch.  With one, the exchange stays OUTSIDE the graph: on the side stream,
        joined to the replay by an event, the record is copied out of the static buffer and all-gathered (eager launches on a
        second stream joined by events are the pattern that is safe between replays on this ROCm, runtime.py); the main stream
        waits only for that 9 KB-per-frame copy before the next replay may overwrite the buffer."""
        self.i = (self.i + 1) % len(self.packed)
        s = self.i
        if not self.collective:
            self.packed[s] = packed
            return
        if self.side is None:                                          # host tensors (gloo rehearsal): synchronous
            self.packed[s].copy_(packed)
            dist.all_gather_into_tensor(self.out[s], self.packed[s], group=self.group)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):                             # (in-order stream: slot s's previous gather is over)
            self.side.wait_event(ready)
            self.packed[s].copy_(packed, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record()
            dist.all_gather_into_tensor(self.out[s], self.packed[s], group=self.group)
            ev = torch.cuda.Event()
            ev.record()
        self.done[s] = ev
        torch.cuda.current_stream().wait_event(copied)

    def result(self):
        s = self.i
        if not self.collective:
            return self.packed[s]
        if self.side is not None and self.done[s] is not None:
            torch.cuda.current_stream().wait_event(self.done[s])
        return self.out[s]
