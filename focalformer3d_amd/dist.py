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


def pack_detections(boxes, scores, labels, count):
    """(B,M,7|9), (B,M), (B,M) int32, (B,) int32 -> (B, M+1, 11) fp32; row 0 carries the count
    (exact in fp32 for M < 2^24), rows 1.. the zero-padded detections."""
    B, M, D = boxes.shape
    out = boxes.new_zeros(B, M + 1, DET_COLS)
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
