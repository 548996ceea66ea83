"""Parity AT THE SHAPE bench.py MEASURES (BASELINE.json configs[1]: 180x180x256 BEV, 3 x 200 queries, batch 32): the
launches that only exist at that size - the halo-tile conv at 256 -> 256 channels, the split-K GEMM at K = 37 632, the
M = 1 360 800 value_proj GEMM, batch indexing / XCD remap at B = 32 - against fp64 references and the CPU oracle."""
import pytest
import torch

from oracle import ff3d_oracle as O
from tests.test_head_gpu import _full_size_case, to_cuda
from tests.util import align_queries, oracle_cfg, permute_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from focalformer3d_amd import ops as o
    return o


def _conv64(x, w, b, stride=1):
    """fp64 3x3 conv (padding 1) on the device as nine shifted matmuls (MIOpen has no fp64 conv)."""
    B, C, H, W = x.shape
    N = w.shape[0]
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.zeros(B, N, Ho, Wo, dtype=torch.float64, device=x.device)
    for dy in range(3):
        for dx in range(3):
            patch = xp[:, :, dy:dy + (Ho - 1) * stride + 1:stride, dx:dx + (Wo - 1) * stride + 1:stride]
            out += torch.einsum('nc,bchw->bnhw', w[:, :, dy, dx].double(), patch)
    return out + b.double().view(1, -1, 1, 1)


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize('B', [4, 9])
def test_halo_conv_at_bench_launch(ops, B):
    """conv3x3 256 -> 256 on 180x180 maps with the default kernel choice (CONV_HALO='auto' picks the halo-tile kernel from
    B = 4): fp32-class error against fp64, NCHW and (hi, lo') pair outputs, every image of the batch."""
    assert ops.CONV_HALO == 'auto'
    g = torch.Generator().manual_seed(B)
    C = N = 256
    x = (torch.randn(B, C, 180, 180, generator=g) * 1.5).cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) * 0.02).cuda()
    b = torch.randn(N, generator=g).cuda()
    halo_blocks = B * 45 * 3 * 2
    assert halo_blocks >= 1024                                          # ops.conv3x3_f16x3's switch
    ref = _conv64(x, w, b)
    f32 = torch.nn.functional.conv2d(x, w, b, padding=1)
    xs, ws = ops.split_f16(x, to_nhwc=True), ops.split_weight_f16(w, bias=b)
    out = ops.conv3x3_f16x3(xs, ws, b, False, 1)
    e_ours, e_vendor = _rel(out, ref), _rel(f32, ref)
    assert e_ours < max(2 * e_vendor, 3e-7), (e_ours, e_vendor)
    for i in range(B):                                                  # per image: nothing mis-indexed across the batch
        assert _rel(out[i], ref[i]) < max(2 * e_vendor, 5e-7), i
    rec = ops.conv3x3_f16x3(xs, ws, b, True, 1, split_out=True).value().permute(0, 3, 1, 2)
    assert _rel(rec, ref.clamp_min(0)) < max(2 * e_vendor, 1e-6)
    # the implicit-GEMM kernel on the same launch gives the same fp32-class answer
    ops.CONV_HALO = '0'
    try:
        other = ops.conv3x3_f16x3(xs, ws, b, False, 1)
    finally:
        ops.CONV_HALO = 'auto'
    assert _rel(other, ref) < max(2 * e_vendor, 3e-7)


def test_pyramid_conv_at_bench_launch(ops):
    """The stride-2 pyramid convs (180 -> 90 -> 45) at C = 256, B = 8."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 256, 180, 180, generator=g).cuda()
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).cuda()
    b = torch.randn(256, generator=g).cuda()
    ref = _conv64(x, w, b, 2)
    out = ops.conv3x3_f16x3(ops.split_f16(x, to_nhwc=True), ops.split_weight_f16(w), b, False, 2)
    f32 = torch.nn.functional.conv2d(x, w, b, stride=2, padding=1)
    assert out.shape == (8, 256, 90, 90)
    assert _rel(out, ref) < max(2 * _rel(f32, ref), 3e-7)


@pytest.mark.parametrize('M,K,N', [(19200, 37632, 512), (1360800, 256, 768), (2400, 37632, 512)])
def test_gemm_f16x3_at_bench_launches(ops, M, K, N):
    """roi_mlp.0 at B = 32 (M = 32*600, K = 3*256*49: split-K on) and at B = 4; batched value_proj at B = 32
    (M = 32*42525, N = 3*256) - against fp64 in row chunks, next to hipBLASLt fp32 on the same operands."""
    g = torch.Generator(device='cuda').manual_seed(M % 1000 + N)
    a = torch.randn(M, K, device='cuda', generator=g)
    if K > 1000:
        a.relu_()                                                       # the RoI matrix is mostly positive, like this
    w = torch.randn(N, K, device='cuda', generator=g) * (1.0 / K ** 0.5)
    b = torch.randn(N, device='cuda', generator=g)
    asp, wsp = ops.split_f16(a), ops.split_weight_f16(w)
    out = ops.gemm_f16x3(asp, wsp, b, relu=True)
    again = ops.gemm_f16x3(asp, wsp, b, relu=True)
    assert torch.equal(out, again)                                      # deterministic, split-K included
    f32 = torch.relu(a @ w.t() + b)
    w64, b64 = w.double(), b.double()
    e_ours = e_vendor = scale = 0.0
    step = 65536
    for lo in range(0, M, step):
        ref = torch.relu(a[lo:lo + step].double() @ w64.t() + b64)
        e_ours = max(e_ours, float((out[lo:lo + step].double() - ref).abs().max()))
        e_vendor = max(e_vendor, float((f32[lo:lo + step].double() - ref).abs().max()))
        scale = max(scale, float(ref.abs().max()))
    assert e_ours / scale < max(2 * e_vendor / scale, 3e-7), (e_ours / scale, e_vendor / scale)


@pytest.mark.parametrize('M,K,N,relu', [(131072 + 77, 256, 200, False), (140000, 128, 384, True), (131072, 256, 130, True),
                                        (131072 + 128 * 5, 128, 128, False), (42525, 256, 768, False), (33000, 256, 1000, True)])
def test_gemm_weight_stationary_form(ops, M, K, N, relu):
    """The weight-stationary GEMM (splitmm_ws_kernel: K = 128 / 256, M >= 32 * 1024) on ragged shapes: M not a multiple of the
    128-row tile, a partial last N-tile, N not a multiple of 4 (scalar-store path), both K - against fp64 and bit-identical
    between two runs; the same operands through the tile-streaming kernel (FF3D_GEMM_WS=0 is read at first launch, so the
    comparison here is with fp64 only)."""
    g = torch.Generator(device='cuda').manual_seed(M % 1000 + N)
    a = torch.randn(M, K, device='cuda', generator=g) * 3
    w = torch.randn(N, K, device='cuda', generator=g) * (1.0 / K ** 0.5)
    b = torch.randn(N, device='cuda', generator=g)
    asp, wsp = ops.split_f16(a), ops.split_weight_f16(w, bias=b)
    out = ops.gemm_f16x3(asp, wsp, b, relu=relu)
    assert torch.equal(out, ops.gemm_f16x3(asp, wsp, b, relu=relu))
    assert out._ff3d_exp is not None and float(out.abs().max()) < 2.0 ** (int(out._ff3d_exp) + 15)
    w64, b64 = w.double(), b.double()
    e_ours = e_vendor = scale = 0.0
    for lo in list(range(0, M, 32768)):
        ref = a[lo:lo + 32768].double() @ w64.t() + b64
        f32 = a[lo:lo + 32768] @ w.t() + b
        if relu:
            ref, f32 = torch.relu(ref), torch.relu(f32)
        e_ours = max(e_ours, float((out[lo:lo + 32768].double() - ref).abs().max()))
        e_vendor = max(e_vendor, float((f32.double() - ref).abs().max()))
        scale = max(scale, float(ref.abs().max()))
    assert e_ours / scale < max(2 * e_vendor / scale, 3e-7), (e_ours / scale, e_vendor / scale)


def _check_vs_oracle(out, labels, ref, aux, taps, k=200):
    for st in taps['stages']:
        v = torch.sort(st['heat'].reshape(1, -1), descending=True).values
        assert ((v[:, k - 1] - v[:, k]) > 1e-6).all(), 'seeded frame has a top-k near-tie'
    nq = labels.numel()
    mine = {key: v[None] for key, v in out.items() if torch.is_tensor(v)}
    perm = align_queries(mine, ref, labels[None], aux['query_labels'], nq, k)            # identity unless two scores tie to round-off
    assert torch.equal(labels[None], permute_queries(aux['query_labels'], perm, nq)), 'query labels bit-exact'
    assert torch.allclose(mine['query_heatmap_score'], permute_queries(ref['query_heatmap_score'], perm, nq), atol=1e-6, rtol=0)
    for key in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap'):
        assert torch.allclose(mine[key], permute_queries(ref[key], perm, nq), atol=1e-4, rtol=1e-4), key
    for m, r in zip(out['multistage_masks'], ref['multistage_masks']):
        assert torch.equal(m, r[0])


def test_head_batch32_c256_every_frame():
    """The benchmarked configuration itself: 32 DISTINCT frames through the head at C = 256 in one batch.
      * frames 0, 17 and 31 against the CPU oracle (labels / masks bit-exact, scores 1e-6, boxes 1e-4);
      * every frame against the same frame run alone (B = 1 takes the implicit-GEMM conv and un-split GEMMs, B = 32 the
        halo-tile conv and split-K: same fp32-class arithmetic, different summation order) - labels and masks bit-exact,
        values to fp32 round-off of their magnitude (centres are O(100) cells: 1e-5 relative)."""
    B = 32
    cfg, head, sd, inputs = _full_size_case(256, B=B, seed=4)
    ocfg = oracle_cfg(cfg)
    head = head.cuda()
    dev_in = to_cuda(inputs)
    out = head(dev_in, None, [{}] * B)[0][0]
    labels = head.query_labels.clone()
    boxes, scores, blabels, count = head.get_bboxes_padded([[out]])
    assert count.tolist() == [200] * B
    host = {k: (v.cpu() if torch.is_tensor(v) else [t.cpu() for t in v]) for k, v in out.items()}
    for f in (0, 17, 31):
        taps = {}
        with torch.no_grad():
            ref, aux = O.focal_decoder_forward(sd, ocfg, [inputs[0][f:f + 1], [t[f:f + 1] for t in inputs[1]]], taps)
        mine = {k: (v[f] if torch.is_tensor(v) else [t[f] for t in v]) for k, v in host.items()}
        _check_vs_oracle(mine, labels[f].cpu(), ref, aux, taps)
    near_tie = 0
    nq = labels.shape[1]
    for f in range(B):
        one = head([dev_in[0][f:f + 1], [t[f:f + 1] for t in dev_in[1]]], None, [{}])[0][0]
        mine = {key: v[f:f + 1] for key, v in out.items() if torch.is_tensor(v)}
        try:
            # identity unless two candidates' scores agree to round-off and swap ranks between the two runs
            perm = align_queries(mine, one, labels[f:f + 1], head.query_labels, nq, 200, max_moved=4)
        except AssertionError:
            near_tie += 1           # a score within rounding of the k-th best of its stage: a different query was selected
            continue
        assert torch.equal(labels[f:f + 1].cpu(), permute_queries(head.query_labels.cpu(), perm, nq)), f
        for m_b, m_1 in zip(out['multistage_masks'], one['multistage_masks']):
            assert torch.equal(m_b[f], m_1[0]), f
        # (different conv kernels at B = 32 and B = 1: the logits agree to fp32 round-off of a 2304-term sum, not bit for bit.
        #  query_heatmap_score holds the post-NMS scores of ALL classes at the query's cell; the score of the query's own
        #  class is what get_bboxes uses and must agree; for the other classes the NMS's exact `heat == local_max` test can
        #  flip on a neighbour tie, which zeroes the entry on one side - rare, and one side is then exactly 0)
        qa, qb = out['query_heatmap_score'][f].cpu(), permute_queries(one['query_heatmap_score'].cpu(), perm, nq)[0]
        own = labels[f].cpu()[None, :]
        assert (qa.gather(0, own) - qb.gather(0, own)).abs().max().item() < 4e-6, f
        diff = (qa - qb).abs() > 2e-5
        assert ((qa == 0) | (qb == 0))[diff].all() and int(diff.sum()) <= 12, (f, int(diff.sum()), qa[diff].tolist(), qb[diff].tolist())
        for key in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap'):
            assert torch.allclose(out[key][f:f + 1].cpu(), permute_queries(one[key].cpu(), perm, nq), atol=2e-5, rtol=1e-5), (f, key)
        # the 200-box cap keeps the best 200 of the 600 decoded boxes in score order: rows with (near-)equal scores may
        # swap, or trade places across the cut - match rows by nearest box instead of by position
        b1, s1, l1, c1 = head.get_bboxes_padded([[one]])
        assert torch.allclose(scores[f], s1[0], atol=1e-6, rtol=1e-5), f
        dist = torch.cdist(boxes[f].double(), b1[0].double())
        unmatched = int((dist.min(1).values > 1e-4).sum()) + int((dist.min(0).values > 1e-4).sum())
        assert unmatched <= 2, (f, unmatched)
    assert near_tie <= 2, f'{near_tie} of {B} frames selected different queries at B=32 and B=1'


def test_head_batch4_c256_configs3_per_gpu_step():
    """The per-GPU step of BASELINE configs[3] (32 frames sharded over 8 GPUs = 4 DISTINCT frames per step) at its real size
    (180 x 180 x 256, Nq = 600): at this batch the head takes the GROUPED launches (three heatmap heads per halo-conv / tail-conv
    launch, grouped NCHW -> NHWC-pair input conversion) that neither the B = 1 nor the B = 32 full-size test exercises.
      * the grouped forms are really the ones that run (launch counters of ops);
      * frames 0 and 3 against the CPU oracle (labels / masks bit-exact, scores 1e-6, regression outputs 1e-4);
      * `get_bboxes_padded` + `pack_detections` (what the RCCL all-gather of tools/test.py:229-233's counterpart carries) against
        the oracle's `get_bboxes` for the same two frames: counts, labels, scores 1e-6, boxes 1e-4, padding rows zero."""
    from focalformer3d_amd import dist as fdist, focal_decoder as FD, ops
    B = 4
    cfg, head, sd, inputs = _full_size_case(256, B=B, seed=11)
    ocfg = oracle_cfg(cfg)
    head = head.cuda()
    dev_in = to_cuda(inputs)
    assert FD.HEATMAP_GROUPED and FD.INPUT_SPLIT_GROUPED and ops.CONV_HALO == 'auto'
    calls, originals = {}, {}
    lib = ops._lib.load()
    for name in ('ff3d_conv3x3_halo_f16x3_group', 'ff3d_conv3x3_small_f16x3_group', 'ff3d_split_f16_nhwc_group'):
        originals[name] = getattr(lib, name)

        def counted(*a, _fn=originals[name], _n=name):
            calls[_n] = calls.get(_n, 0) + 1
            return _fn(*a)
        setattr(lib, name, counted)
    try:
        out = head(dev_in, None, [{}] * B)[0][0]
    finally:
        for name, fn in originals.items():
            setattr(lib, name, fn)
    assert calls.get('ff3d_conv3x3_halo_f16x3_group', 0) >= 1 and calls.get('ff3d_conv3x3_small_f16x3_group', 0) >= 1 \
        and calls.get('ff3d_split_f16_nhwc_group', 0) >= 1, calls
    labels = head.query_labels.clone()
    boxes, scores, blabels, count = head.get_bboxes_padded([[out]])
    packed = fdist.pack_detections(boxes, scores, blabels, count)
    assert packed.shape == (B, 201, fdist.DET_COLS)
    unpacked = fdist.unpack_detections(packed)
    host = {k: (v.cpu() if torch.is_tensor(v) else [t.cpu() for t in v]) for k, v in out.items()}
    for f in (0, 3):
        taps = {}
        with torch.no_grad():
            ref, aux = O.focal_decoder_forward(sd, ocfg, [inputs[0][f:f + 1], [t[f:f + 1] for t in inputs[1]]], taps)
            res, _ = O.focal_decoder_get_bboxes(ref, aux, ocfg)
        mine = {k: (v[f] if torch.is_tensor(v) else [t[f] for t in v]) for k, v in host.items()}
        _check_vs_oracle(mine, labels[f].cpu(), ref, aux, taps)
        rb, rs, rl = res[0]
        ub, us, ul = unpacked[f]
        n = int(count[f])
        assert n == len(rb) == len(ub) and float(packed[f, 0, 0]) == n and int(packed[f, 0, 1]) == rb.shape[1]
        order = torch.sort(rs, descending=True, stable=True).indices          # ours are in score order (FD:1392-1400 keeps the best 200)
        assert torch.allclose(us, rs[order], atol=1e-6, rtol=1e-5)
        # rows with (near-)equal scores may swap: match by nearest box
        dist_ = torch.cdist(ub.double(), rb.double())
        assert int((dist_.min(1).values > 1e-4 * (1 + rb.abs().max())).sum()) == 0
        assert torch.equal(torch.sort(ul).values, torch.sort(rl.to(torch.int32)).values)
        assert float(packed[f, 1 + n:].abs().max() if n < 200 else 0.0) == 0.0
