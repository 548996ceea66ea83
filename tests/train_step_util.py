"""Shared by the CPU and GPU tests of the training-mode forward (focalformer3d_amd/train_forward.py): loads
tests/golden/train_step_*.npz (oracle/gen_golden.py:gen_train_step - one training step executed by the REFERENCE), builds our
head with the same configuration (dropout probabilities 0, as in the golden), runs forward -> loss -> backward with the recorded
ground-truth-group noise replayed, and compares predictions, losses, BatchNorm buffers and every gradient.

``oracle_kernels()`` (CPU tests only) substitutes the oracle's restatements for the HIP entry points, so that the host logic -
module wiring, autograd route, ground-truth groups, attention masks, output assembly - is checked without a GPU; the GPU test
runs the very same comparison through libff3d_hip.so."""
import contextlib
import copy
import json
import os

import numpy as np
import torch

from tests.util import GOLDEN, head_kwargs


def load_train_step(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return json.loads(bytes(z['cfg']).decode()), z


def build_train_head(cfg):
    import focalformer3d_amd.focal_decoder  # noqa: F401
    from focalformer3d_amd.registry import build_head
    h = dict(cfg['head'], input_img=False, iterbev_wo_img=True, multiscale=True, bevpos=True, mask_heatmap_mode='poscls')
    kw = head_kwargs(h)
    tl = kw['decoder_cfg']['transformerlayers']
    tl['attn_cfgs'][0]['dropout'] = 0.0
    tl['attn_cfgs'][1]['dropout'] = 0.0
    tl['ffn_dropout'] = 0.0
    kw.update(roi_dropout_rate=0.0, train_cfg=cfg['train_cfg'], gt_center_limit=h['gt_center_limit'], bn_momentum=0.1,
              **cfg['losses'])
    for key in ('add_gt_groups', 'add_gt_groups_noise', 'add_gt_groups_noise_box', 'add_gt_pos_thresh',
                'add_gt_pos_boxnoise_thresh'):
        if key in h:
            kw[key] = h[key]
    return build_head(kw)


@contextlib.contextmanager
def injected_sampling_locations(z):
    """The fixture's step was evaluated at DE-SINGULARISED sampling locations (oracle.ff3d_oracle.desingularise_sampling: the
    location gradient of bilinear sampling jumps where a pixel coordinate crosses an integer, so a sample within rounding of a
    crossing would make the recorded gradient a coin toss - the root cause of round 2's "suite-order dependent" deviation: one
    flipped sample of decoder.1.layers.1 = 3.8 % of the largest sampling-offset gradient entry, with which side it fell on
    decided by the last bit of a vendor GEMM).  Every deformable-attention call of the step gets the recorded values for the
    recorded coordinates (``msda_fix/<call>/idx|val``; each moved by at most tau = 1e-3 px, i.e. tau / size of its level in
    the normalised units of ``loc`` - they must agree with ours to 2 tau / size + fp32 round-off, per level and axis, and there
    are only a few dozen of them per call) - with the gradient of our
    own locations (straight-through) - so the reference and this implementation differentiate the same function away from its
    kinks."""
    from focalformer3d_amd import autograd as A
    inner = A.MultiScaleDeformableAttnFunction
    calls = [0]

    class Injected:
        @staticmethod
        def apply(value, shapes, start, loc, w, step=64):
            i = calls[0]
            calls[0] += 1
            idx = torch.from_numpy(z[f'msda_fix/{i}/idx']).to(loc.device)
            val = torch.from_numpy(z[f'msda_fix/{i}/val']).to(loc.device)
            fixed = loc.detach().clone().reshape(-1)
            # loc is (B, Nq, heads, L, P, 2): flat index -> level (idx // (2 P)) % L, axis idx % 2 (0 = x: W_l, 1 = y: H_l)
            L, P = loc.shape[3], loc.shape[4]
            hw = torch.tensor([[float(w_), float(h_)] for h_, w_ in shapes], device=loc.device)        # (L, 2) = (W_l, H_l)
            size = hw[(idx // (2 * P)) % L, idx % 2]
            bound = 2e-3 / size + 4e-6                       # 2 tau in pixels + a few ulps of a coordinate in [0, 1]
            assert bool(((fixed[idx] - val).abs() <= bound).all()), 'a recorded sampling location is far from ours'
            assert idx.numel() <= max(128, loc.numel() // 500), 'suspiciously many injected coordinates'
            fixed[idx] = val
            return inner.apply(value, shapes, start, loc + (fixed.view_as(loc) - loc.detach()), w, step)
    A.MultiScaleDeformableAttnFunction = Injected
    try:
        yield
    finally:
        A.MultiScaleDeformableAttnFunction = inner
    assert calls[0] == sum(1 for k in z.files if k.startswith('msda_fix/') and k.endswith('/idx')), 'MSDA call count differs'


def run_train_step(head, z, device, forward=None):
    """-> (preds dict, losses dict, {param name: grad}, [input grads])."""
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not unexpected and all('num_batches_tracked' in m for m in missing), (missing, unexpected)
    head.to(device).train()
    ins = [torch.from_numpy(z['in/pts_feat_conv']).to(device).requires_grad_(True)]
    ins += [torch.from_numpy(z[f'in/stage_{i}']).to(device).requires_grad_(True) for i in range(3)]
    B = ins[0].shape[0]
    gts = [torch.from_numpy(z[f'in/gt_boxes_{b}']).to(device) for b in range(B)]
    labels = [torch.from_numpy(z[f'in/gt_labels_{b}']).to(device) for b in range(B)]
    draws = [torch.from_numpy(z[k]) for k in sorted((k for k in z.files if k.startswith('rand/')), key=lambda s: int(s[5:]))]
    it = iter(draws)

    def replay(shape, dev):
        r = next(it)
        assert tuple(r.shape) == tuple(shape)
        return r.to(dev)
    head._rand = replay
    with injected_sampling_locations(z):
        if forward is None:
            preds = head([ins[0], list(ins[1:])], None, [{}] * B, gt_bboxes_3d=gts, gt_labels_3d=labels)
        else:
            preds = [[forward(head, [ins[0], list(ins[1:])], gts, labels)]]
    assert next(it, None) is None, 'not every recorded torch.rand draw was consumed'
    p0 = dict(preds[0][0])
    p0['dense_heatmap'] = list(p0['dense_heatmap'])
    losses = head.loss(gts, labels, preds)
    total = sum(v for n, v in losses.items() if 'loss' in n)
    total.backward()
    grads = {n: p.grad for n, p in head.named_parameters()}
    return p0, losses, grads, [t.grad for t in ins]


def _close(a, b, name, rtol=2e-4, atol_frac=2e-5):
    a, b = a.detach().float().cpu(), b.float()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    atol = atol_frac * max(1.0, float(b.abs().max()))
    assert torch.allclose(a, b, rtol=rtol, atol=atol), (name, float((a - b).abs().max()), float(b.abs().max()))


def check_train_step(z, p0, losses, grads, gin, head, grad_atol_frac=2e-4):
    """Everything the reference produced for this step vs ours.  Queries are compared in order: the golden's top-k margins are
    wide (asserted by the set comparison of the labels first)."""
    for key in z.files:
        if key.startswith('pred/'):
            parts = key.split('/')
            ours = p0[parts[1]] if len(parts) == 2 else p0[parts[1]][int(parts[2])]
            ref = torch.from_numpy(z[key])
            if ref.dtype == torch.bool or not ref.dtype.is_floating_point:
                assert torch.equal(ours.cpu().to(ref.dtype), ref), key
            else:
                _close(ours, ref, key)
    ref_losses = {k[5:]: float(z[k]) for k in z.files if k.startswith('loss/')}
    assert set(losses) == set(ref_losses)
    for name, v in losses.items():
        assert abs(float(v) - ref_losses[name]) <= 1e-4 * max(1.0, abs(ref_losses[name])), (name, float(v), ref_losses[name])
    for key in z.files:
        if key.startswith('bn_after/'):
            _close(head.state_dict()[key[9:]], torch.from_numpy(z[key]), key)
    # Gradients: every tensor within `grad_atol_frac` of its largest entry - no allowance.  (Round 2 tolerated a "suite-order
    # dependent deviation of vendor weight-gradient kernels"; round 3 root-caused it: not vendor kernels but a kink of the op -
    # see injected_sampling_locations() above; tools/debug_suite_order.py and the controlled experiment recorded in DESIGN.md §4
    # reproduce the GPU failure digit for digit on CPU by moving the sampling locations two ulps.)  A miss still writes a
    # diagnostic record (per-tensor errors, device, allocator state, the step re-run under one changed suspect at a time) to
    # gpurun_out/train_step_diag/ before the test fails.
    n_checked, missed, table = 0, [], []

    def grad_close(g, ref, name):
        m = max(float(ref.abs().max()), 1e-30)
        table.append((float((g.detach().float().cpu() - ref.float()).abs().max()) / m, name))
        try:
            _close(g, ref, name, rtol=2e-3, atol_frac=grad_atol_frac)
        except AssertionError as e:
            missed.append(str(e))
            return False
        return True
    hard = []
    for key in z.files:
        if key.startswith('grad/'):
            name = key[5:]
            ref = torch.from_numpy(z[key])
            g = grads[name]
            if g is None:
                assert not bool(z['hasgrad/' + name]) or float(ref.abs().max()) == 0.0, name
                continue
            if not grad_close(g, ref, key):
                hard.append(key)
            n_checked += 1
    assert n_checked > 50
    for i, g in enumerate(gin):
        ref = torch.from_numpy(z[f'gin/{i}'])
        assert float(ref.abs().max()) > 0
        if not grad_close(g, ref, f'gin/{i}'):
            hard.append(f'gin/{i}')
    if missed:
        _dump_train_diag(table, missed, hard)
    assert not hard, (hard, missed)


def _dump_train_diag(table, missed, hard):
    """A gradient missed the strict bound: leave a record that can be root-caused (gpurun_out/ travels back from the GPU box)."""
    import time
    root = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'gpurun_out', 'train_step_diag')
    try:
        os.makedirs(out, exist_ok=True)
        rec = {'missed': missed, 'hard': hard, 'errors_rel_to_max': sorted(table, reverse=True)[:60]}
        if torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(0)
            st = torch.cuda.memory_stats()
            rec['device'] = {'name': pr.name, 'cus': pr.multi_processor_count, 'mem': pr.total_memory,
                             'gcn': getattr(pr, 'gcnArchName', ''), 'torch': torch.__version__, 'hip': torch.version.hip}
            rec['allocator'] = {k: st[k] for k in ('reserved_bytes.all.current', 'allocated_bytes.all.current',
                                                   'reserved_bytes.all.peak', 'num_alloc_retries')}
            rec['flags'] = {'cudnn.benchmark': torch.backends.cudnn.benchmark, 'cudnn.allow_tf32': torch.backends.cudnn.allow_tf32,
                            'matmul.allow_tf32': torch.backends.cuda.matmul.allow_tf32,
                            'env': {k: v for k, v in os.environ.items() if k.startswith(('FF3D', 'MIOPEN', 'HIPBLASLT', 'ROCBLAS', 'TORCH'))}}
        tag = f'{os.getpid()}_{int(time.time())}'
        json.dump(rec, open(os.path.join(out, f'miss_{tag}.json'), 'w'), indent=1)
        print('TRAIN-STEP GRADIENT MISS:', json.dumps(rec['errors_rel_to_max'][:30]), flush=True)
        if torch.cuda.is_available() and os.environ.get('FF3D_TRAIN_DIAG_PROBES', '1') == '1':
            import contextlib as _c
            import io
            import sys as _s
            _s.path.insert(0, os.path.join(root, 'tools'))
            import debug_suite_order as D
            buf = io.StringIO()
            with _c.redirect_stdout(buf):
                D.probes(os.path.join(out, f'probes_{tag}'))
            open(os.path.join(out, f'probes_{tag}.log'), 'w').write(buf.getvalue())
            print(buf.getvalue(), flush=True)
    except Exception as e:                                   # diagnostics must never mask the assertion that follows
        print('train-step diagnostics failed:', repr(e), flush=True)


@contextlib.contextmanager
def oracle_kernels(head):
    """CPU stand-ins (the oracle's restatements) for the HIP entry points the training-mode forward and the loss reach."""
    from focalformer3d_amd import autograd as A
    from focalformer3d_amd import ops
    from oracle import ff3d_oracle as O
    from oracle import train_oracle as T
    dataset = head.test_cfg['dataset']
    small = O.SMALL_CLASSES[dataset]
    saved = {n: getattr(ops, n) for n in ('heatmap_nms', 'topk', 'query_gather', 'sine_embed', 'boxes_iou3d',
                                          'gaussian_heatmap_targets')}
    saved_fn, saved_roi, saved_decode = A.MultiScaleDeformableAttnFunction, A.RoIGridSampleFunction, head.bbox_coder.decode
    saved_decode_all = head.bbox_coder.decode_all

    def heatmap_nms(logits, mask_in=None, logits_b=None, nms_kernel=3, small_bits=0, want_mask_next=True):
        heat = logits.sigmoid() if logits_b is None else (logits.sigmoid() + logits_b.sigmoid()) / 2
        if mask_in is not None:
            heat = heat * mask_in
        heat = O.local_max_nms(heat, nms_kernel, small)
        nxt = (mask_in.clone() if mask_in is not None else torch.ones_like(heat)) if want_mask_next else None
        return heat.contiguous(), None, nxt

    def topk(heat, hist, k, workspace=None):
        return O.topk_deterministic(heat.reshape(heat.shape[0], -1), k)

    def query_gather(feat, heat, idx, cls_w, cls_b, qfeat, qpos, qscore, qlabel, mask, q_offset, mask_mode, nms_kernel, bits):
        B, K, H, W = heat.shape
        k = idx.shape[1]
        cell, cls = idx % (H * W), idx // (H * W)
        sl = slice(q_offset, q_offset + k)
        qpos[:, sl] = O.create_2d_grid(H, W).expand(B, -1, -1).gather(1, cell[:, :, None].expand(-1, -1, 2))
        qscore[:, :, sl] = heat.view(B, K, -1).gather(2, cell[:, None, :].expand(-1, K, -1))
        qlabel[:, sl] = cls
        if mask is not None and mask_mode:
            new = O.mask_update(mask.view(B, -1), idx, K, H, W, {1: 'poscls', 2: 'pos'}[mask_mode], nms_kernel, small)
            mask.copy_(new.view(B, K, H, W))

    def sine_embed(pos, dim_t, W=1.0, H=1.0):
        p = pos / torch.tensor([W, H])
        return O.gen_sineembed_for_position(p[None] if p.dim() == 2 else p).reshape(*pos.shape[:-1], 256)

    class MSDA:
        @staticmethod
        def apply(value, shapes, start, loc, w, step=64):
            return O.msda_core(value, [tuple(s) for s in shapes], loc, w)

    class RoI:
        @staticmethod
        def apply(feat_cl, query_box, level_hw, g, expand, coder, roi_range, layout=1):
            B, C = feat_cl.shape[0], feat_cl.shape[-1]
            ocfg = O.head_config(pc_range=(coder[3], coder[4]), voxel_size=(coder[1], coder[2]), out_size_factor=coder[0],
                                 dataset=dataset)
            assert tuple(O.ROI_PC_RANGE[dataset][:2]) == tuple(roi_range[:2])
            levels, off = [], 0
            for h, w in level_hw:
                levels.append(feat_cl[:, off:off + h * w].transpose(1, 2).reshape(B, C, h, w))
                off += h * w
            roi = O.roi_sample(levels, O.roi_grid_points(query_box, expand, g, ocfg))    # [level][channel][point]
            if layout == 1:
                roi = roi.view(roi.shape[0], len(level_hw), C, g * g).permute(0, 1, 3, 2).reshape(roi.shape[0], -1)
            return roi

    def decode(heatmap, rot, dim, center, height, vel, filter=False):
        ocfg = O.head_config(pc_range=tuple(head.bbox_coder.pc_range), voxel_size=tuple(head.bbox_coder.voxel_size),
                             out_size_factor=head.bbox_coder.out_size_factor)
        boxes = O.decode_box(rot, dim, center, height, vel, ocfg)
        return [dict(bboxes=boxes[i], scores=heatmap[i].max(0).values, labels=heatmap[i].max(0).indices)
                for i in range(heatmap.shape[0])]

    def decode_all(heatmap, rot, dim, center, height, vel):
        return torch.stack([d['bboxes'] for d in decode(heatmap, rot, dim, center, height, vel)])

    def gaussian_heatmap_targets(gt, labels, K, H, W, coder, overlap, min_radius):
        osf, vx, vy, px, py = coder
        hm = torch.zeros(K, H, W)
        centre = torch.cat([gt[:, :2], (gt[:, 2] + gt[:, 5] * 0.5)[:, None], gt[:, 3:]], 1)        # gravity centre, FD:1135
        for i in range(len(gt)):
            width, length = centre[i][3] / vx / osf, centre[i][4] / vy / osf
            if width > 0 and length > 0:
                radius = max(min_radius, int(T.gaussian_radius((length, width), min_overlap=overlap)))
                cx, cy = (centre[i][0] - px) / vx / osf, (centre[i][1] - py) / vy / osf
                T.draw_heatmap_gaussian(hm[labels[i]], torch.tensor([cx, cy], dtype=torch.float32).to(torch.int32), radius)
        return hm
    try:
        ops.heatmap_nms, ops.topk, ops.query_gather, ops.sine_embed = heatmap_nms, topk, query_gather, sine_embed
        ops.boxes_iou3d, ops.gaussian_heatmap_targets = T.boxes_iou3d, gaussian_heatmap_targets
        A.MultiScaleDeformableAttnFunction, A.RoIGridSampleFunction = MSDA, RoI
        head.bbox_coder.decode, head.bbox_coder.decode_all = decode, decode_all
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        A.MultiScaleDeformableAttnFunction, A.RoIGridSampleFunction = saved_fn, saved_roi
        head.bbox_coder.decode, head.bbox_coder.decode_all = saved_decode, saved_decode_all


__all__ = ['load_train_step', 'build_train_head', 'run_train_step', 'check_train_step', 'oracle_kernels', 'copy']
