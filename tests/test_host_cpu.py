"""CPU tests of the host-side mirror: registry-driven construction from reference-style config dicts,
state-dict layout against the weights the REFERENCE module produced (golden
fixtures), API surface, and that the product refuses to compute without the GPU path."""
import os

import numpy as np
import pytest
import torch

import focalformer3d_amd.focal_decoder  # noqa: F401  (registration side effect)
from focalformer3d_amd import registry
from tests.util import head_kwargs, load_golden

HEADS = ['head_focal_L', 'head_focal_LC', 'head_deform_L', 'head_waymo',
         'head_opt_classaware', 'head_opt_posmask', 'head_opt_singlescale', 'head_opt_singleheat',
         'head_opt_heatbox', 'head_opt_boxcls']


@pytest.mark.parametrize('name', HEADS)
def test_state_dict_layout_matches_reference(name):
    cfg, sd, _, _, _ = load_golden(name)
    head = registry.build_head(head_kwargs(cfg))
    ours = {k: tuple(v.shape) for k, v in head.state_dict().items() if 'num_batches_tracked' not in k}
    ref = {k: tuple(v.shape) for k, v in sd.items()}
    assert ours == ref                      # same keys, same shapes as the reference module's state dict
    head.load_state_dict(sd, strict=False)


def test_registry_type_strings():
    for reg, names in ((registry.HEADS, ['FocalDecoder']), (registry.BBOX_CODERS, ['TransFusionBBoxCoder']),
                       (registry.TRANSFORMER_LAYER_SEQUENCE, ['DeformableDetrTransformerDecoder']),
                       (registry.TRANSFORMER_LAYER, ['DetrTransformerDecoderLayer']),
                       (registry.ATTENTION, ['MultiheadAttention', 'MultiScaleDeformableAttention']),
                       (registry.FEEDFORWARD_NETWORK, ['FFN'])):
        for n in names:
            assert reg.get(n) is not None, n


def test_head_api_surface_and_reference_quirks():
    cfg, sd, inp, _, _ = load_golden('head_focal_L')
    head = registry.build_head(head_kwargs(cfg), test_cfg=None)
    assert head.multistage_heatmap == cfg['multistage_heatmap'] + 1          # FD:138-139 (+1 with reuse_first_heatmap)
    assert head.heatmap_head_img[0] is None                                   # FD:226-227
    assert head.heatmap_head[1].bias is not None                              # bias='auto' is truthy (FD:213-220)
    assert head.heatmap_head[0].conv.bias is None
    assert head.bev_pos.shape == (1, cfg['grid'] ** 2, 2)
    assert torch.equal(head.bev_pos[0, 1], torch.tensor([1.5, 0.5]))         # (x+0.5, y+0.5), x fastest
    for attr in ('query_labels', 'num_proposals', 'num_proposals_ori', 'bbox_coder', 'test_cfg', 'num_classes'):
        assert hasattr(head, attr)
    with pytest.raises(NotImplementedError):
        head.get_heatmap_targets()                                            # heatmap_box branch: training side not mirrored
    with pytest.raises(RuntimeError):
        head.loss(None, None, None)                                           # targets / losses need train_cfg
    with pytest.raises(RuntimeError):                                         # training-mode forward: no CPU fallback either
        head.train()([inp['pts_feat_conv']], None, [{}])
    with pytest.raises(RuntimeError):                                         # no CPU fallback
        head.eval()([inp['pts_feat_conv'], [inp['stage_0'], inp['stage_1'], inp['stage_2']]], None, [{}])


def test_unsupported_reference_options_raise_at_build():
    cfg, *_ = load_golden('head_focal_L')
    # heatmap_box alone = DCNSeparateHead task heads (mmdet3d deformable convs); boxcls without the heatmap boxes has no boxes to test
    for bad in (dict(heatmap_box=True), dict(boxpos='xywlr'), dict(mask_heatmap_mode='boxcls'),
                dict(initialize_by_heatmap=False)):
        kw = head_kwargs(cfg)
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            registry.build_head(kw)
    kw = head_kwargs(cfg)
    kw.update(heatmap_box=True, thin_heatmap_box=True, multistage_heatmap=None)      # the task heads are per stage (FD:221-231)
    with pytest.raises(ValueError):
        registry.build_head(kw)


def test_thin_heatmap_box_head_builds_with_the_reference_parameter_layout():
    """heatmap_box + thin_heatmap_box (FD:231-287): one (ConvModule, Conv2d -> 6 x 10) task head per stage; the state dict of the
    reference-built head (the fixture's) loads strictly."""
    cfg, sd, inp, _, _ = load_golden('head_opt_boxcls')
    head = registry.build_head(head_kwargs(cfg))
    assert head.heatmap_box and head.thin_heatmap_box and head.mask_heatmap_mode == 'boxcls'
    assert len(head.multi_stage_task_heads) == head.multistage_heatmap == 3
    assert head.multi_stage_task_heads[0][1].weight.shape == (60, cfg['hidden_channel'], 3, 3)
    assert [t['num_class'] for t in head.heatmap_tasks] == [1, 2, 2, 1, 2, 2]
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    with pytest.raises(RuntimeError):                                         # training side of the branch: not built, and no CPU route
        head.train()([inp['pts_feat_conv']], None, [{}])


def test_points_in_boxes_restatement_hand_case():
    """oracle.points_in_boxes (mmdet3d v0.17.1 points_in_boxes_gpu, un-vendored): offset rotated by rz + pi / 2, local_x against l,
    local_y against w, strict inequalities, first containing box wins, bottom-centre z."""
    from oracle import ff3d_oracle as O
    # box 0: centre (0, 0), w = 2, l = 6, rz = 0 -> rotation by pi / 2: local_x = -y... a point is inside iff |y| < 3 and |x| < 1
    # box 1: same footprint turned by rz = pi / 2 -> inside iff |x| < 3 and |y| < 1;  box 2: far away
    boxes = torch.tensor([[[0., 0., -1., 2., 6., 2., 0.], [0., 0., -1., 2., 6., 2., np.pi / 2], [50., 50., -1., 1., 1., 2., 0.3]]])
    pts = torch.tensor([[[0.5, 2.5, 0.], [2.5, 0.5, 0.], [0.5, 0.5, 0.], [2.5, 2.5, 0.], [0.5, 2.5, 1.5], [1.0, 0., 0.], [50.2, 50.1, 0.]]])
    got = O.points_in_boxes(pts, boxes)[0].tolist()
    #        box 0 only | box 1 only | both: first | none | above the box (z in [-1, 1]) | on box 0's edge: strict -> box 1 | box 2
    assert got == [0, 1, 0, -1, -1, 1, 2]


def test_bn_folding_matches_module():
    from focalformer3d_amd.layers import ConvModule
    torch.manual_seed(0)
    m = ConvModule(8, 6, 3, padding=1, conv_cfg=dict(type='Conv2d'), norm_cfg=dict(type='BN2d')).eval()
    with torch.no_grad():
        m.bn.running_mean.normal_()
        m.bn.running_var.uniform_(0.5, 1.5)
        m.bn.weight.normal_()
        m.bn.bias.normal_()
        x = torch.randn(2, 8, 7, 7)
        w, b = m.folded()
        assert torch.allclose(torch.relu(torch.nn.functional.conv2d(x, w, b, padding=1)), m(x), atol=1e-5)


def test_prediction_head_fusion_matches_per_head():
    from focalformer3d_amd.layers import FFN
    torch.manual_seed(1)
    heads = dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2), heatmap=(10, 2))
    m = FFN(16, heads).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.normal_()
        for n, b in m.named_buffers():
            if 'running_var' in n:
                b.uniform_(0.5, 1.5)
        x = torch.randn(3, 16, 11)
        w1, b1, w2, b2, sizes = m.fused_weights()
        hid = torch.relu(torch.nn.functional.linear(x.transpose(1, 2), w1, b1))
        out = torch.matmul(w2, hid.transpose(1, 2)) + b2[:, None]
        for (k, ref), got in zip(m(x).items(), out.split(sizes, 1)):
            assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4), k


def test_lift_splat_shoot_state_dict_layout_matches_reference():
    """LiftSplatShoot mirror: same parameter / buffer names and shapes as the reference module (golden's state dict)."""
    from focalformer3d_amd.lss import LiftSplatShoot
    cfg, sd, _, _, _ = load_golden('lss_small')
    m = LiftSplatShoot(img_scale=tuple(cfg['img_scale']), camera_depth_range=cfg['depth_range'], pc_range=cfg['pc_range'],
                       downsample=cfg['downsample'], grid=cfg['grid'], inputC=cfg['inputC'], outputC=cfg['outputC'],
                       camC=cfg['camC'])
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items() if 'num_batches_tracked' not in k}
    assert ours == {k: tuple(v.shape) for k, v in sd.items()}
    assert torch.equal(m.frustum.data, sd['frustum'])                  # create_frustum (lss.py:217-230) reproduced exactly
    with pytest.raises(RuntimeError, match='HIP'):
        m.eval()(torch.zeros(1, 1, cfg['inputC'], 8, 16), torch.eye(3).view(1, 1, 3, 3), torch.zeros(1, 1, 3))


def test_tta_mapping_back_and_bev_helpers_match_oracle():
    """merge_augs host algebra (bbox3d_mapping_back, bev / xywhr2xyxyr) against the oracle restatement."""
    from focalformer3d_amd import merge_augs as MA
    from oracle import ff3d_oracle as O
    g = torch.Generator().manual_seed(3)
    b = torch.randn(50, 9, generator=g)
    for scale, fh, fv in [(1.0, False, False), (0.95, True, False), (1.05, False, True), (1.1, True, True)]:
        assert torch.allclose(MA.bbox3d_mapping_back(b, scale, fh, fv), O.bbox3d_mapping_back(b, scale, fh, fv), atol=1e-6)
    assert torch.allclose(MA.xywhr2xyxyr(MA.bev_of(b)), O.xywhr2xyxyr(b[:, [0, 1, 3, 4, 6]]))
    with pytest.raises(RuntimeError, match='HIP'):
        MA.merge_boxes(b, torch.rand(50), torch.zeros(50, dtype=torch.long))


def test_split_weight_pairs_and_zero_row_contract():
    """Range-normalised (hi, lo') weight planes: 2^exp * (hi + lo'/2048) reproduces the fp32 weight to ~2^-22 whatever its
    magnitude, the scaled maximum sits in [2^13, 2^14), conv weights are tap-major, each plane is followed by its zero row
    (and class padding rows are zero); the output-bound scalars are {largest row sum of |W|, max|bias|}."""
    from focalformer3d_amd import ops
    g = torch.Generator().manual_seed(5)
    for scale in (0.03, 3e4, 1e-7):
        w = torch.randn(10, 32, 3, 3, generator=g) * scale
        b = torch.randn(10, generator=g)
        pair = ops.split_weight_f16(w, pad_rows_to=16, bias=b)
        hi, lo = pair
        assert hi.shape == (16, 3, 3, 32) and hi.dtype == torch.float16 and pair.exp.dtype == torch.int32
        rec = pair.value()[:10].permute(0, 3, 1, 2)
        assert ((rec - w).abs() <= w.abs() * 2.0 ** -21 + w.abs().max() * 2.0 ** -30).all()
        assert 2.0 ** 13 <= float(hi.float().abs().max()) <= 2.0 ** 14
        assert torch.allclose(pair.bound, torch.stack((w.abs().flatten(1).sum(1).max(), b.abs().max())))
        assert not hi[10:].any() and not lo[10:].any()
        row = hi[0].numel()
        for plane in (hi, lo):                                             # the storage continues with one zero row
            tail = torch.as_strided(plane, (row,), (1,), plane.storage_offset() + plane.numel())
            assert not tail.any()
    lin = torch.randn(7, 64, generator=g)
    p2 = ops.split_weight_f16(lin)
    assert p2[0].shape == (7, 64) and torch.allclose(p2.value(), lin, rtol=2.0 ** -20, atol=1e-9)
    assert float(p2.bound[1]) == 0.0
    v = p2.view(7, 8, 8)                                                   # views keep the exponent
    assert v.exp is p2.exp and v[0].shape == (7, 8, 8)


def test_dense_mode_switch():
    from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg
    head = build_head_from_cfg(focalformer3d_l_head_cfg(C=32, grid=16, num_proposals=8, stages=2, decoder_stages=1, ffn=32,
                                                        hidden_channel_roi=32))
    assert head.dense_mode in ('f16x3', 'vendor')
    head.set_dense_mode('vendor')
    assert head.dense_mode == 'vendor' and head._cache is None
    with pytest.raises(AssertionError):
        head.set_dense_mode('bf16')


def test_weight_caches_follow_in_place_weight_updates():
    """Derived caches (folded BN, fused heads, concatenated projections) are keyed on (data_ptr, _version) of the weights:
    a checkpoint load that bypasses torch's load_state_dict hooks - mmcv ``load_checkpoint`` recurses through
    ``_load_from_state_dict`` - or any other in-place update must rebuild them."""
    cfg, sd, _, _, _ = load_golden('head_focal_L')
    head = registry.build_head(head_kwargs(cfg)).eval()
    d0 = head._derived()
    assert head._derived() is d0                                  # unchanged weights: cache hit
    w_before = d0['hm'][0].clone()
    msda = head.decoder[0].layers[0].attentions[1]
    f0 = msda._fused_offlog()[0].clone()
    # mmcv-style load: per-module _load_from_state_dict, no post hooks
    for prefix, m in head.named_modules():
        m._load_from_state_dict(sd, prefix + '.' if prefix else '', {}, False, [], [], [])
    d1 = head._derived()
    assert d1 is not d0
    assert not torch.equal(d1['hm'][0], w_before)
    assert torch.equal(d1['hm'][0], head.heatmap_head[0].folded()[0])
    assert not torch.equal(msda._fused_offlog()[0], f0)
    assert torch.equal(msda._fused_offlog()[0][:msda.sampling_offsets.weight.shape[0]], msda.sampling_offsets.weight)
    # optimizer / EMA style in-place update
    with torch.no_grad():
        head.heatmap_head[0].conv.weight.mul_(2.0)
    assert torch.equal(head._derived()['hm'][0], head.heatmap_head[0].folded()[0])
    # two heads in one process keep their own dense mode
    other = registry.build_head(head_kwargs(cfg)).eval()
    other.set_dense_mode('vendor')
    head.set_dense_mode('f16x3')
    assert other.decoder[0].layers[0].attentions[0].attn_f16x3 is False
    assert head.decoder[0].layers[0].attentions[0].attn_f16x3 is True


def test_reference_bev_pool_kernel_builds_from_its_own_source():
    """oracle/build_ref.py compiles the reference's bev_pool_cuda.cu where it lies (no copy, no stand-in headers) and the
    library exports the reference's own launcher; on a machine without /root/reference the recipe is a no-op."""
    import ctypes
    import os
    from oracle import build_ref
    libs = build_ref.build(verbose=False)
    if not os.path.exists(build_ref.BEV_POOL_SRC):
        assert libs == []
        return
    assert libs == [build_ref.BEV_POOL_LIB] and os.path.exists(build_ref.BEV_POOL_LIB)
    assert hasattr(ctypes.CDLL(build_ref.BEV_POOL_LIB), build_ref.BEV_POOL_SYMBOL)
    tracked = os.popen('git -C %s ls-files oracle/_ref' % os.path.dirname(build_ref.HERE)).read().strip()
    assert tracked == ''


def test_apply_3d_transformation_product_vs_oracle_and_round_trip():
    """mmdet3d's apply_3d_transformation (un-vendored, A.5): the product composes the recorded flow into one affine map
    (coord_transform.py), the oracle applies it step by step; forward then reverse is the identity; a hand-computed case."""
    import math
    import numpy as np
    import torch
    from focalformer3d_amd import coord_transform as T
    from oracle import ff3d_oracle as O
    ang = 0.3
    rot_t = torch.tensor([[math.cos(ang), -math.sin(ang), 0.0], [math.sin(ang), math.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    meta = dict(pcd_rotation=rot_t, pcd_scale_factor=1.05, pcd_trans=np.array([0.3, -0.2, 0.1]), pcd_horizontal_flip=True,
                pcd_vertical_flip=False, transformation_3d_flow=['HF', 'VF', 'R', 'S', 'T'])
    p = torch.randn(50, 3, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    for rev in (False, True):
        a, b = T.apply_3d_transformation(p, 'LIDAR', meta, rev), O.apply_3d_transformation(p, meta, rev)
        assert torch.allclose(a, b, atol=1e-12)
    back = T.apply_3d_transformation(T.apply_3d_transformation(p, 'LIDAR', meta, False), 'LIDAR', meta, True)
    assert torch.allclose(back, p, atol=1e-9)
    one = T.apply_3d_transformation(torch.tensor([[1.0, 2.0, 3.0]], dtype=torch.float64), 'LIDAR', meta, False)[0]
    x, y, z = 1.0, -2.0, 3.0                                          # HF; VF is recorded but did not happen
    xr, yr = x * math.cos(ang) + y * math.sin(ang), -x * math.sin(ang) + y * math.cos(ang)       # points @ rot_mat_T
    assert torch.allclose(one, torch.tensor([xr * 1.05 + 0.3, yr * 1.05 - 0.2, z * 1.05 + 0.1], dtype=torch.float64), atol=1e-6)
    l2i = np.eye(4, dtype=np.float32)[None]
    folded = T.fold_into_lidar2img(l2i, meta)[0]
    q = torch.cat([p, torch.ones(50, 1, dtype=torch.float64)], 1) @ torch.from_numpy(folded).double().t()
    assert torch.allclose(q[:, :3], O.apply_3d_transformation(p, meta, True), atol=1e-6)


def test_gemm_ksplit_fills_the_last_round():
    """ops.gemm_ksplit: long-K GEMMs on at most one round of the 512 resident blocks are cut to fill it; grids of a few rounds
    get the slice count whose LAST round is fullest (roi_mlp.0: 600 tiles at 32 frames -> 5 slices = 5.86 rounds); short K and
    large grids are left alone; the slices stay at least 64 K-steps long."""
    from focalformer3d_amd import ops
    K = 37632
    assert ops.gemm_ksplit(600, 512, K) == 25                      # 1 frame: 20 tiles -> 500 blocks
    assert ops.gemm_ksplit(2400, 512, K) == 6                      # 4 frames: 76 tiles -> 456 blocks
    for rows in (9600, 19200, 38400):                              # 16 / 32 / 64 frames
        ks = ops.gemm_ksplit(rows, 512, K)
        tiles = -(-rows // 128) * 4
        rounds = tiles * ks / 512.0
        assert 1 <= ks <= 6 and rounds / -(-tiles * ks // 512) > 0.93, (rows, ks)
    assert ops.gemm_ksplit(19200, 512, 1024) == 1                  # short K
    assert ops.gemm_ksplit(1360800, 768, 4096) == 1                # 63 792 tiles: rounds do not matter
    assert ops.gemm_ksplit(19200, 512, 4096) in (1, 2)             # 128 K-steps: at most two slices of >= 64 steps


def test_no_memset_or_memcpy_nodes_can_enter_a_captured_graph():
    """Round 6 root cause of the [replay, eager launch, synchronise, replay] GPU fault: hipMemsetAsync captured as a memset node
    (profiles/r06_a_graph_fault_bisect.txt).  The library zero-fills with a kernel; the only hipMemsetAsync left in the sources is the
    A/B hook behind FF3D_MEMSET_NODES=1 inside zero_u32, and nothing calls hipMemcpy*.  The guard class of rounds 3-5 is gone and torch's
    entry points are untouched."""
    import glob
    import re
    import torch
    from focalformer3d_amd import runtime as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = []
    for path in sorted(glob.glob(os.path.join(root, 'focalformer3d_amd', 'csrc', '*'))):
        text = re.sub(r'//[^\n]*', '', open(path).read())                 # (comments may name the calls)
        hits += [(os.path.basename(path), m.group(0)) for m in re.finditer(r'hipMem(set|cpy)\w*\s*\(', text)]
    assert hits == [('heatmap.hip', 'hipMemsetAsync(')], hits
    src = open(os.path.join(root, 'focalformer3d_amd', 'csrc', 'heatmap.hip')).read()
    body = src[src.index('int zero_u32('):src.index('}  // namespace', src.index('int zero_u32('))]
    assert 'FF3D_MEMSET_NODES' in body and 'hipMemsetAsync' in body and 'zero_u32_kernel' in body
    assert not hasattr(R, '_ReplayGuard') and not hasattr(R.GraphedHead, 'host_synced') and not hasattr(R.PipelinedHead, 'host_synced')
    assert torch.cuda.synchronize.__module__.startswith('torch')


def test_heuristic_assigner_scatter_form_equals_the_reference_loop():
    """HeuristicAssigner3D's device form (arg-min over (distance rank, box index) per proposal by scatter-reduce) against the
    oracle's restatement of the reference loop (hungarian_assigner.py:58-91) on random cases, IoU calculator stubbed with the
    oracle's (the HIP IoU kernel has its own GPU test)."""
    from oracle import train_oracle as T
    from focalformer3d_amd import training as TR
    g = torch.Generator().manual_seed(3)

    def boxes(n):
        b = torch.zeros(n, 9)
        b[:, :2] = torch.rand(n, 2, generator=g) * 60 - 30
        b[:, 3:6] = torch.rand(n, 3, generator=g) + 0.5
        b[:, 6] = torch.rand(n, generator=g) * 6 - 3
        return b
    for trial in range(12):
        P, G = int(torch.randint(5, 80, (1,), generator=g)), int(torch.randint(1, 60, (1,), generator=g))
        pred, gt = boxes(P), boxes(G)
        gl, ql = torch.randint(0, 4, (G,), generator=g), torch.randint(0, 4, (P,), generator=g)
        a = TR.HeuristicAssigner3D.__new__(TR.HeuristicAssigner3D)
        a.dist_thre, a.iou_calculator = 5.0 + trial, (lambda x, y: T.boxes_iou3d(x, y))
        q = ql if trial % 2 else None
        res = a.assign(pred, gt, None, gl, q)
        inds, ov, lab = T.heuristic_assign(pred, gt, gl, q, 5.0 + trial)
        assert torch.equal(res.gt_inds, inds) and torch.equal(res.labels.float(), lab), trial
        assert torch.allclose(res.max_overlaps, ov, atol=1e-6), trial


def test_pipelined_head_refuses_overlap_with_vendor_gemms():
    """runtime.PipelinedHead: more than one batch in flight is only allowed while every dense launch is one of the package's own
    kernels - with vendor GEMMs in the step two overlapping replays can hang the GPU (profiles/r04_d_waymo_two_slots_hang.txt).
    The configuration-level refusal (dense mode 'vendor') is a host-side check made before anything touches the device; the
    trace-level one (round 5: what the warm-up actually handed to the vendor libraries, ops.note_vendor) has its GPU test in
    tests/test_small_batch_gpu.py.  The bf16 mode no longer refuses: it runs on own kernels."""
    import pytest
    from focalformer3d_amd import ops
    from focalformer3d_amd.runtime import PipelinedHead

    class Head(torch.nn.Module):
        training = False
    h = Head().eval()
    x = [torch.zeros(1, 4, 4, 4), [torch.zeros(1, 4, 4, 4)]]
    h.gemm_dtype, h.dense_mode = torch.float32, 'vendor'
    with pytest.raises(ValueError, match='vendor GEMMs'):
        PipelinedHead(h, x, slots=2)
    assert ops.VENDOR_CALLS is None
    ops.note_vendor('x', 1, 2, 3)                       # not tracing: a no-op
    ops.VENDOR_CALLS = []
    try:
        ops.note_vendor('value_proj', 10, 20, 30)
        assert ops.VENDOR_CALLS == [('value_proj', 10, 20, 30)]
    finally:
        ops.VENDOR_CALLS = None


def test_bench_collective_preflight_is_rccl_only():
    """bench.py decides in child processes whether the all-gather can live inside the captured graphs; for any backend other than
    RCCL ('nccl') the answer is no without starting anything (the 2-rank gloo rehearsal on one GPU takes this branch)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('ff3d_bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                           'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.collective_capturable(2, 0, 0, None, 'gloo') is False
    assert callable(bench.preflight_collective) and callable(bench.other_workloads)


def test_gemm_ksplit_rule():
    """Split-K of the long-K GEMMs: one full round of blocks for small grids, the fullest LAST round for a few rounds - the
    32-frame roi_mlp.0 (600 tiles) gets 5 slices, the measured optimum of a 1-8 sweep (profiles/r04_zg_roi_mlp_ab.txt)."""
    from focalformer3d_amd import ops
    assert ops.gemm_ksplit(19200, 512, 37632) == 5
    assert ops.gemm_ksplit(600, 512, 37632) == 25              # 1 frame: 5 x 4 tiles -> 512 // 20 slices
    assert ops.gemm_ksplit(19200, 512, 1024) == 1              # short K: never split
    assert ops.gemm_ksplit(1360800, 768, 256) == 1             # the value GEMM
    for M in (600, 2400, 4800, 9600, 19200, 38400):
        ks = ops.gemm_ksplit(M, 512, 37632)
        assert 1 <= ks <= 64 and 37632 // 32 // ks >= 8


def test_rank_cpu_slices_are_disjoint_and_numa_local():
    """dist.rank_cpu_slice (bench.py pins every rank's host threads with it): shares of the allowed cores are disjoint and cover
    the ranks; with the GPU's NUMA node known a rank stays inside that node; fewer cores than ranks -> one core each, round-robin."""
    from focalformer3d_amd import dist as D
    assert D._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(128))
    shares = [D.rank_cpu_slice(allowed, r, 8) for r in range(8)]
    assert all(len(s) == 16 for s in shares) and sorted(c for s in shares for c in s) == allowed
    node1 = list(range(64, 128))
    on1 = [D.rank_cpu_slice(allowed, r, 8, node_cpus=node1, ranks_on_node=(r - 4, 4)) for r in range(4, 8)]
    assert all(set(s) <= set(node1) and len(s) == 16 for s in on1) and len({c for s in on1 for c in s}) == 64
    assert [D.rank_cpu_slice(range(4), r, 8) for r in range(8)] == [[0], [1], [2], [3], [0], [1], [2], [3]]
    assert D.rank_cpu_slice([], 0, 8) == []
    # a cgroup that hides the GPU's node: fall back to the plain share instead of an empty set
    assert D.rank_cpu_slice(range(16), 1, 2, node_cpus=[64, 65], ranks_on_node=(0, 1)) == list(range(8, 16))


def test_bench_watchdog_prints_an_error_line_instead_of_hanging():
    """bench.Watchdog (VERDICT r04 #5c): a stage that overruns its allowance ends the process with exit code 3 after rank 0
    printed ONE JSON line with "error", the metric and n_gpus; a finished run prints nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(0.4, 0, 8); w.stage('timed region'); "
            "time.sleep(30)") % root
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert 'watchdog' in d['error'] and d['stage'] == 'timed region' and d['n_gpus'] == 8 and d['value'] is None and d['metric']
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(0.3, 0, 1); w.stage('a'); w.stage('b'); "
            "w.finish(); time.sleep(1.0); print('CLEAN')") % root
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == 'CLEAN'
    # SIGTERM from a launcher (another rank died): answered with the same kind of line, also while the main thread is busy
    import signal
    import time
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(60, 0, 4); w.stage('timed region'); "
            "print('READY', flush=True); time.sleep(30)") % root
    p = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == 'READY'
    time.sleep(0.2)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=60)
    d = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
    assert p.returncode == 143 and 'SIGTERM' in d['error'] and d['n_gpus'] == 4


def test_folded_batchnorm_is_kept_per_weight_version():
    """focal_encoder._fold / ConvBNReLU.folded (round 5): without autograd the folded (weight, bias) is one object per weight
    version - what the split-fp16 plane cache of dense_conv3x3 is keyed on - and an in-place weight change (checkpoint load, EMA
    swap) rebuilds it; under autograd nothing is kept (the fold has to stay in the graph)."""
    import torch.nn as nn
    from focalformer3d_amd import focal_encoder as fe
    from focalformer3d_amd.local_attention import ConvBNReLU
    torch.manual_seed(0)
    conv, bn = nn.Conv2d(8, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8)
    bn.running_var.uniform_(0.5, 2.0), bn.running_mean.normal_()
    bn.eval()
    with torch.no_grad():
        w0, b0 = fe._fold(conv, bn)
        w1, b1 = fe._fold(conv, bn)
        assert w1 is w0 and b1 is b0
        ref = conv.weight * (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(-1, 1, 1, 1)
        assert torch.equal(w0, ref)
        bn.weight.mul_(2.0)
        w2, _ = fe._fold(conv, bn)
        assert w2 is not w0 and torch.equal(w2, ref * 2.0)
        bn.running_var.add_(1.0)
        w3, _ = fe._fold(conv, bn)
        assert w3 is not w2 and not torch.equal(w3, w2)
    wg, _ = fe._fold(conv, bn)
    assert wg is not w3 and wg.requires_grad and '_ff3d_fold' in conv.__dict__ and conv.__dict__['_ff3d_fold'][1] is w3
    m = ConvBNReLU(8, 8, 3).eval()
    with torch.no_grad():
        a, _ = m.folded()
        b_, _ = m.folded()
        assert a is b_
        m.conv.weight.add_(1.0)
        c, _ = m.folded()
        assert c is not a


def test_tile_weight_f16_layout_contract():
    """ops.tile_weight_f16 (the weight operand of ff3d_ffn_rows, include/ff3d.h): plane[ks][n][j] = W[n][32 ks + j] * 2^-exp as (hi, lo) with the
    low part UNSCALED - hi + lo reproduces the scaled weight to 2^-22 of the tensor's maximum, every tile complete, no zero row."""
    from focalformer3d_amd import ops
    g = torch.Generator().manual_seed(4)
    w = torch.randn(96, 128, generator=g) * torch.logspace(-3, 0, 96).view(-1, 1)
    t = ops.tile_weight_f16(w)
    assert (t.N, t.K) == (96, 128) and t.hi.shape == t.lo.shape == (4, 96, 32) and t.hi.is_contiguous() and t.lo.is_contiguous()
    assert t.hi.dtype == t.lo.dtype == torch.float16
    sp = ops.split_weight_f16(w)
    assert torch.equal(t.exp, sp.exp)
    for ks in range(4):
        assert torch.equal(t.hi[ks], sp[0][:, 32 * ks:32 * ks + 32])
        lo_s = sp[1][:, 32 * ks:32 * ks + 32].float()
        normal = lo_s.abs() >= 2.0 ** -3                                                             # lo' / 2048 >= 2^-14: a normal fp16 number
        assert torch.equal((t.lo[ks].float() * 2048.0)[normal], lo_s[normal])                       # lo = lo' / 2048, exactly
        assert float((t.lo[ks].float() * 2048.0 - lo_s).abs().max()) <= 2.0 ** -13                  # below that: fp16 subnormal spacing 2^-24 (x 2048)
    scaled = torch.ldexp(w, (-t.exp).expand(w.shape))
    rec = (t.hi.float() + t.lo.float()).permute(1, 0, 2).reshape(96, 128)
    assert float((rec - scaled).abs().max()) <= float(scaled.abs().max()) * 2.0 ** -21
    with pytest.raises(RuntimeError):
        ops.tile_weight_f16(torch.randn(8, 40))                                                      # K % 32


def test_linear_wgrad_slice_rule_and_cpu_route_of_train_linear():
    """Host side of round 6's weight-gradient path.  ff3d_linear_wgrad_slices (the workspace contract of ff3d_linear_wgrad_f16x3:
    slices * (N * K + N) floats) is host arithmetic: 256 / tiles slices, never more than the 32-row steps there are, no empty slice.
    autograd.train_linear below the row threshold, on CPU tensors or without a trainable parameter IS the framework's linear."""
    import torch.nn.functional as F
    from focalformer3d_amd import _lib, autograd as ag
    lib = _lib.load()
    assert lib.ff3d_linear_wgrad_slices(170100, 256, 256) == 127          # 2 tiles -> 128 asked, 5316 steps / 42 per slice = 127 used
    assert lib.ff3d_linear_wgrad_slices(2880, 256, 1024) == 30            # 8 tiles -> 32 asked, 90 steps / 3 per slice
    assert lib.ff3d_linear_wgrad_slices(10, 8, 8) == 1 and lib.ff3d_linear_wgrad_slices(33, 8, 8) == 2
    assert lib.ff3d_linear_wgrad_slices(2400, 37632, 512) == 1            # more tiles than CUs: no row split
    assert lib.ff3d_linear_wgrad_slices(0, 8, 8) == 0
    for M, K, N in ((170100, 256, 256), (999, 36, 20), (64, 4, 4)):
        s = lib.ff3d_linear_wgrad_slices(M, K, N)
        steps = (M + 31) // 32
        per = (steps + s - 1) // s
        assert (s - 1) * per < steps <= s * per
    x = torch.randn(3, 20000, 8, requires_grad=True)
    w, b = torch.randn(12, 8, requires_grad=True), torch.randn(12, requires_grad=True)
    y = ag.train_linear(x, w, b)                                           # CPU tensors: the framework's op, whatever the row count
    assert 'LinearWgrad' not in type(y.grad_fn).__name__
    y.sum().backward()
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
    F.linear(xr, wr, br).sum().backward()
    assert torch.equal(x.grad, xr.grad) and torch.equal(w.grad, wr.grad) and torch.equal(b.grad, br.grad)
    assert ag.WGRAD_MIN_ROWS == int(os.environ.get('FF3D_WGRAD_MIN_ROWS', '16384'))


def test_conv_ksplit_rule_and_the_small_row_dispatch_default():
    """Host-side rules of round 6's third session: K slices of the fp32-output implicit-GEMM conv (ops.conv_ksplit: at most 256 output
    tiles, at least 12 K-steps per slice, at most 512 blocks - the pyramid's stride-2 convs at 1 - 4 frames, never the 32-frame step) and
    the decoder's projections on the own kernels at every row count (transformer.LIN_F16X3_MIN_ROWS = 0 unless FF3D_LIN_MIN_ROWS says so)."""
    from focalformer3d_amd import ops, transformer as TR
    if 'FF3D_CONV_KSPLIT_FORCE' not in os.environ and ops.CONV_KSPLIT:
        for M, N, K, want in ((8100, 256, 2304, 4), (2025, 256, 2304, 6), (2 * 8100, 256, 2304, 2), (4 * 8100, 256, 2304, 1),
                              (4 * 2025, 256, 2304, 4), (8 * 2025, 256, 2304, 2), (32 * 8100, 256, 2304, 1), (32 * 2025, 256, 2304, 1),
                              (2025, 256, 288, 1), (8100, 128, 1152, 3), (900, 64, 576, 1)):
            ks = ops.conv_ksplit(M, N, K)
            assert ks == want, (M, N, K, ks, want)
            tiles, nk = -(-M // 128) * -(-N // 128), K // 32
            assert ks == 1 or (tiles <= ops.CONV_KSPLIT_MAX_TILES and tiles * ks <= ops.CONV_KSPLIT_BLOCKS
                               and nk // ks >= ops.CONV_KSPLIT_MIN_STEPS)
    assert TR.LIN_F16X3_MIN_ROWS == int(os.environ.get('FF3D_LIN_MIN_ROWS', '0'))
