"""Host logic of the training-mode forward (focalformer3d_amd/train_forward.py + the training routes of transformer.py) on CPU:
the HIP entry points are replaced by the oracle's restatements (tests/train_step_util.oracle_kernels), everything else - module
wiring under autograd, batch-statistics BatchNorm, ground-truth query groups, attention masks, output assembly, loss - is the
product code, compared with one training step executed by the REFERENCE (tests/golden/train_step_*.npz).  The same comparison
runs through libff3d_hip.so in tests/test_train_forward_gpu.py."""
import pytest
import torch

from tests.train_step_util import build_train_head, check_train_step, load_train_step, oracle_kernels, run_train_step


@pytest.mark.parametrize('name,batched', [('train_step_nus', True), ('train_step_waymo', True), ('train_step_nus', False)])
def test_training_step_matches_reference_with_oracle_kernels(name, batched):
    """batched: targets of the whole batch with two host round trips (training.head_get_targets_batched) or the reference's
    per-frame get_targets_single."""
    from focalformer3d_amd import train_forward as TF
    cfg, z = load_train_step(name)
    head = build_train_head(cfg)
    head.batched_targets = batched
    with oracle_kernels(head):
        p0, losses, grads, gin = run_train_step(head, z, 'cpu', forward=TF.forward_train)
        if cfg['head'].get('add_gt_groups', 0):
            assert 'center_gtgroups' in p0 and p0['batch_valid_gt_mask'].dtype == torch.bool
        check_train_step(z, p0, losses, grads, gin, head)


def test_bev_corners_match_mmdet3d_order():
    """The four BEV corners the ground-truth groups are built from (FD:397) vs the shim's restatement of mmdet3d's
    LiDARInstance3DBoxes.corners."""
    from focalformer3d_amd.train_forward import bev_corners
    from oracle.ref_shims import LiDARInstance3DBoxes
    g = torch.Generator().manual_seed(0)
    t = torch.rand(17, 7, generator=g) * torch.tensor([80., 80, 2, 4, 6, 2, 6.2]) - torch.tensor([40., 40, 2, -0.3, -0.3, -1, 3.1])
    ref = LiDARInstance3DBoxes(t).corners.reshape(-1, 4, 2, 3)[:, :4, 0, :2]
    assert torch.allclose(bev_corners(t), ref, atol=1e-5)
