"""Drop-in check against the reference's OWN config files (projects/configs/focalformer3d/*.py): every shipped config's
``pts_bbox_head`` / ``imgpts_neck`` dict must build our modules unchanged through the registry, and the parameter layout
(state-dict names + shapes) must equal that of the reference classes built from the same dict (imported from
/root/reference under oracle/ref_shims.py, in a subprocess).  Skipped where the reference tree is absent (GPU box)."""
import glob
import json
import os
import runpy
import subprocess
import sys

import pytest

REF_CFG = '/root/reference/projects/configs/focalformer3d'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = sorted(os.path.basename(f) for f in glob.glob(os.path.join(REF_CFG, '*.py')))

pytestmark = pytest.mark.skipif(not CONFIGS, reason='/root/reference is not present on this machine')


def _model(name):
    return runpy.run_path(os.path.join(REF_CFG, name))['model']


def _layout(m):
    return {k: list(v.shape) for k, v in m.state_dict().items() if 'num_batches_tracked' not in k}


@pytest.mark.parametrize('name', CONFIGS)
def test_reference_config_builds_our_modules_with_the_reference_layout(name):
    import focalformer3d_amd.focal_decoder  # noqa: F401
    from focalformer3d_amd.focal_encoder import NECKS
    from focalformer3d_amd.registry import build_head
    model = _model(name)
    hc = dict(model['pts_bbox_head'])
    tc, te = model.get('train_cfg'), model.get('test_cfg')
    hc.update(train_cfg=tc['pts'] if tc else None, test_cfg=te['pts'] if te else None)       # focalformer3d.py:55-59
    head = build_head(hc)
    if tc:        # the training side of the same dict: assigner, match costs, IoU calculator, losses resolve under the reference's names
        from focalformer3d_amd import training as T
        head._init_assigner_sampler()
        assert isinstance(head.bbox_assigner, T.HungarianAssigner3D) and isinstance(head.bbox_sampler, T.PseudoSampler)
        assert isinstance(head.bbox_assigner.iou_calculator, T.BboxOverlaps3D)
        for attr, cfg_key in (('loss_cls', 'loss_cls'), ('loss_bbox', 'loss_bbox'), ('loss_heatmap', 'loss_heatmap')):
            assert type(getattr(head, attr)).__name__ == model['pts_bbox_head'][cfg_key]['type']
        assert head.add_gt_groups == model['pts_bbox_head'].get('add_gt_groups', 0)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'oracle.ref_config_state_dict', os.path.join(REF_CFG, name)],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('JSON:')][-1][5:])
    assert _layout(head) == ref['head']
    if model.get('imgpts_neck') is not None:
        neck = NECKS.build(dict(model['imgpts_neck']))
        if 'neck' in ref:
            assert _layout(neck) == ref['neck']
        else:
            pytest.skip('reference neck not constructible under the shims: ' + ref.get('neck_error', ''))
