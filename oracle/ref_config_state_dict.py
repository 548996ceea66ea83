"""Test infrastructure (container only): build the REFERENCE FocalDecoder / FocalEncoder from one of the reference's own
config files under the import shims and print the state-dict layout (name -> shape) as JSON.

    python -m oracle.ref_config_state_dict /root/reference/projects/configs/focalformer3d/FocalFormer3D_L.py

Run in a subprocess by tests/test_reference_configs_cpu.py (the shims plant stand-in modules in sys.modules).  Nothing
from the reference is copied: its modules are imported from /root/reference where they lie.
"""
import json
import runpy
import sys

sys.dont_write_bytecode = True


def main(path):
    from oracle import ref_shims as S
    ref = S.load_reference()
    model = runpy.run_path(path)['model']
    hc = dict(model['pts_bbox_head'])
    hc.pop('type')
    tc, te = model.get('train_cfg'), model.get('test_cfg')
    # focalformer3d.py:55-59 injects train_cfg / test_cfg; the assigner built from train_cfg has no parameters and needs
    # mmdet's registry, so the layout is read with train_cfg=None
    hc.update(train_cfg=None, test_cfg=te['pts'] if te else None)
    out = {}
    with S.cpu_device_patch():
        head = ref.FocalDecoder(**hc)
    out['head'] = {k: list(v.shape) for k, v in head.state_dict().items() if 'num_batches_tracked' not in k}
    neck = model.get('imgpts_neck')
    if neck is not None:
        nc = dict(neck)
        nc.pop('type')
        try:
            with S.cpu_device_patch():
                nk = ref.FocalEncoder(**nc)
            out['neck'] = {k: list(v.shape) for k, v in nk.state_dict().items() if 'num_batches_tracked' not in k}
        except Exception as e:           # e.g. a torchvision backbone the shim does not provide
            out['neck_error'] = f'{type(e).__name__}: {e}'
    print('JSON:' + json.dumps(out))


if __name__ == '__main__':
    main(sys.argv[1])
