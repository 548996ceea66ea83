"""Registry boundary of the drop-in.

The reference registers its head with ``@HEADS.register_module()`` (focal_decoder.py:33) and builds the
decoder from config type strings through mmcv's ``build_transformer_layer_sequence`` (focal_decoder.py:16,304).
When mmcv / mmdet / mmdet3d are importable our classes are registered into THEIR registries (``force=True``),
so ``dict(type='FocalDecoder', ...)`` from an unchanged reference config builds the MI355X head.  When they are
absent (this image), equivalent minimal registries with the same names keep the same config-driven flow working.
"""


class Registry:
    """The subset of mmcv.utils.Registry the reference configs exercise."""

    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force and self._modules[key] is not cls:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop('type')
        cls = typ if isinstance(typ, type) else self._modules.get(typ)
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        return cls(**args)


def _third_party(path, name):
    try:
        mod = __import__(path, fromlist=[name])
        return getattr(mod, name)
    except Exception:
        return None


def _pick(path, name, fallback_name):
    reg = _third_party(path, name)
    return (reg, True) if reg is not None else (Registry(fallback_name), False)


HEADS, HEADS_IS_MMDET3D = _pick('mmdet3d.models.builder', 'HEADS', 'head')
BBOX_CODERS, _ = _pick('mmdet.core.bbox.builder', 'BBOX_CODERS', 'bbox_coder')
ATTENTION, _ = _pick('mmcv.cnn.bricks.registry', 'ATTENTION', 'attention')
FEEDFORWARD_NETWORK, _ = _pick('mmcv.cnn.bricks.registry', 'FEEDFORWARD_NETWORK', 'feed-forward network')
TRANSFORMER_LAYER, _ = _pick('mmcv.cnn.bricks.registry', 'TRANSFORMER_LAYER', 'transformer layer')
TRANSFORMER_LAYER_SEQUENCE, _ = _pick('mmcv.cnn.bricks.registry', 'TRANSFORMER_LAYER_SEQUENCE',
                                      'transformer layer sequence')


BBOX_ASSIGNERS, _ = _pick('mmdet.core.bbox.builder', 'BBOX_ASSIGNERS', 'bbox_assigner')
MATCH_COST, _ = _pick('mmdet.core.bbox.match_costs.builder', 'MATCH_COST', 'match cost')
IOU_CALCULATORS, _ = _pick('mmdet.core.bbox.iou_calculators.builder', 'IOU_CALCULATORS', 'iou calculator')
LOSSES, _ = _pick('mmdet.models.builder', 'LOSSES', 'loss')


def register(registry):
    """``@register(HEADS)``: register under the class name, overriding a third-party class of that name."""
    def deco(cls):
        registry.register_module(name=cls.__name__, force=True, module=cls)
        return cls
    return deco


def build_head(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d ``builder.build_head`` as the detector calls it (focalformer3d.py:55-59)."""
    default = {}
    if train_cfg is not None:
        default['train_cfg'] = train_cfg
    if test_cfg is not None:
        default['test_cfg'] = test_cfg
    return HEADS.build(cfg, default_args=default or None)


def build_bbox_coder(cfg):
    return BBOX_CODERS.build(cfg)


def build_attention(cfg):
    return ATTENTION.build(cfg)


def build_feedforward_network(cfg):
    return FEEDFORWARD_NETWORK.build(cfg)


def build_transformer_layer(cfg):
    return TRANSFORMER_LAYER.build(cfg)


def build_transformer_layer_sequence(cfg):
    return TRANSFORMER_LAYER_SEQUENCE.build(cfg)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg)


def build_match_cost(cfg):
    return MATCH_COST.build(cfg)


def build_iou_calculator(cfg):
    return IOU_CALCULATORS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)
