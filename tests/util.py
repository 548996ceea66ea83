"""Shared helpers for the parity tests: golden-fixture loading and index-set matching."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    sd, inp, out = {}, {}, {}
    for k in z.files:
        v = z[k]
        if k.startswith('sd/'):
            sd[k[3:]] = torch.from_numpy(v)
        elif k.startswith('in/'):
            inp[k[3:]] = torch.from_numpy(v)
        elif k.startswith('out/'):
            out[k[4:]] = torch.from_numpy(v)
    cfg = json.loads(bytes(z['cfg']).decode()) if 'cfg' in z.files else None
    return cfg, sd, inp, out, z


def oracle_cfg(cfg):
    from oracle import ff3d_oracle as O
    keys = ('num_proposals hidden_channel num_classes num_decoder_layers num_heads nms_kernel_size multiscale '
            'multistage_heatmap reuse_first_heatmap extra_feat bevpos input_img iterbev_wo_img mask_heatmap_mode '
            'roi_feats roi_expand_ratio roi_based_reg dataset pc_range voxel_size out_size_factor '
            'post_center_range score_threshold').split()
    kw = {k: cfg[k] for k in keys}
    kw['common_heads'] = {k: tuple(v) for k, v in cfg['common_heads'].items()}
    return O.head_config(**kw)


def head_inputs(cfg, inp):
    n = sum(1 for k in inp if k.startswith('stage_'))
    maps = [inp[f'stage_{i}'] for i in range(n)]
    second = maps if cfg['multistage_heatmap'] else maps[0]
    return [inp['pts_feat_conv'], second]


def stage_perm(ref_idx, our_idx):
    """Permutation p with ref_idx[:, p] == our_idx per row, asserting equal index SETS
    (the reference's top-k order is implementation-defined, FD:688)."""
    perms = []
    for r, o in zip(ref_idx, our_idx):
        assert torch.equal(torch.sort(r).values, torch.sort(o).values), 'top-k index sets differ'
        pos = {int(v): i for i, v in enumerate(r.tolist())}
        perms.append(torch.tensor([pos[int(v)] for v in o.tolist()]))
    return torch.stack(perms)
