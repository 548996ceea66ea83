"""Controlled experiment behind the round-3 root cause of the "suite-order dependent" training-step gradient deviation
(VERDICT r02 weak #1): the location gradient of bilinear sampling is discontinuous where a pixel coordinate crosses an integer.

Runs the training-step comparison on CPU (product host code + the oracle's kernels, tests/train_step_util.oracle_kernels) with
every sampling location of the deformable attention scaled by (1 + f), f = a few fp32 ulps, once WITHOUT and once WITH the
fixture's de-singularised coordinates injected (tests/train_step_util.injected_sampling_locations).

    python tools/experiments/exp_loc_kink.py [golden name]

On the round-2 fixture (no injection) f = +2e-7 gave 121 of 221 gradient tensors off by more than 2e-4 of their maximum with
EXACTLY the digits the MI355X run showed late in the suite (0.03862 / 0.03822 sampling_offsets bias / weight of
decoder.1.layers.1, 0.00944 norms.0.weight, 0.00907 dconv2.bn.weight ...; profiles/r03_b_train_step_gradient_miss.json).
"""
import contextlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import train_forward as TF          # noqa: E402
from focalformer3d_amd import transformer as T            # noqa: E402
from tests import train_step_util as U                      # noqa: E402


def run(name, f, inject):
    cfg, z = U.load_train_step(name)
    head = U.build_train_head(cfg)
    orig = T.MultiScaleDeformableAttention.forward_train_bf

    def perturbed(self, x, value_cl, pos, reference_points, level_hw):
        return orig(self, x, value_cl, pos, reference_points * (1.0 + f), level_hw)
    T.MultiScaleDeformableAttention.forward_train_bf = perturbed
    saved = U.injected_sampling_locations
    if not inject:
        U.injected_sampling_locations = lambda z_: contextlib.nullcontext()
    try:
        with U.oracle_kernels(head):
            p0, losses, grads, gin = U.run_train_step(head, z, 'cpu', forward=TF.forward_train)
    finally:
        T.MultiScaleDeformableAttention.forward_train_bf = orig
        U.injected_sampling_locations = saved
    errs = []
    for key in z.files:
        if key.startswith('grad/') and grads[key[5:]] is not None:
            ref = torch.from_numpy(z[key])
            m = float(ref.abs().max())
            if m > 1e-6:
                errs.append((float((grads[key[5:]] - ref).abs().max()) / m, key[5:]))
    errs.sort(reverse=True)
    print(f'{name}  f = {f:+.0e}  injection {"on " if inject else "off"}: {sum(e > 2e-4 for e, _ in errs):3d} of {len(errs)} '
          f'gradient tensors off by > 2e-4 of their maximum; worst {errs[0][0]:.2e} {errs[0][1]}', flush=True)


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'train_step_waymo'
    for inject in (False, True):
        for f in (0.0, 2e-7, -2e-7, 5e-7, -5e-7, 2e-6, -2e-6):
            run(name, f, inject)
