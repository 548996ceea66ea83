"""Root-cause probe for the suite-order dependent gradient deviation of the training-step golden test.

Runs, IN ONE PROCESS (so that every piece of process state survives): the GPU test files named on the command line through
``pytest.main`` and then the training-step comparison with a per-tensor error report, followed by probes that each change ONE
suspect (allocator cache, MIOpen, SDPA backend, a second identical run) and report again.  The kernel-name census of the
step (torch.profiler) is written next to the report, so a fresh-process run and a late-in-suite run can be diffed.

    python tools/debug_suite_order.py OUT_DIR [test files ...]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.train_step_util import build_train_head, load_train_step, run_train_step  # noqa: E402


def step_errors(name, tag, out_dir, profile=False, mutate=None):
    cfg, z = load_train_step(name)
    head = build_train_head(cfg)
    if mutate:
        mutate(head)
    prof = None
    if profile:
        from torch.profiler import ProfilerActivity, profile as tprofile
        prof = tprofile(activities=[ProfilerActivity.CUDA])
        prof.__enter__()
    p0, losses, grads, gin = run_train_step(head, z, 'cuda')
    torch.cuda.synchronize()
    if prof is not None:
        prof.__exit__(None, None, None)
        census = {}
        for ev in prof.key_averages():
            census[ev.key] = census.get(ev.key, 0) + ev.count
        json.dump(census, open(os.path.join(out_dir, f'census_{name}_{tag}.json'), 'w'), indent=0, sort_keys=True)
    errs = []
    for key in z.files:
        if key.startswith('grad/') and grads[key[5:]] is not None:
            ref = torch.from_numpy(z[key])
            g = grads[key[5:]].detach().cpu()
            m = float(ref.abs().max())
            if m > 1e-6:
                errs.append((float((g - ref).abs().max()) / m, key[5:]))
    for i, g in enumerate(gin):
        ref = torch.from_numpy(z[f'gin/{i}'])
        errs.append((float((g.cpu() - ref).abs().max()) / float(ref.abs().max()), f'gin/{i}'))
    perr = 0.0
    for key in z.files:
        if key.startswith('pred/'):
            parts = key.split('/')
            ours = p0[parts[1]] if len(parts) == 2 else p0[parts[1]][int(parts[2])]
            ref = torch.from_numpy(z[key]).float()
            perr = max(perr, float((ours.detach().float().cpu() - ref).abs().max()))
    errs.sort(reverse=True)
    bad = [(round(e, 6), k) for e, k in errs if e > 2e-4]
    st = torch.cuda.memory_stats()
    print(f'[{tag}] {name}: worst pred err {perr:.2e}; {len(bad)} of {len(errs)} gradient tensors off by > 2e-4 of their max; '
          f'worst {errs[0][0]:.2e} {errs[0][1]}; reserved {st["reserved_bytes.all.current"] / 2**30:.1f} GiB '
          f'allocated {st["allocated_bytes.all.current"] / 2**30:.2f} GiB', flush=True)
    for e, k in bad:
        print(f'      {e:.4e}  {k}', flush=True)
    return bad, grads


def flags(tag):
    print(tag, 'flags: matmul.allow_tf32', torch.backends.cuda.matmul.allow_tf32, 'cudnn.allow_tf32', torch.backends.cudnn.allow_tf32,
          'float32_matmul_precision', torch.get_float32_matmul_precision(), 'cudnn.benchmark', torch.backends.cudnn.benchmark,
          'cudnn.enabled', torch.backends.cudnn.enabled, 'deterministic', torch.are_deterministic_algorithms_enabled(),
          'default dtype', torch.get_default_dtype(), 'grad enabled', torch.is_grad_enabled(),
          'env', {k: v for k, v in os.environ.items() if k.startswith(('FF3D', 'MIOPEN', 'HIPBLASLT', 'ROCBLAS', 'TORCH', 'PYTORCH'))}, flush=True)


def main():
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    files = sys.argv[2:]
    names = ('train_step_nus', 'train_step_waymo')
    flags('start')
    if os.environ.get('FF3D_DIAG_FRESH_FIRST') == '1':
        for n in names:
            step_errors(n, 'fresh', out_dir, profile=True)
    if files:
        import pytest
        rc = pytest.main(['-q', '-m', 'gpu', '-p', 'no:cacheprovider'] + files)
        print('pytest.main rc', rc, flush=True)
    flags('after suite')
    any_bad = False
    for n in names:
        bad, _ = step_errors(n, 'after_suite', out_dir, profile=True)
        any_bad |= bool(bad)
    if not any_bad:
        print('no deviation in this process state', flush=True)
        return
    probes(out_dir)


def probes(out_dir, names=('train_step_nus', 'train_step_waymo')):
    """The step again, then again with ONE suspect changed at a time (called by tests/train_step_util.py when a gradient misses)."""
    os.makedirs(out_dir, exist_ok=True)
    flags('probes')
    for n in names:
        step_errors(n, 'as_is', out_dir, profile=True)
    for n in names:
        step_errors(n, 'again', out_dir)
    # one suspect at a time
    torch.cuda.empty_cache()
    for n in names:
        step_errors(n, 'after_empty_cache', out_dir)
    for n in names:
        step_errors(n, 'sdpa_math', out_dir, mutate=lambda h: [setattr(m, 'train_sdpa', 'math') for m in h.modules()])
    torch.backends.cudnn.enabled = False
    for n in names:
        step_errors(n, 'miopen_off', out_dir, profile=True)
    torch.backends.cudnn.enabled = True
    for n in names:
        step_errors(n, 'roi_grid_sample', out_dir, mutate=lambda h: setattr(h, 'train_roi_sampler', 'grid_sample'))


if __name__ == '__main__':
    main()
