"""Does the training step read uninitialised memory?  Fill the caching allocator's free blocks with a value first.
argv: golden name, variant (default | grid_sample | per_frame_targets | nopoison)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.train_step_util import build_train_head, load_train_step, run_train_step
name, variant = sys.argv[1], sys.argv[2]
if variant != 'nopoison':
    blocks = [torch.full((n,), 1.0, device='cuda') for n in (2 ** 28, 2 ** 26, 2 ** 24, 2 ** 22, 2 ** 20, 2 ** 18, 2 ** 16, 2 ** 14, 2 ** 12, 2 ** 10) for _ in range(3)]
    del blocks
cfg, z = load_train_step(name)
head = build_train_head(cfg)
if variant == 'grid_sample':
    head.train_roi_sampler = 'grid_sample'
if variant == 'per_frame_targets':
    head.batched_targets = False
p0, losses, grads, gin = run_train_step(head, z, 'cuda')
perr = {}
for key in z.files:
    if key.startswith('pred/'):
        parts = key.split('/')
        ours = p0[parts[1]] if len(parts) == 2 else p0[parts[1]][int(parts[2])]
        ref = torch.from_numpy(z[key]).float()
        perr[key] = float((ours.detach().float().cpu() - ref).abs().max())
print(variant, 'worst pred err', sorted(perr.items(), key=lambda kv: -kv[1])[:3])
print(variant, 'losses', {k: (round(float(v), 5), round(float(z['loss/' + k]), 5)) for k, v in losses.items()})
errs = []
for key in z.files:
    if key.startswith('grad/') and grads[key[5:]] is not None:
        ref = torch.from_numpy(z[key]); g = grads[key[5:]].cpu()
        if float(ref.abs().max()) > 1e-6:
            errs.append((float((g - ref).abs().max()) / float(ref.abs().max()), key))
errs.sort(reverse=True)
print(variant, 'worst grads', [(round(e, 5), k) for e, k in errs[:8]], 'n>1e-3:', sum(e > 1e-3 for e, _ in errs), 'of', len(errs), flush=True)
