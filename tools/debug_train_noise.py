"""Run the training-step golden comparison several times and print the worst gradient error (relative to the largest entry)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.train_step_util import build_train_head, load_train_step, run_train_step
name = sys.argv[1]
for it in range(4):
    cfg, z = load_train_step(name)
    head = build_train_head(cfg)
    p0, losses, grads, gin = run_train_step(head, z, 'cuda')
    worst, wn = 0.0, ''
    for key in z.files:
        if key.startswith('grad/') and grads[key[5:]] is not None:
            ref = torch.from_numpy(z[key]); g = grads[key[5:]].cpu()
            e = float((g - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
            if e > worst and float(ref.abs().max()) > 1e-6: worst, wn = e, key
    print(name, os.environ.get('FF3D_TRAIN_SDPA', 'math'), 'run', it, 'worst grad rel err %.2e' % worst, wn, flush=True)
