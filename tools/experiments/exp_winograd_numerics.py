"""CPU experiment (no GPU): would a Winograd F(2x2, 3x3) form of the 256 -> 256 heatmap conv keep fp32-class accuracy on the
split-fp16 arithmetic?  Emulates the arithmetic of csrc/splitmm.hip with torch-CPU fp32 matmuls: operands as (hi, lo') fp16
pairs with a power-of-two range normalisation, three products hi*hi + (hi*lo' + lo'*hi) / 2048 accumulated in fp32.
Direct form: 9 taps x C channels.  Winograd: V = B^T d B (fp32, from the 22-bit reconstructed inputs), U = G g G^T (fp64 ->
fp32), both split to pairs, 16 channel-contractions, Y = A^T M A in fp32.  Errors are max |y - ref| / max |ref| against fp64.
    python tools/experiments/exp_winograd_numerics.py [C] [HW]"""
import sys

import torch
import torch.nn.functional as F


def split(x):
    """fp32 tensor -> (hi, lo') fp16 pair values (as fp32) + exponent: x ~= 2^e (hi + lo'/2048), max|x| 2^-e in [2^13, 2^14)."""
    m = x.abs().max().item()
    e = 0 if m == 0 else int(torch.frexp(torch.tensor(m))[1]) - 14
    xs = torch.ldexp(x, torch.tensor(-e))
    hi = xs.half()
    lo = ((xs - hi.float()) * 2048.0).half()
    return hi.float(), lo.float(), e


def mm3(a, b):
    """(M, K) x (N, K)^T with the three-pass split arithmetic, fp32 accumulation."""
    ah, al, ea = split(a)
    bh, bl, eb = split(b)
    main = ah @ bh.t()
    cross = ah @ bl.t() + al @ bh.t()
    return torch.ldexp(main + cross / 2048.0, torch.tensor(ea + eb))


def direct(x, w):
    """x (C, H, W), w (N, C, 3, 3) -> (N, H-2, W-2) valid conv as an implicit GEMM over (tap, channel)."""
    C, H, W = x.shape
    cols = F.unfold(x[None], 3)[0].t().contiguous()           # (P, C*9)
    return mm3(cols, w.reshape(w.shape[0], -1)).t().reshape(w.shape[0], H - 2, W - 2)


def winograd(x, w, quantise_input=True):
    C, H, W = x.shape
    N = w.shape[0]
    if quantise_input:                                        # the map arrives as a 22-bit pair
        h, l, e = split(x)
        x = torch.ldexp(h + l / 2048.0, torch.tensor(e))
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    U = (G @ w.double() @ G.t()).float()                      # (N, C, 4, 4)
    th, tw = (H - 2) // 2, (W - 2) // 2
    tiles = x.unfold(1, 4, 2).unfold(2, 4, 2)                 # (C, th, tw, 4, 4)
    V = Bt @ tiles @ Bt.t()                                   # fp32
    M = torch.empty(th * tw, N, 4, 4)
    for i in range(4):
        for j in range(4):
            M[:, :, i, j] = mm3(V[..., i, j].reshape(C, -1).t().contiguous(), U[:, :, i, j].contiguous())
    Y = At @ M @ At.t()                                       # (T, N, 2, 2)
    return Y.reshape(th, tw, N, 2, 2).permute(2, 0, 3, 1, 4).reshape(N, th * 2, tw * 2)


def main(C=256, HW=34):
    g = torch.Generator().manual_seed(0)
    rows = []
    for tag, xs, ws in (('N(0,1) x, 0.02 N(0,1) w', 1.0, 0.02), ('relu-like x >= 0', None, 0.02)):
        x = torch.randn(C, HW, HW, generator=g) * (xs or 1.0)
        if xs is None:
            x = x.relu() * 2
        w = torch.randn(C, C, 3, 3, generator=g) * ws
        ref = F.conv2d(x.double()[None], w.double())[0]
        sc = ref.abs().max()
        f32 = F.conv2d(x[None], w)[0]
        e = lambda y: float((y.double() - ref).abs().max() / sc)          # noqa: E731
        rows.append((tag, e(f32), e(direct(x, w)), e(winograd(x, w)), e(winograd(x, w, quantise_input=False))))
    print(f'C = {C}, {HW}x{HW} map; max |y - fp64| / max |fp64|')
    print(f'{"case":28s} {"torch fp32":>11s} {"direct 3-pass":>14s} {"winograd 3-pass":>16s} {"(exact input)":>14s}')
    for r in rows:
        print(f'{r[0]:28s} {r[1]:11.2e} {r[2]:14.2e} {r[3]:16.2e} {r[4]:14.2e}')


if __name__ == '__main__':
    main(*(int(v) for v in sys.argv[1:3]))
