"""Python operators over the C ABI (include/ff3d.h).  PyTorch is plumbing here: it owns device
memory and the stream; every op is one hand-written gfx950 kernel launch.

Every wrapper validates device / dtype / contiguity and raises - there is no fallback path.
Reference call sites are cited per function (FD = focal_decoder.py, EU = encoder_utils.py,
UT = utils.py, BC = transfusion_bbox_coder.py of the reference plugin).
"""
import ctypes as C
import os

import torch

from . import _lib

HIST_BINS = 4096

# Optional kernel timing hook (bench.py): when set to a list, the deformable-attention launches are
# bracketed by HIP events recorded on the launch stream and (start, end, algorithmic_bytes) is appended.
MSDA_EVENTS = None
GATHER_EVENTS = None          # ff3d_msda_gather_rows launches (value mode 'gather_first'): (start, end, requested bytes, unique bytes)
# Same hook for the split-fp16 dense kernel (conv3x3_f16x3 / gemm_f16x3): (start, end, tag, algorithmic fp32 flops).
DENSE_EVENTS = None
# stride-1 wide convs: halo-tile kernel (convhalo.hip) or implicit GEMM (splitmm.hip): 'auto' (by size), '1' (always), '0' (never)
CONV_HALO = os.environ.get('FF3D_CONV_HALO', 'auto')
GEMM_KSPLIT = os.environ.get('FF3D_GEMM_KSPLIT', '1') != '0'
# query self-attention on the fp16 matrix cores (attn_f16x3.hip) unless the dense mode is 'vendor' (then exact-fp32 MFMA, attn.hip)
ATTN_F16X3 = os.environ.get('FF3D_DENSE_MODE', 'f16x3') != 'vendor'


# Vendor-GEMM trace: every place that hands a dense layer to hipBLASLt / MIOpen reports it here; runtime.PipelinedHead
# reads the list after its capture warm-up and refuses overlapping replays when one fired (a vendor kernel that spin-waits on its own
# grid - stream-K - deadlocks beside another graph's kernels, profiles/r04_d_waymo_two_slots_hang.txt).  None = not tracing.
VENDOR_CALLS = None


def note_vendor(what, M=0, N=0, K=0):
    if VENDOR_CALLS is not None:
        VENDOR_CALLS.append((what, int(M), int(N), int(K)))


def msda_algorithmic_bytes(B, Nq, heads, Dh, L, P, value_bytes=4, out_bytes=4):
    """corner reads + loc/weight reads + output write, per launch."""
    return B * Nq * heads * L * P * (4 * Dh * value_bytes + 12) + B * Nq * heads * Dh * out_bytes


def _stream():
    # the raw handle of torch's current stream on the current device (torch.cuda.current_stream() builds a Stream object per
    # call: ~9 us, 110 calls per decoder step - round 3 host profile, profiles/r03_k_host_profile_b4.txt)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _chk(t, dtype=torch.float32, name='tensor'):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'{name}: expected a CUDA (HIP) tensor - the HIP decoder path has no CPU fallback')
    if t.dtype != dtype:
        raise RuntimeError(f'{name}: expected dtype {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise RuntimeError(f'{name}: expected a contiguous tensor')
    return C.c_void_p(t.data_ptr())


def _opt(t, dtype=torch.float32, name='tensor'):
    return C.c_void_p(0) if t is None else _chk(t, dtype, name)


def _levels(level_hw):
    arr = (C.c_int32 * (2 * len(level_hw)))(*[int(v) for hw in level_hw for v in hw])
    return arr, len(level_hw)


def _floats(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def small_class_bits(dataset, num_classes):
    """FD:564-569: classes whose NMS / dilation kernel is 1."""
    small = {'nuScenes': (8, 9), 'Waymo': (1, 2)}[dataset]
    bits = 0
    for c in small:
        if c < num_classes:
            bits |= 1 << c
    return bits


def msda_fwd(value, level_hw, loc, attn_w, out=None):
    """mmcv ``ms_deform_attn_forward`` (reached from FD:927-933).  value (B,Nv,heads,Dh) fp32|bf16,
    loc (B,Nq,heads,L,P,2), attn_w (B,Nq,heads,L,P) -> (B,Nq,heads*Dh) fp32."""
    lib = _lib.load()
    B, Nv, M, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    dt = {torch.float32: 0, torch.bfloat16: 1}[value.dtype]
    if out is None:
        out = torch.empty(B, Nq, M * D, device=value.device, dtype=torch.float32)
    lv, nl = _levels(level_hw)
    assert nl == L
    st = lib.ff3d_msda_fwd(_chk(value, value.dtype, 'value'), dt, _chk(loc, name='loc'), _chk(attn_w, name='attn_w'),
                           _chk(out, name='out'), B, Nv, Nq, M, D, L, P, lv, _stream())
    _lib.check(st, 'ff3d_msda_fwd')
    return out


def msda_fwd_dev(value, spatial_shapes, level_start_index, loc, attn_w, out=None):
    """mmcv ``ms_deform_attn_forward`` with mmcv's own arguments: ``spatial_shapes`` (L,2) / ``level_start_index`` (L) are
    int64 DEVICE tensors (FD:837-841) and are never copied to the host - no synchronisation, graph-capturable."""
    lib = _lib.load()
    B, Nv, M, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    if tuple(spatial_shapes.shape) != (L, 2) or level_start_index.numel() != L:
        raise RuntimeError('spatial_shapes must be (L, 2) and level_start_index (L) for L = loc.shape[3]')
    dt = {torch.float32: 0, torch.bfloat16: 1}[value.dtype]
    if out is None:
        out = torch.empty(B, Nq, M * D, device=value.device, dtype=torch.float32)
    st = lib.ff3d_msda_fwd_dev(_chk(value, value.dtype, 'value'), dt, _chk(spatial_shapes, torch.int64, 'spatial_shapes'),
                               _chk(level_start_index, torch.int64, 'level_start_index'), _chk(loc, name='loc'),
                               _chk(attn_w, name='attn_w'), _chk(out, name='out'), B, Nv, Nq, M, D, L, P, _stream())
    _lib.check(st, 'ff3d_msda_fwd_dev')
    return out


def msda_bwd(value, level_hw, loc, attn_w, grad_out):
    """Backward of msda_fwd (mmcv ms_deform_attn_backward): value (B,Nv,heads,Dh), loc (B,Nq,heads,L,P,2), attn_w
    (B,Nq,heads,L,P), grad_out (B,Nq,heads*Dh) -> (grad_value, grad_loc, grad_attn_w)."""
    lib = _lib.load()
    B, Nv, M, D = value.shape
    Nq = loc.shape[1]
    L, P = loc.shape[3], loc.shape[4]
    gv = torch.zeros_like(value)
    gl = torch.empty_like(loc)
    gw = torch.empty_like(attn_w)
    lv, _ = _levels(level_hw)
    st = lib.ff3d_msda_bwd(_chk(value, name='value'), _chk(loc, name='loc'), _chk(attn_w, name='attn_w'),
                           _chk(grad_out, name='grad_out'), _chk(gv), _chk(gl), _chk(gw), B, Nv, Nq, M, D, L, P, lv, _stream())
    _lib.check(st, 'ff3d_msda_bwd')
    return gv, gl, gw


def msda_fused_fwd(value, level_hw, ref_pts, off, logits, P, out=None):
    """MSDA with softmax + ``ref + off/(W,H)`` fused.  value (B,Nv,heads,Dh) - dense, or a column block of
    a wider (B,Nv,n*heads*Dh) GEMM output viewed as (B,Nv,heads,Dh) (only the cell stride may be larger);
    ref_pts (B,Nq,2); off / logits: 2-D row views (B*Nq, heads*L*P*2) / (B*Nq, heads*L*P) with unit inner
    stride (column blocks of one GEMM output are fine)."""
    lib = _lib.load()
    B, Nv, M, D = value.shape
    if not (value.is_cuda and value.stride(3) == 1 and value.stride(2) == D and value.stride(0) == Nv * value.stride(1)):
        raise RuntimeError('value: expected a CUDA (B,Nv,heads,Dh) tensor, dense in (heads,Dh), batch stride Nv*cell stride')
    value_ld = value.stride(1)
    Nq = ref_pts.shape[1]
    L = len(level_hw)
    dt = {torch.float32: 0, torch.bfloat16: 1}[value.dtype]
    for t, w, n in ((off, M * L * P * 2, 'off'), (logits, M * L * P, 'logits')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape == (B * Nq, w) and t.stride(1) == 1):
            raise RuntimeError(f'{n}: expected a CUDA fp32 (B*Nq, {w}) view with unit inner stride')
    if out is None:
        out = torch.empty(B, Nq, M * D, device=value.device, dtype=torch.float32)
    lv, _ = _levels(level_hw)
    ev = None
    if MSDA_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    st = lib.ff3d_msda_fused_fwd(C.c_void_p(value.data_ptr()), dt, value_ld, _chk(ref_pts, name='ref_pts'),
                                 C.c_void_p(off.data_ptr()), off.stride(0), C.c_void_p(logits.data_ptr()),
                                 logits.stride(0), _chk(out, name='out'), B, Nv, Nq, M, D, L, P, lv, _stream())
    if ev is not None:
        ev[1].record()
        MSDA_EVENTS.append((ev[0], ev[1], msda_algorithmic_bytes(B, Nq, M, D, L, P, value.element_size())))
    _lib.check(st, 'ff3d_msda_fused_fwd')
    return out


def msda_gather_rows(value_cl, level_hw, ref_pts, off, logits, P, heads, groups=1):
    """ff3d_msda_gather_rows (the opt-in 'gather_first' value mode): value_cl (B, Nv, C) fp32 UN-projected, ref_pts (B, Nq, 2), off /
    logits = column blocks of the (B*Nq, heads*L*P*3) projection (row-strided views) -> (B*Nq, heads*C + 32*groups) fp32, ``groups``
    column groups of heads/groups heads: [the group's C-wide weighted sums | its sums of valid weights | zero padding to 32]."""
    lib = _lib.load()
    B, Nv, C_ = value_cl.shape
    Nq = ref_pts.shape[1]
    lv, L = _levels(level_hw)
    for t, name in ((off, 'off'), (logits, 'logits')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == B * Nq):
            raise RuntimeError(f'{name}: expected a (B*Nq, n) fp32 CUDA tensor with unit column stride')
    out = torch.empty(B * Nq, heads * C_ + 32 * groups, device=value_cl.device)
    ev = None
    if GATHER_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    st = lib.ff3d_msda_gather_rows(_chk(value_cl, name='value'), _chk(ref_pts, name='ref_pts'), C.c_void_p(off.data_ptr()),
                                   off.stride(0), C.c_void_p(logits.data_ptr()), logits.stride(0), _chk(out), out.shape[1],
                                   int(groups), B, Nv, Nq, heads, C_, L, P, lv, _stream())
    _lib.check(st, 'ff3d_msda_gather_rows')
    if ev is not None:
        ev[1].record()
        # bytes this form REQUESTS: every corner is a C-wide row per head (served by L2 / MALL: the unique bytes are the maps, B*Nv*C*4)
        GATHER_EVENTS.append((ev[0], ev[1], B * Nq * heads * L * P * (4 * C_ * 4 + 12) + out.numel() * 4, B * Nv * C_ * 4 + out.numel() * 4))
    return out


def self_attention(q, k, v, heads, scale=None, f16x3=None):
    """softmax(scale * q k^T) v per (frame, head).  q, k, v: (B,N,heads*Dh) views with unit inner stride and
    row stride = stride(1) (column blocks of wider GEMM outputs are fine) -> (B,N,heads*Dh) contiguous."""
    lib = _lib.load()
    B, N, C_ = q.shape
    Dh = C_ // heads
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.shape == (B, N, C_) and t.stride(2) == 1
                and t.stride(0) == N * t.stride(1)):
            raise RuntimeError(f'{n}: expected a CUDA fp32 (B,N,C) view with unit inner stride and batch stride N*row stride')
    out = torch.empty(B, N, C_, device=q.device)
    # fp16 matrix cores with (hi, lo') operand pairs (fp32-class) when the head size allows, exact-fp32 MFMA otherwise
    use_f16x3 = ATTN_F16X3 if f16x3 is None else f16x3          # per-module choice (set_dense_mode); the global is the default
    fn = lib.ff3d_self_attention_f16x3 if (use_f16x3 and Dh in (16, 32)) else lib.ff3d_self_attention
    st = fn(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), _chk(out),
            B, N, heads, Dh, q.stride(1), k.stride(1), v.stride(1), C_,
            float(scale if scale is not None else Dh ** -0.5), _stream())
    _lib.check(st, 'ff3d_self_attention')
    return out


def _rows(t, heads, name):
    """(B, N, heads*Dh) view with unit inner stride -> (pointer, row stride)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1)):
        raise RuntimeError(f'{name}: expected a CUDA fp32 (B, N, C) tensor whose rows are evenly strided')
    return C.c_void_p(t.data_ptr()), t.stride(1)


def mha_train_fwd(q, k, v, heads, mask=None, keep=None, keep_scale=1.0, scale=None):
    """Training-route self-attention core (ff3d_mha_train_fwd): q, k, v (B, N, C) (column blocks allowed), mask (B, N, N)
    uint8 non-zero = blocked, keep (B, heads, N, N) uint8 dropout keep-mask -> (out (B, N, C), lse (B, heads, N))."""
    lib = _lib.load()
    B, N, C_ = q.shape
    Dh = C_ // heads
    out = torch.empty(B, N, C_, device=q.device)
    lse = torch.empty(B, heads, N, device=q.device)
    (qp, lq), (kp, lk), (vp, lv) = _rows(q, heads, 'q'), _rows(k, heads, 'k'), _rows(v, heads, 'v')
    st = lib.ff3d_mha_train_fwd(qp, kp, vp, _opt(mask, torch.uint8, 'mask'), _opt(keep, torch.uint8, 'keep'), float(keep_scale),
                                _chk(out), _chk(lse), B, N, heads, Dh, lq, lk, lv, C_,
                                float(scale if scale is not None else Dh ** -0.5), _stream())
    _lib.check(st, 'ff3d_mha_train_fwd')
    return out, lse


def mha_train_bwd(q, k, v, heads, out, lse, grad_out, mask=None, keep=None, keep_scale=1.0, scale=None):
    """Backward of mha_train_fwd -> (grad_q, grad_k, grad_v) (B, N, C) contiguous."""
    lib = _lib.load()
    B, N, C_ = q.shape
    Dh = C_ // heads
    gq, gk, gv = (torch.empty(B, N, C_, device=q.device) for _ in range(3))
    ws = torch.empty(B, heads, N, device=q.device)
    (qp, lq), (kp, lk), (vp, lv), (gp, lg) = _rows(q, heads, 'q'), _rows(k, heads, 'k'), _rows(v, heads, 'v'), _rows(grad_out, heads, 'grad_out')
    st = lib.ff3d_mha_train_bwd(qp, kp, vp, _opt(mask, torch.uint8, 'mask'), _opt(keep, torch.uint8, 'keep'), float(keep_scale),
                                _chk(out), _chk(lse), gp, _chk(gq), _chk(gk), _chk(gv), _chk(ws), B, N, heads, Dh,
                                lq, lk, lv, C_, lg, C_, C_, C_, float(scale if scale is not None else Dh ** -0.5), _stream())
    _lib.check(st, 'ff3d_mha_train_bwd')
    return gq, gk, gv


def add_layer_norm(a, b, gamma, beta, eps=1e-5, pos=None):
    """LayerNorm(a + b) over the last dim (b may be None); with ``pos`` also returns the normalised rows + pos."""
    lib = _lib.load()
    C_ = a.shape[-1]
    rows = a.numel() // C_
    out = torch.empty_like(a)
    out_pos = torch.empty_like(a) if pos is not None else None
    st = lib.ff3d_add_layer_norm(_chk(a, name='a'), _opt(b, name='b'), _chk(gamma, name='gamma'), _chk(beta, name='beta'),
                                 _opt(pos, name='pos'), _chk(out), _opt(out_pos), rows, C_, float(eps), _stream())
    _lib.check(st, 'ff3d_add_layer_norm')
    return (out, out_pos) if pos is not None else out


def bias_relu_(x, bias=None, upper=0.0):
    """In-place min(relu(x + bias[c]), upper) on an (N, C, H, W) / (N, C, L) map (upper <= 0: no upper clamp)."""
    lib = _lib.load()
    N, C_ = x.shape[:2]
    HW = x[0, 0].numel()
    st = lib.ff3d_bias_relu(_chk(x, name='x'), _opt(bias, name='bias'), N, C_, HW, float(upper), _stream())
    _lib.check(st, 'ff3d_bias_relu')
    # the kernel writes through the raw pointer: torch's version counter does not move, so a pair a producer left on this tensor
    # (``_ff3d_pair``, honoured while ``_version == 0``) would go stale silently - drop it
    x.__dict__.pop('_ff3d_pair', None)
    x.__dict__.pop('_ff3d_exp', None)
    return x


def linear_relu(x, weight, bias):
    """relu(x @ weight^T + bias) as ONE hipBLASLt GEMM with the bias + ReLU epilogue fused."""
    x2 = x.reshape(-1, x.shape[-1])
    y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False)
    return y.view(*x.shape[:-1], weight.shape[0])


def absmax_partials(x, out=None):
    """ff3d_absmax_partials_f32: 256 partial maxima of |x| (fp32 CUDA tensor, dense storage) -> ``out`` (256,) fp32."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise RuntimeError('absmax_partials: expected a contiguous CUDA fp32 tensor')
    if out is None:
        out = torch.empty(256, device=x.device)
    _lib.check(lib.ff3d_absmax_partials_f32(C.c_void_p(x.data_ptr()), x.numel(), _chk(out, name='out'), _stream()),
               'ff3d_absmax_partials_f32')
    return out


def linear_wgrad_ok(x2, dy2):
    """Shapes / layouts ff3d_linear_wgrad_f16x3 takes: (M, K) and (M, N) fp32 CUDA rows with unit inner stride, K, N and the row
    strides multiples of 4 floats, 16-byte aligned bases."""
    return (x2.is_cuda and dy2.is_cuda and x2.dtype == torch.float32 and dy2.dtype == torch.float32 and x2.dim() == 2
            and dy2.dim() == 2 and x2.shape[0] == dy2.shape[0] and x2.shape[0] > 0 and x2.stride(1) == 1 and dy2.stride(1) == 1
            and x2.shape[1] % 4 == 0 and dy2.shape[1] % 4 == 0 and x2.shape[1] >= 4 and dy2.shape[1] >= 4
            and x2.stride(0) % 4 == 0 and dy2.stride(0) % 4 == 0 and x2.stride(0) >= x2.shape[1]
            and dy2.stride(0) >= dy2.shape[1] and x2.data_ptr() % 16 == 0 and dy2.data_ptr() % 16 == 0)


def rows_absmax(t2):
    """The 256 partial maxima of a (M, n) fp32 row matrix for linear_wgrad (a column block of a wider matrix is measured over its
    enclosing rows: an upper bound, which is all the scaling needs)."""
    if t2.is_contiguous():
        return absmax_partials(t2)
    return absmax_partials(t2.as_strided(((t2.shape[0] - 1) * t2.stride(0) + t2.shape[1],), (1,)))


def linear_wgrad(x2, dy2, want_bias=True, amax_x=None):
    """Weight (and bias) gradient of ``y = x W^T + b`` on the fp16 matrix cores with fp32-class accuracy (ff3d_linear_wgrad_f16x3,
    csrc/wgrad.hip): x2 (M, K), dy2 (M, N) fp32 rows -> (dW (N, K), db (N) | None).  ``amax_x``: rows_absmax(x2) when the caller
    already has it (the layers that share an input - the six value_proj of the decoder read one flattened pyramid - measure it
    once, in the forward pass)."""
    lib = _lib.load()
    if not linear_wgrad_ok(x2, dy2):
        raise RuntimeError('linear_wgrad: unsupported operand layout (see linear_wgrad_ok)')
    M, K = x2.shape
    N = dy2.shape[1]
    dev = x2.device
    if amax_x is None:
        amax_x = rows_absmax(x2)
    amax_y = rows_absmax(dy2)
    S = lib.ff3d_linear_wgrad_slices(M, K, N)
    ws = torch.empty(S * (N * K + N), device=dev)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev) if want_bias else None
    st = lib.ff3d_linear_wgrad_f16x3(C.c_void_p(x2.data_ptr()), x2.stride(0), C.c_void_p(dy2.data_ptr()), dy2.stride(0),
                                     _chk(amax_x, name='amax_x'), _chk(amax_y), M, K, N, _chk(dw), _opt(db), _chk(ws), _stream())
    _lib.check(st, 'ff3d_linear_wgrad_f16x3')
    return dw, db


def _lin_weight_args(w_split):
    """ctypes arguments of a split linear weight, validated once per Pair object: (w_hi, w_lo, w_exp, N, K)."""
    args = getattr(w_split, '_lin_args', None)
    if args is None:
        wh, wl = w_split
        exp = as_pair(w_split).exp
        args = (_plane(wh, 'w_hi'), _plane(wl, 'w_lo'), _opt(exp, torch.int32, 'w_exp'), wh.shape[0], wh.shape[1])
        if isinstance(w_split, Pair):
            w_split._lin_args = args
    return args


def _lin_rows(x, K, name):
    x2 = x if x.dim() == 2 else x.reshape(-1, K)
    if not (x2.is_cuda and x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.shape[1] == K):
        raise RuntimeError(f'{name}: expected a CUDA fp32 (M, K) operand with unit inner stride')
    return x2


def linear_f16x3(x, w_split, bias=None, relu=False, x2=None, n_split=0):
    """act(x @ W^T + bias) for the query-side projections of the decoder on the fp16 matrix cores with fp32-class accuracy
    (ff3d_linear_f16x3, csrc/linear.hip): x (..., K) fp32 with unit inner stride (rows at any 4-float-aligned stride), W =
    split_weight_f16(weight) (N, K), K % 32 == 0 -> (..., N) fp32.  The activation is normalised per row and split inside the
    kernel: no exponent plumbing, any fp32 magnitude.  With ``x2`` (same shape and row stride) the output columns from
    ``n_split`` (a multiple of 128) on are x2 @ W[n_split:]^T (ff3d_linear_dual_f16x3: q | k | v of nn.MultiheadAttention from
    x + pos and x in one launch)."""
    lib = _lib.load()
    wh_p, wl_p, exp_p, N, K = _lin_weight_args(w_split)
    a = _lin_rows(x, K, 'linear_f16x3')
    M = a.shape[0]
    out = torch.empty(M, N, device=x.device)
    ev = _dense_event_start()
    if x2 is None:
        st = lib.ff3d_linear_f16x3(C.c_void_p(a.data_ptr()), a.stride(0), wh_p, wl_p, exp_p, _opt(bias, name='bias'), int(relu),
                                   C.c_void_p(out.data_ptr()), N, M, N, K, _stream())
    else:
        b = _lin_rows(x2, K, 'linear_f16x3 (x2)')
        if b.shape != a.shape or b.stride(0) != a.stride(0):
            raise RuntimeError('linear_f16x3: x2 must have the shape and row stride of x')
        st = lib.ff3d_linear_dual_f16x3(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), int(n_split), a.stride(0), wh_p,
                                        wl_p, exp_p, _opt(bias, name='bias'), int(relu), C.c_void_p(out.data_ptr()), N, M, N, K,
                                        _stream())
    _dense_event_end(ev, f'linear {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_linear_f16x3')
    return out.view(*x.shape[:-1], N)


def linear_add_ln_f16x3(x, w_split, bias, residual, gamma, beta, eps=1e-5, pos=None):
    """LayerNorm(residual + x @ W^T + bias) in one launch (ff3d_linear_add_ln_f16x3: the projection of linear_f16x3 with the
    decoder layer's residual + post-norm as its epilogue; N = 256); with ``pos`` also returns the normalised rows + pos."""
    lib = _lib.load()
    wh_p, wl_p, exp_p, N, K = _lin_weight_args(w_split)
    a = _lin_rows(x, K, 'linear_add_ln_f16x3')
    M = a.shape[0]
    if residual.numel() != M * N:
        raise RuntimeError('linear_add_ln_f16x3: residual must be (M, N)')
    out = torch.empty_like(residual)
    out_pos = torch.empty_like(residual) if pos is not None else None
    ev = _dense_event_start()
    st = lib.ff3d_linear_add_ln_f16x3(C.c_void_p(a.data_ptr()), a.stride(0), wh_p, wl_p, exp_p, _opt(bias, name='bias'),
                                      _chk(residual, name='residual'), _chk(gamma, name='gamma'), _chk(beta, name='beta'),
                                      float(eps), _opt(pos, name='pos'), C.c_void_p(out.data_ptr()), _opt(out_pos), M, N, K,
                                      _stream())
    _dense_event_end(ev, f'linear+ln {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_linear_add_ln_f16x3')
    return (out, out_pos) if pos is not None else out


def linear_kslices_f16x3(x, w_slices, kslices, n):
    """K-sliced projection of a few rows (ff3d_linear_kslices_f16x3): x (..., kslices * K) fp32, ``w_slices`` = split_weight_f16 of the
    layer's (n, kslices * K) weight re-laid as (kslices * n, K) (slice s of the K axis = rows s * n .. s * n + n - 1; kslice_weight) ->
    (M, kslices * n) fp32 partial columns for sum_add_layer_norm."""
    lib = _lib.load()
    wh_p, wl_p, exp_p, N_, K = _lin_weight_args(w_slices)
    if N_ != kslices * n:
        raise RuntimeError('linear_kslices_f16x3: the weight planes must hold kslices * n rows')
    a = _lin_rows(x, kslices * K, 'linear_kslices_f16x3')
    M = a.shape[0]
    out = torch.empty(M, N_, device=x.device)
    ev = _dense_event_start()
    st = lib.ff3d_linear_kslices_f16x3(C.c_void_p(a.data_ptr()), a.stride(0), int(kslices), wh_p, wl_p, exp_p, C.c_void_p(out.data_ptr()),
                                       N_, M, int(n), K, _stream())
    _dense_event_end(ev, f'linear {M}x{K}x{n} x{kslices} K slices', 2.0 * M * N_ * K)
    _lib.check(st, 'ff3d_linear_kslices_f16x3')
    return out


def kslice_weight(weight, kslices):
    """(n, kslices * K) weight -> split_weight_f16 of its K slices stacked along the rows ((kslices * n, K)); once per weight load."""
    n, kk = weight.shape
    k = kk // kslices
    return split_weight_f16(weight.detach().reshape(n, kslices, k).permute(1, 0, 2).reshape(kslices * n, k).contiguous())


def sum_add_layer_norm(parts, nparts, bias, residual, gamma, beta, eps=1e-5, pos=None):
    """LayerNorm(residual + bias + the sum of the ``nparts`` column blocks of parts (M, nparts * C)) (ff3d_sum_add_layer_norm); with
    ``pos`` also returns the normalised rows + pos."""
    lib = _lib.load()
    C_ = residual.shape[-1]
    rows = residual.numel() // C_
    if parts.numel() != rows * nparts * C_ or not parts.is_contiguous():
        raise RuntimeError('sum_add_layer_norm: parts must be contiguous (rows, nparts * C)')
    out = torch.empty_like(residual)
    out_pos = torch.empty_like(residual) if pos is not None else None
    st = lib.ff3d_sum_add_layer_norm(_chk(parts, name='parts'), int(nparts), nparts * C_, _opt(bias, name='bias'),
                                     _chk(residual, name='residual'), _chk(gamma, name='gamma'), _chk(beta, name='beta'),
                                     _opt(pos, name='pos'), _chk(out), _opt(out_pos), rows, C_, float(eps), _stream())
    _lib.check(st, 'ff3d_sum_add_layer_norm')
    return (out, out_pos) if pos is not None else out


class Bf16Weight(tuple):
    """A bf16 linear weight for ff3d_linear_rows' one-plane mode: ``(plane,)`` = the (N, K) bf16 view of an (N + 1, K) buffer whose
    last row is zero (the kernels' padding source, ff3d.h ZERO-ROW CONTRACT), plus ``bias`` = the bias rounded to bf16 and widened
    back to fp32 (the oracle's ``lin(lowp=True)`` rounds it, the kernel adds it in fp32)."""

    def __new__(cls, plane, bias=None):
        self = super().__new__(cls, (plane,))
        self.bias = bias
        return self

    def __getnewargs__(self):
        return (self[0], self.bias)

    def __deepcopy__(self, memo):               # (without the cached ctypes argument tuple, see Pair.__deepcopy__)
        import copy
        return Bf16Weight(copy.deepcopy(self[0], memo), copy.deepcopy(self.bias, memo))


def bf16_weight(w, bias=None):
    """Once per weight load (cached by the caller): (N, K) fp32 -> Bf16Weight (round-to-nearest-even, zero row appended)."""
    w = w.detach().float().contiguous()
    N, K = w.shape
    buf = torch.zeros(N + 1, K, dtype=torch.bfloat16, device=w.device)
    buf[:N] = w.to(torch.bfloat16)
    b = None if bias is None else bias.detach().to(torch.bfloat16).float().contiguous()
    return Bf16Weight(buf[:N], b)


def _rows_weight_args(w):
    """ctypes arguments of a weight for ff3d_linear_rows, validated once per object: (w_hi, w_lo | NULL, w_exp | NULL, N, K)."""
    args = getattr(w, '_rows_args', None)
    if args is None:
        if isinstance(w, Bf16Weight):
            t = w[0]
            if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()
                    and t.untyped_storage().nbytes() >= (t.storage_offset() + t.numel() + t.shape[-1]) * 2):
                raise RuntimeError('linear_rows: expected a contiguous CUDA bf16 plane followed by its zero row (ops.bf16_weight)')
            args = (C.c_void_p(t.data_ptr()), C.c_void_p(0), C.c_void_p(0), t.shape[0], t.shape[1])
        else:
            args = _lin_weight_args(w)
        try:
            w._rows_args = args
        except AttributeError:
            pass
    return args


def linear_rows(x, w, bias=None, relu=False, x2=None, n_split=0, residual=None, gamma=None, beta=None, eps=1e-5, pos=None):
    """ff3d_linear_rows (csrc/linrows.hip): act(x @ W^T + bias) on the row-owning kernel; ``w`` = split_weight_f16(weight) (fp32-class,
    three fp16 MFMA passes) or bf16_weight(weight, bias) (bf16 operands, fp32 accumulate, result rounded to bf16 and returned as fp32:
    BASELINE configs[4] mode; ``bias`` then defaults to the weight's rounded bias).  x (..., K) fp32 with unit inner stride, K % 32 == 0.
    ``x2`` / ``n_split`` (a multiple of 256): output columns from n_split on are x2 @ W[n_split:]^T.  ``residual`` (M, 256) +
    ``gamma`` / ``beta``: LayerNorm(residual + .) over N = 256 columns, and with ``pos`` also (normalised rows + pos) as a second
    result."""
    lib = _lib.load()
    wh_p, wl_p, exp_p, N, K = _rows_weight_args(w)
    if isinstance(w, Bf16Weight) and bias is None:
        bias = w.bias
    a = _lin_rows(x, K, 'linear_rows')
    M = a.shape[0]
    z = C.c_void_p(0)
    a2p = z
    if x2 is not None:
        b = _lin_rows(x2, K, 'linear_rows (x2)')
        if b.shape != a.shape or b.stride(0) != a.stride(0):
            raise RuntimeError('linear_rows: x2 must have the shape and row stride of x')
        a2p = C.c_void_p(b.data_ptr())
    ev = _dense_event_start()
    if residual is not None:
        if residual.numel() != M * N:
            raise RuntimeError('linear_rows: residual must be (M, N)')
        out = torch.empty_like(residual)
        out_pos = torch.empty_like(residual) if pos is not None else None
        st = lib.ff3d_linear_rows(C.c_void_p(a.data_ptr()), z, 0, a.stride(0), wh_p, wl_p, exp_p, _opt(bias, name='bias'), 0,
                                  _chk(residual, name='residual'), _chk(gamma, name='gamma'), _chk(beta, name='beta'), float(eps),
                                  _opt(pos, name='pos'), C.c_void_p(out.data_ptr()), _opt(out_pos), N, M, N, K, _stream())
        _dense_event_end(ev, f'linear+ln rows {M}x{K}x{N}', 2.0 * M * N * K)
        _lib.check(st, 'ff3d_linear_rows')
        return (out, out_pos) if pos is not None else out
    out = torch.empty(M, N, device=x.device)
    st = lib.ff3d_linear_rows(C.c_void_p(a.data_ptr()), a2p, int(n_split), a.stride(0), wh_p, wl_p, exp_p, _opt(bias, name='bias'),
                              int(relu), z, z, z, 0.0, z, C.c_void_p(out.data_ptr()), z, N, M, N, K, _stream())
    _dense_event_end(ev, f'linear rows {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_linear_rows')
    return out.view(*x.shape[:-1], N)


class TiledWeight:
    """K-step-tiled split planes of a linear weight for ff3d_ffn_rows: ``hi`` / ``lo`` (K / 32, N, 32) fp16 contiguous - tile ks row n =
    W[n, 32 ks : 32 ks + 32] * 2^-exp as (hi, lo) with the low part UNSCALED (hi + lo, not hi + lo'/2048) - plus the device exponent
    of the split."""

    def __init__(self, hi, lo, exp, N, K):
        self.hi, self.lo, self.exp, self.N, self.K = hi, lo, exp, N, K

    def __deepcopy__(self, memo):
        import copy
        return TiledWeight(copy.deepcopy(self.hi, memo), copy.deepcopy(self.lo, memo), copy.deepcopy(self.exp, memo), self.N, self.K)


def tile_weight_f16(weight, bias=None):
    """split_weight_f16(weight) re-laid in K-step tiles (once per weight load; cached by the caller)."""
    sp = split_weight_f16(weight, bias=bias)
    N, K = sp[0].shape
    if K % 32:
        raise RuntimeError('tile_weight_f16: K % 32 != 0')
    tile = lambda t: t.reshape(N, K // 32, 32).permute(1, 0, 2).contiguous()
    # the low plane UNSCALED (lo = lo' / 2048, an exact power-of-two step): ffnrows.hip adds the three passes in one accumulator
    return TiledWeight(tile(sp[0]), tile((sp[1].float() * (1.0 / 2048.0)).half()), sp.exp, N, K)


def ffn_rows(x, w1t, b1, w2t, b2, residual, gamma, beta, eps=1e-5, pos=None):
    """ff3d_ffn_rows (csrc/ffnrows.hip): LayerNorm(residual + relu(x @ W1^T + b1) @ W2^T + b2) (+ pos as a second result) in one
    launch, fp32-class.  x (..., 256) fp32 with unit inner stride; ``w1t`` / ``w2t`` = tile_weight_f16 of the two fcs' weights
    ((hidden, 256) and (256, hidden), hidden % 128 == 0); residual / pos contiguous (M, 256)."""
    lib = _lib.load()
    if not (isinstance(w1t, TiledWeight) and isinstance(w2t, TiledWeight) and w1t.K == 256 and w2t.N == 256 and w1t.N == w2t.K
            and w1t.N % 128 == 0):
        raise RuntimeError('ffn_rows: expected tile_weight_f16 planes of a (hidden, 256) and a (256, hidden) weight, hidden % 128 == 0')
    a = _lin_rows(x, 256, 'ffn_rows')
    M = a.shape[0]
    if residual.numel() != M * 256 or not residual.is_contiguous():
        raise RuntimeError('ffn_rows: residual must be contiguous (M, 256)')
    out = torch.empty_like(residual)
    out_pos = torch.empty_like(residual) if pos is not None else None
    ev = _dense_event_start()
    st = lib.ff3d_ffn_rows(C.c_void_p(a.data_ptr()), a.stride(0), _chk(w1t.hi, torch.float16), _chk(w1t.lo, torch.float16),
                           _opt(w1t.exp, torch.int32, 'w1_exp'), _chk(b1, name='b1'), w1t.N, _chk(w2t.hi, torch.float16),
                           _chk(w2t.lo, torch.float16), _opt(w2t.exp, torch.int32, 'w2_exp'), _opt(b2, name='b2'),
                           _chk(residual, name='residual'), _chk(gamma, name='gamma'), _chk(beta, name='beta'), float(eps),
                           _opt(pos, name='pos'), C.c_void_p(out.data_ptr()), _opt(out_pos), M, _stream())
    _dense_event_end(ev, f'ffn+ln rows {M}x256x{w1t.N}', 4.0 * M * 256 * w1t.N)
    _lib.check(st, 'ff3d_ffn_rows')
    return (out, out_pos) if pos is not None else out


def _bf16_plane(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.dim() == 2
            and t.untyped_storage().nbytes() >= (t.storage_offset() + t.numel() + t.shape[-1]) * 2):
        raise RuntimeError(f'{name}: expected a contiguous CUDA bf16 (rows, K) plane followed by its zero row')
    return C.c_void_p(t.data_ptr())


def gemm_bf16(a, w, relu=False, out_bf16=False, ksplit=None):
    """ff3d_gemm_bf16: act(A (M, K) bf16 @ W^T + bias) with the bf16 arithmetic of BASELINE configs[4] (exact products, fp32
    accumulation, one rounding to bf16).  ``a``: a bf16 plane followed by its zero row (bev_flatten_multi(bf16=True),
    roi_grid_sample(out_dtype=torch.bfloat16)); ``w``: bf16_weight(weight, bias).  -> (M, N) fp32 holding bf16 values, or with
    ``out_bf16`` a bf16 tensor (the projected value the deformable gather reads)."""
    lib = _lib.load()
    M, K = a.shape
    N = w[0].shape[0]
    ks = 1 if out_bf16 else (gemm_ksplit(M, N, K) if ksplit is None else int(ksplit))
    out = torch.empty(M, N, device=a.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    ws = torch.empty(ks, M, N, device=a.device) if ks > 1 else None
    z = C.c_void_p(0)
    ev = _dense_event_start()
    st = lib.ff3d_gemm_bf16(_bf16_plane(a, 'a'), _bf16_plane(w[0], 'w'), _opt(w.bias, name='bias'), int(relu),
                            z if out_bf16 else _chk(out), C.c_void_p(out.data_ptr()) if out_bf16 else z, M, N, K, ks, _opt(ws), _stream())
    _dense_event_end(ev, f'gemm bf16 {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_gemm_bf16')
    return out


def relu_conv3x3_small(x, in_bias, weight, bias, relu=True):
    """conv3x3(relu(x + in_bias)) + bias for K <= 16 output channels (heatmap_head tail, FD:204-220), one launch."""
    lib = _lib.load()
    B, C_, H, W = x.shape
    K = weight.shape[0]
    out = torch.empty(B, K, H, W, device=x.device)
    st = lib.ff3d_relu_conv3x3_small(_chk(x, name='x'), _opt(in_bias, name='in_bias'), 1 if relu else 0,
                                     _chk(weight, name='weight'), _opt(bias, name='bias'), _chk(out), B, C_, H, W, K, _stream())
    _lib.check(st, 'ff3d_relu_conv3x3_small')
    return out


def heatmap_nms(logits, mask_in=None, logits_b=None, nms_kernel=3, small_bits=0, want_mask_next=True):
    """FD:631-634/662-666 + FD:672-685 (and FD:549 with ``logits_b``).  Returns (heat, hist, mask_next)."""
    lib = _lib.load()
    B, K, H, W = logits.shape
    heat = torch.empty_like(logits)
    hist = torch.empty(B, HIST_BINS, device=logits.device, dtype=torch.int32)
    mask_next = torch.empty_like(logits) if want_mask_next else None
    st = lib.ff3d_heatmap_nms(_chk(logits, name='logits'), _opt(logits_b, name='logits_b'), _opt(mask_in, name='mask_in'),
                              _opt(mask_next, name='mask_next'), _chk(heat), _chk(hist, torch.int32), B, K, H, W,
                              nms_kernel, small_bits, _stream())
    _lib.check(st, 'ff3d_heatmap_nms')
    return heat, hist, mask_next


def topk(heat, hist, k, workspace=None):
    """FD:688 / FD:574, deterministic (score desc, ties by lowest index).  heat (B,...) -> idx (B,k) int64."""
    lib = _lib.load()
    B = heat.shape[0]
    n = heat[0].numel()
    need = lib.ff3d_topk_workspace_bytes(B, n)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, device=heat.device, dtype=torch.uint8)
    idx = torch.empty(B, k, device=heat.device, dtype=torch.int64)
    st = lib.ff3d_topk(_chk(heat, name='heat'), _chk(hist, torch.int32, 'hist'), _chk(idx, torch.int64),
                       _chk(workspace, workspace.dtype), B, n, k, _stream())
    _lib.check(st, 'ff3d_topk')
    return idx


def query_gather(feat, heat, idx, cls_w, cls_b, qfeat, qpos, qscore, qlabel, mask, q_offset, mask_mode, nms_kernel,
                 small_bits):
    """FD:690-706 + FD:725-782.  qfeat is a (B,Nq,C) view with arbitrary strides; the other outputs
    are contiguous (B,Nq,2), (B,K,Nq), (B,Nq) int64; ``mask`` (B,K,H,W) is cleared in place."""
    lib = _lib.load()
    B, C_, H, W = feat.shape
    K = heat.shape[1]
    k = idx.shape[1]
    Nq = qpos.shape[1]
    assert qfeat.shape == (B, Nq, C_) and qfeat.is_cuda and qfeat.dtype == torch.float32
    st = lib.ff3d_query_gather(_chk(feat, name='feat'), _chk(heat, name='heat'), _chk(idx, torch.int64, 'idx'),
                               _chk(cls_w, name='cls_w'), _chk(cls_b, name='cls_b'), C.c_void_p(qfeat.data_ptr()),
                               qfeat.stride(0), qfeat.stride(1), qfeat.stride(2), _chk(qpos), _chk(qscore),
                               _chk(qlabel, torch.int64), _opt(mask), B, C_, K, H, W, k, q_offset, Nq,
                               mask_mode if mask is not None else 0, nms_kernel, small_bits, _stream())
    _lib.check(st, 'ff3d_query_gather')


# class -> task group of the heatmap_box branch (FD:232-239: car | truck, construction_vehicle | bus, trailer | barrier | motorcycle,
# bicycle | pedestrian, traffic_cone)
HEATMAP_TASK_OF_CLASS = (0, 1, 1, 2, 2, 3, 4, 4, 5, 5)


def heatmap_box_gather(raw, idx, query_box, q_offset, num_classes, class_task=HEATMAP_TASK_OF_CLASS):
    """FD:606-629 / 641-660 + FD:708-722: raw (B, T*10, H, W) task-head output, idx (B, k) -> query_box[:, :, q_offset:q_offset+k]
    (query_box (B, 10, Nq), written in place)."""
    lib = _lib.load()
    B, TC, H, W = raw.shape
    k = idx.shape[1]
    assert TC % 10 == 0 and query_box.shape[:2] == (B, 10) and len(class_task) >= num_classes
    ct = (C.c_int32 * num_classes)(*class_task[:num_classes])
    st = lib.ff3d_heatmap_box_gather(_chk(raw, name='raw'), _chk(idx, torch.int64, 'idx'), ct, _chk(query_box, name='query_box'),
                                     B, num_classes, TC // 10, H, W, k, q_offset, query_box.shape[2], _stream())
    _lib.check(st, 'ff3d_heatmap_box_gather')
    return query_box


def box_class_mask(query_box, qlabel, mask, k, q_offset, coder, center_range, nms_kernel, small_bits, margin=1.0, min_bev_dim=0.7,
                   max_bev_dim=10.0):
    """FD:732-768 + FD:774-782, the box part of mask_heatmap_mode='boxcls': clears, in ``mask`` (B,K,H,W), the (dilated) cells whose
    centre lies inside a box of the stage's queries [q_offset, q_offset + k), in the query's class plane."""
    lib = _lib.load()
    B, K, H, W = mask.shape
    st = lib.ff3d_box_class_mask(_chk(query_box, name='query_box'), _chk(qlabel, torch.int64, 'qlabel'), _chk(mask, name='mask'),
                                 B, K, H, W, k, q_offset, query_box.shape[2], _floats(coder), _floats(center_range), float(margin),
                                 float(min_bev_dim), float(max_bev_dim), nms_kernel, small_bits, _stream())
    _lib.check(st, 'ff3d_box_class_mask')
    return mask


def bev_flatten(levels, pos_embed=None, want_raw=True, want_value=True, value_split=False, level_exps=None, pe_exp=None):
    """FD:823 (+ FD:886).  levels: list of (B,C,H_l,W_l) -> (raw (B,Nv,C) | None, value (B,Nv,C) | None).
    value_split: the value comes back as the (hi, lo') fp16 Pair consumed by gemm_f16x3 instead of fp32.  ``level_exps``
    (one device int32 bound exponent per level, from the op that produced / split the level) and ``pe_exp`` range-normalise
    the pair (ff3d.h); the raw tensor's bound exponent is attached to it as ``raw._ff3d_exp`` for roi_grid_sample."""
    lib = _lib.load()
    B, C_ = levels[0].shape[:2]
    level_hw = [tuple(f.shape[2:]) for f in levels]
    Nv = sum(h * w for h, w in level_hw)
    ptrs = (C.c_void_p * len(levels))(*[_chk(f, name='level').value for f in levels])
    raw = torch.empty(B, Nv, C_, device=levels[0].device) if want_raw else None
    lv, L = _levels(level_hw)
    z = C.c_void_p(0)
    if want_value and value_split:
        pair = _split_planes(B * Nv, C_, levels[0].device)
        exps = None
        vexp = rexp = None
        if level_exps is not None:
            exps = (C.c_void_p * L)(*[_chk(e, torch.int32, 'level_exp').value for e in level_exps])
            vexp, rexp = _new_exp(levels[0].device), _new_exp(levels[0].device)
        st = lib.ff3d_bev_flatten(ptrs, _opt(pos_embed, name='pos_embed'), _opt(raw), _chk(pair, torch.float16), 2, B, C_, L,
                                  lv, exps if exps is not None else z, _opt(pe_exp, torch.int32, 'pe_exp'),
                                  _opt(vexp, torch.int32), _opt(rexp, torch.int32), _stream())
        _lib.check(st, 'ff3d_bev_flatten')
        if raw is not None and rexp is not None:
            raw._ff3d_exp = rexp
        return raw, Pair(pair[0, :-1].view(B, Nv, C_), pair[1, :-1].view(B, Nv, C_), vexp)
    val = torch.empty(B, Nv, C_, device=levels[0].device) if want_value else None
    st = lib.ff3d_bev_flatten(ptrs, _opt(pos_embed, name='pos_embed'), _opt(raw), _opt(val), 0, B, C_, L, lv, z, z, z, z,
                              _stream())
    _lib.check(st, 'ff3d_bev_flatten')
    return raw, val


def bev_flatten_multi(levels, pos_embeds, want_raw, level_exps=None, pe_exps=None, bf16=False):
    """One pyramid pass, several value tensors: value_s = pyramid + pos_embeds[s] (one per decoder stage, FD:886) plus the raw
    (B, Nv, C) pyramid when ``want_raw``.  With ``level_exps`` / ``pe_exps`` the values are range-normalised (hi, lo') Pairs
    (the split-fp16 value GEMM's operand) -> (raw | None, [Pair, ...]); without them plain fp32 (B, Nv, C) tensors (the vendor
    value projection) -> (raw | None, [tensor, ...]); ``bf16``: bf16 (B, Nv, C) planes (zero row behind them) for gemm_bf16."""
    lib = _lib.load()
    B, C_ = levels[0].shape[:2]
    level_hw = [tuple(f.shape[2:]) for f in levels]
    Nv = sum(h * w for h, w in level_hw)
    dev = levels[0].device
    n = len(pos_embeds)
    ptrs = (C.c_void_p * len(levels))(*[_chk(f, name='level').value for f in levels])
    raw = torch.empty(B, Nv, C_, device=dev) if want_raw else None
    lv, L = _levels(level_hw)
    if bf16:
        # BASELINE configs[4] mode (round 5): every value tensor as ONE bf16 plane (round to nearest even) followed by its zero row -
        # the operand of gemm_bf16 (rounds 1-4: fp32 values, cast launches, the vendor's bf16 GEMM)
        bufs = [torch.empty(B * Nv + 1, C_, dtype=torch.bfloat16, device=dev) for _ in range(n)]
        for b_ in bufs:
            b_[B * Nv].zero_()
        arr_ = lambda items: (C.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in items])     # noqa: E731
        for pe_ in pos_embeds:
            _chk(pe_, name='pos_embed')
        st = lib.ff3d_bev_flatten_multi(ptrs, n, arr_(pos_embeds), _opt(raw), arr_(bufs), 1, B, C_, L, lv, None, None, None, None,
                                        _stream())
        _lib.check(st, 'ff3d_bev_flatten_multi')
        return raw, [b_[:B * Nv].view(B, Nv, C_) for b_ in bufs]
    if level_exps is None:
        outs = [torch.empty(B, Nv, C_, device=dev) for _ in range(n)]
        arr_ = lambda items: (C.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in items])     # noqa: E731
        for pe_ in pos_embeds:
            _chk(pe_, name='pos_embed')
        st = lib.ff3d_bev_flatten_multi(ptrs, n, arr_(pos_embeds), _opt(raw), arr_(outs), 0, B, C_, L, lv, None, None, None, None,
                                        _stream())
        _lib.check(st, 'ff3d_bev_flatten_multi')
        return raw, outs
    bufs = [_split_planes(B * Nv, C_, dev) for _ in range(n)]
    vexps = [_new_exp(dev) for _ in range(n)]
    rexp = _new_exp(dev)
    arr = lambda items: (C.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in items])          # noqa: E731
    exps = (C.c_void_p * L)(*[_chk(e, torch.int32, 'level_exp').value for e in level_exps])
    st = lib.ff3d_bev_flatten_multi(ptrs, n, arr(pos_embeds), _opt(raw), arr(bufs), 2, B, C_, L, lv, exps, arr(pe_exps),
                                    arr(vexps), _chk(rexp, torch.int32), _stream())
    _lib.check(st, 'ff3d_bev_flatten_multi')
    if raw is not None:
        raw._ff3d_exp = rexp
    return raw, [Pair(b[0, :-1].view(B, Nv, C_), b[1, :-1].view(B, Nv, C_), e) for b, e in zip(bufs, vexps)]


def sine_embed(pos, dim_t, W, H):
    """UT:40-53 with the FD:869/883 normalisation fused.  pos (...,2) -> (...,256)."""
    lib = _lib.load()
    N = pos.numel() // 2
    emb = torch.empty(*pos.shape[:-1], 256, device=pos.device)
    st = lib.ff3d_sine_embed(_chk(pos, name='pos'), _chk(dim_t, name='dim_t'), _chk(emb), N, float(W), float(H), _stream())
    _lib.check(st, 'ff3d_sine_embed')
    return emb


def roi_grid_sample(feat_cl, level_hw, query_box, g, expand, coder, roi_range, layout=0, want_grid=False,
                    out_dtype=torch.float32, feat_exp=None):
    """FD:891-919.  feat_cl (B,Nv,C), query_box (B,box_dim,Nq) -> (B*Nq, L*C*g*g)[, grid (B,Nq,g*g,2)].
    coder = (out_size_factor, voxel_x, voxel_y, pc_x, pc_y); roi_range = (x0, y0, x1, y1).  out_dtype 'f16split': the
    (hi, lo') Pair for gemm_f16x3, range-normalised with ``feat_exp`` (default: the bound exponent bev_flatten attached
    to feat_cl) - bilinear samples never exceed the map's maximum."""
    lib = _lib.load()
    B, Nv, C_ = feat_cl.shape
    box_dim, Nq = query_box.shape[1:]
    lv, L = _levels(level_hw)
    split = out_dtype == 'f16split'             # (hi, lo') fp16 pair for gemm_f16x3
    if feat_exp is None:
        feat_exp = getattr(feat_cl, '_ff3d_exp', None)
    if split:
        buf = _split_planes(B * Nq, L * C_ * g * g, feat_cl.device)
        out, dt_code, dt = Pair(buf[0, :-1], buf[1, :-1], feat_exp), 2, torch.float16
    elif out_dtype == torch.bfloat16:                 # one bf16 plane + its zero row: the operand of gemm_bf16
        buf = torch.empty(B * Nq + 1, L * C_ * g * g, device=feat_cl.device, dtype=torch.bfloat16)
        buf[B * Nq].zero_()
        out, dt_code, dt = buf[:B * Nq], 1, torch.bfloat16
    else:
        buf = out = torch.empty(B * Nq, L * C_ * g * g, device=feat_cl.device, dtype=out_dtype)
        dt_code, dt = 0, out_dtype
    grid = torch.empty(B, Nq, g * g, 2, device=feat_cl.device) if want_grid else None
    st = lib.ff3d_roi_grid_sample(_chk(feat_cl, name='feat_cl'), _chk(query_box, name='query_box'), _chk(buf, dt),
                                  dt_code, _opt(grid),
                                  B, Nq, C_, L, lv, g, box_dim, float(expand), _floats(coder), _floats(roi_range),
                                  layout, _opt(feat_exp if split else None, torch.int32, 'feat_exp'), _stream())
    _lib.check(st, 'ff3d_roi_grid_sample')
    return (out, grid) if want_grid else out


def roi_grid_sample_bwd(grad_out, feat_shape, level_hw, query_box, g, expand, coder, roi_range, layout=0):
    """Gradient of roi_grid_sample with respect to feat_cl: grad_out (B*Nq, L*C*g*g) -> (B, Nv, C) (training path)."""
    lib = _lib.load()
    B, Nv, C_ = feat_shape
    box_dim, Nq = query_box.shape[1:]
    lv, L = _levels(level_hw)
    assert grad_out.shape == (B * Nq, L * C_ * g * g)
    grad_feat = torch.zeros(B, Nv, C_, device=grad_out.device)
    st = lib.ff3d_roi_grid_sample_bwd(_chk(grad_out, name='grad_out'), _chk(query_box, name='query_box'), _chk(grad_feat),
                                      B, Nq, C_, L, lv, g, box_dim, float(expand), _floats(coder), _floats(roi_range), layout,
                                      _stream())
    _lib.check(st, 'ff3d_roi_grid_sample_bwd')
    return grad_feat


def box_decode(preds, q0, Nq, qscore, qlabel, coder, post_center_range, score_threshold=0.0, max_out=200):
    """FD:1317-1331 + BC:71-158 + FD:1395-1400.  preds: dict of (B,n,ld) tensors (heatmap, center,
    height, dim, rot[, vel]).  Returns padded (boxes (B,max_out,7|9), scores, labels int32, count int32).
    ``post_center_range=None`` = ``decode(filter=False)``: every query decoded into its own row (max_out >= Nq), nothing
    dropped or moved - non-finite boxes included (the training targets index these rows by query)."""
    lib = _lib.load()
    cls = preds['heatmap']
    B, K, ld = cls.shape
    vel = preds.get('vel')
    box_dim = 9 if vel is not None else 7
    dev = cls.device
    # one zero-filled allocation for the four padded results (one fill launch instead of four: 3 of the ~120 launches of a one-frame step)
    nb_, ns_ = B * max_out * box_dim, B * max_out
    flat = torch.zeros(nb_ + 2 * ns_ + B, device=dev)
    boxes, scores = flat[:nb_].view(B, max_out, box_dim), flat[nb_:nb_ + ns_].view(B, max_out)
    labels = flat[nb_ + ns_:nb_ + 2 * ns_].view(torch.int32).view(B, max_out)
    count = flat[nb_ + 2 * ns_:].view(torch.int32)
    st = lib.ff3d_box_decode(_chk(cls, name='heatmap'), _chk(preds['center']), _chk(preds['height']), _chk(preds['dim']),
                             _chk(preds['rot']), _opt(vel), ld, q0, _chk(qscore, name='qscore'),
                             _chk(qlabel, torch.int64, 'qlabel'), _chk(boxes), _chk(scores), _chk(labels, torch.int32),
                             _chk(count, torch.int32), B, K, Nq, max_out, _floats(coder),
                             None if post_center_range is None else _floats(post_center_range),
                             float(score_threshold or 0.0), _stream())
    _lib.check(st, 'ff3d_box_decode')
    return boxes, scores, labels, count


def box_update(raw, bias, ref, prev_box, results, q0, offsets, roi_based_reg, W, H, rows=False):
    """FD:936-957 + FD:970-987 in one launch.  raw (B,S,Nq) = fused prediction GEMM output (no bias) - or with ``rows`` the
    (B,Nq,S) output of a query-major GEMM -, ref (B,Nq,2), prev_box (B,8|10,Nq) | None, results: dict key -> (B,n,ld) tensors
    receiving this stage's slice at column q0, offsets: dict key -> first channel in raw.  Returns (qpos (B,Nq,2), query_box
    (B,8|10,Nq))."""
    lib = _lib.load()
    if rows:
        B, Nq, S = raw.shape
    else:
        B, S, Nq = raw.shape
    K = results['heatmap'].shape[1]
    vel = results.get('vel')
    nb = 10 if vel is not None else 8
    qpos = torch.empty(B, Nq, 2, device=raw.device)
    box = torch.empty(B, nb, Nq, device=raw.device)
    off = (C.c_int32 * 6)(offsets['center'], offsets['height'], offsets['dim'], offsets['rot'], offsets.get('vel', -1),
                           offsets['heatmap'])
    fn = lib.ff3d_box_update_rows if rows else lib.ff3d_box_update
    st = fn(_chk(raw, name='raw'), _chk(bias, name='bias'), _chk(ref, name='ref'), _opt(prev_box, name='prev_box'),
                             _chk(results['center']), _chk(results['height']), _chk(results['dim']), _chk(results['rot']),
                             _opt(vel), _chk(results['heatmap']), _chk(qpos), _chk(box), B, S, Nq, K,
                             results['center'].shape[2], q0, off, int(bool(roi_based_reg)), float(W), float(H), _stream())
    _lib.check(st, 'ff3d_box_update')
    return qpos, box


def pack_detections(boxes, scores, labels, count, out=None):
    """Padded detections -> the (B, M+1, 11) fp32 record all-gathered across ranks (dist.py; replaces the pickled-bytes
    gather of tools/test.py:229-233)."""
    lib = _lib.load()
    B, M, D = boxes.shape
    if out is None:
        out = torch.empty(B, M + 1, 11, device=boxes.device)
    st = lib.ff3d_pack_detections(_chk(boxes, name='boxes'), _chk(scores, name='scores'), _chk(labels, torch.int32, 'labels'),
                                  _chk(count, torch.int32, 'count'), _chk(out, name='packed'), B, M, D, _stream())
    _lib.check(st, 'ff3d_pack_detections')
    return out


def nchw_to_nhwc(x):
    """(N,C,H,W) -> (N,H,W,C) contiguous."""
    lib = _lib.load()
    N, C_, H, W = x.shape
    out = torch.empty(N, H, W, C_, device=x.device)
    st = lib.ff3d_nchw_to_nhwc(_chk(x, name='x'), _chk(out), N, C_, H * W, _stream())
    _lib.check(st, 'ff3d_nchw_to_nhwc')
    return out


def cam_sample(img_cl, lidar2img, img_aug, qk, H, W, Z, pc_range, input_hw):
    """EU:210-258 core.  img_cl (B,Ncam,Hi,Wi,Ci), lidar2img (B,Ncam,4,4), qk (B,H*W,Ci)
    -> ctx (B,H*W,Ci), valid (B,H*W) uint8."""
    lib = _lib.load()
    B, Ncam, Hi, Wi, Ci = img_cl.shape
    ctx = torch.empty(B, H * W, Ci, device=img_cl.device)
    valid = torch.empty(B, H * W, device=img_cl.device, dtype=torch.uint8)
    st = lib.ff3d_cam_sample(_chk(img_cl, name='img_cl'), _chk(lidar2img, name='lidar2img'), _opt(img_aug, name='img_aug'),
                             _chk(qk, name='qk'), _chk(ctx), _chk(valid, torch.uint8), B, Ncam, Ci, Hi, Wi, H, W, Z,
                             _floats(pc_range), _floats(input_hw), _stream())
    _lib.check(st, 'ff3d_cam_sample')
    return ctx, valid


def locatt_similar(x_ori, x_loc, kH, kW):
    """locatt_ops ``similar_forward`` (similar.cu / kernels.cuh:4-42): (B,C,H,W) x2 -> (B,H,W,kH*kW)."""
    lib = _lib.load()
    B, C_, H, W = x_ori.shape
    y = torch.empty(B, H, W, kH * kW, device=x_ori.device)
    st = lib.ff3d_locatt_similar(_chk(x_ori, name='x_ori'), _chk(x_loc, name='x_loc'), _chk(y), B, C_, H, W, kH, kW, _stream())
    _lib.check(st, 'ff3d_locatt_similar')
    return y


def locatt_weighting(x_ori, x_weight, kH, kW):
    """locatt_ops ``weighting_forward`` (weighting.cu / kernels.cuh:44-80): (B,C,H,W), (B,H,W,kH*kW) -> (B,C,H,W)."""
    lib = _lib.load()
    B, C_, H, W = x_ori.shape
    y = torch.empty_like(x_ori)
    st = lib.ff3d_locatt_weighting(_chk(x_ori, name='x_ori'), _chk(x_weight, name='x_weight'), _chk(y), B, C_, H, W, kH, kW,
                                   _stream())
    _lib.check(st, 'ff3d_locatt_weighting')
    return y


def locatt_ck2c_loc(x, weight, kH, kW):
    """locatt_ops ``ck2c_loc`` (kernels.cuh:82-119): x (B,C,H,W), weight (B,H,W,kH*kW) -> (B,C,H,W); the backward of
    ``similar`` with respect to its second operand and of ``weighting`` with respect to its first (ff3d.h)."""
    lib = _lib.load()
    B, C_, H, W = x.shape
    y = torch.empty_like(x)
    st = lib.ff3d_locatt_ck2c_loc(_chk(x, name='x'), _chk(weight, name='weight'), _chk(y), B, C_, H, W, kH, kW, _stream())
    _lib.check(st, 'ff3d_locatt_ck2c_loc')
    return y


def local_attention(query, key, value, k, scale):
    """EU:158-161 fused: weighting(value, softmax(scale * similar(query, key)))."""
    lib = _lib.load()
    B, C_, H, W = query.shape
    out = torch.empty_like(value)
    st = lib.ff3d_local_attention(_chk(query, name='query'), _chk(key, name='key'), _chk(value, name='value'), _chk(out),
                                  B, C_, H, W, k, k, float(scale), _stream())
    _lib.check(st, 'ff3d_local_attention')
    return out


def local_attention_pair(q, k, v, B, H, W, ksize, scale):
    """EU:158-161 on the fp16 matrix cores (ff3d_local_attention_pair, csrc/locatt_mfma.hip): q, k, v = NHWC ``Pair``s with planes
    (B*H*W, C) (views of (rows + 1 zero row, C) buffers, as every pair-producing kernel of this package returns them) -> the context as
    an NHWC Pair (B*H*W, C) with v's exponent."""
    lib = _lib.load()
    q, k, v = as_pair(q), as_pair(k), as_pair(v)
    M, C_ = q[0].shape
    assert M == B * H * W and k[0].shape == (M, C_) and v[0].shape == (M, C_)
    for t, name in ((q, 'q'), (k, 'k'), (v, 'v')):
        _plane(t[0], name + '_hi'), _plane(t[1], name + '_lo')
    dev = q[0].device
    ws = torch.empty(int(lib.ff3d_local_attention_pair_workspace_halfs(B, C_, H, W)), dtype=torch.float16, device=dev)
    buf = _split_planes(M, C_, dev)
    exp = lambda t: C.c_void_p(0 if t.exp is None else t.exp.data_ptr())                  # noqa: E731
    st = lib.ff3d_local_attention_pair(C.c_void_p(q[0].data_ptr()), C.c_void_p(q[1].data_ptr()), exp(q), C.c_void_p(k[0].data_ptr()),
                                       C.c_void_p(k[1].data_ptr()), exp(k), C.c_void_p(v[0].data_ptr()), C.c_void_p(v[1].data_ptr()),
                                       C.c_void_p(ws.data_ptr()), C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr()),
                                       B, C_, H, W, int(ksize), float(scale), _stream())
    _lib.check(st, 'ff3d_local_attention_pair')
    return Pair(buf[0, :-1], buf[1, :-1], v.exp)


def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, B, D, H, W):
    """``bev_pool_ext.bev_pool_forward`` (bev_pool.cpp:21-53): x (n,c) sorted by rank, geom_feats (n,4) int32,
    interval_* (n_intervals) int32 -> (B, D, H, W, c)."""
    lib = _lib.load()
    n, c = x.shape
    out = torch.zeros(B, D, H, W, c, device=x.device)
    st = lib.ff3d_bev_pool(_chk(x, name='x'), _chk(geom_feats, torch.int32, 'geom_feats'),
                           _chk(interval_starts, torch.int32, 'interval_starts'),
                           _chk(interval_lengths, torch.int32, 'interval_lengths'), _chk(out), B, D, H, W, n, c,
                           interval_starts.numel(), _stream())
    _lib.check(st, 'ff3d_bev_pool')
    return out


def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, B, D, H, W):
    """``bev_pool_ext.bev_pool_backward`` (bev_pool.cpp:55-88): out_grad (B, D, H, W, c) -> x_grad (n, c)."""
    lib = _lib.load()
    n, c = geom_feats.shape[0], out_grad.shape[-1]
    x_grad = torch.empty(n, c, device=out_grad.device)
    st = lib.ff3d_bev_pool_bwd(_chk(out_grad, name='out_grad'), _chk(geom_feats, torch.int32, 'geom_feats'),
                               _chk(interval_starts, torch.int32, 'interval_starts'),
                               _chk(interval_lengths, torch.int32, 'interval_lengths'), _chk(x_grad), B, D, H, W, n, c,
                               interval_starts.numel(), _stream())
    _lib.check(st, 'ff3d_bev_pool_bwd')
    return x_grad


def bev_pool(feats, coords, B, D, H, W):
    """``bev_pool`` of ops/bev_pool/bev_pool_op.py:81-97 (+ QuickCumsumCuda.forward :37-56): rank, sort, intervals
    (framework index ops, as in the reference) then the pooling kernel; returns (B, c, D, H, W)."""
    assert feats.shape[0] == coords.shape[0]
    ranks = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    indices = ranks.argsort()
    feats, coords, ranks = feats[indices], coords[indices], ranks[indices]
    kept = torch.ones(feats.shape[0], device=feats.device, dtype=torch.bool)
    kept[1:] = ranks[1:] != ranks[:-1]
    interval_starts = torch.where(kept)[0].int()
    interval_lengths = torch.zeros_like(interval_starts)
    interval_lengths[:-1] = interval_starts[1:] - interval_starts[:-1]
    interval_lengths[-1] = feats.shape[0] - interval_starts[-1]
    out = bev_pool_forward(feats.contiguous(), coords.int().contiguous(), interval_lengths, interval_starts, B, D, H, W)
    return out.permute(0, 4, 1, 2, 3).contiguous()


def circle_nms(boxes, scores, labels, count, num_classes, class_task, task_radius, max_out=200, post_max_size=83):
    """FD:1352-1393 for nms_type='circle' on padded detections (from box_decode with max_out = Nq)."""
    lib = _lib.load()
    B, M, D = boxes.shape
    dev = boxes.device
    ob = torch.zeros(B, max_out, D, device=dev)
    os_ = torch.zeros(B, max_out, device=dev)
    ol = torch.zeros(B, max_out, device=dev, dtype=torch.int32)
    oc = torch.zeros(B, device=dev, dtype=torch.int32)
    ct = (C.c_int32 * num_classes)(*[int(v) for v in class_task])
    st = lib.ff3d_circle_nms(_chk(boxes, name='boxes'), _chk(scores, name='scores'), _chk(labels, torch.int32, 'labels'),
                             _chk(count, torch.int32, 'count'), _chk(ob), _chk(os_), _chk(ol, torch.int32),
                             _chk(oc, torch.int32), B, M, D, max_out, num_classes, ct, len(task_radius),
                             _floats(task_radius), post_max_size, _stream())
    _lib.check(st, 'ff3d_circle_nms')
    return ob, os_, ol, oc


def rotate_nms(boxes, scores, labels, count, num_classes, class_task, task_thresh, pre_max_size, post_max_size,
               max_out=200):
    """FD:1369-1393 for nms_type='rotate' (mmdet3d nms_gpu per task) on padded detections (box_decode, max_out = Nq)."""
    lib = _lib.load()
    B, M, D = boxes.shape
    dev = boxes.device
    ob = torch.zeros(B, max_out, D, device=dev)
    os_ = torch.zeros(B, max_out, device=dev)
    ol = torch.zeros(B, max_out, device=dev, dtype=torch.int32)
    oc = torch.zeros(B, device=dev, dtype=torch.int32)
    ct = (C.c_int32 * num_classes)(*[int(v) for v in class_task])
    big = 1 << 30
    st = lib.ff3d_rotate_nms(_chk(boxes, name='boxes'), _chk(scores, name='scores'), _chk(labels, torch.int32, 'labels'),
                             _chk(count, torch.int32, 'count'), _chk(ob), _chk(os_), _chk(ol, torch.int32),
                             _chk(oc, torch.int32), B, M, D, max_out, num_classes, ct, len(task_thresh),
                             _floats(task_thresh), big if pre_max_size is None else int(pre_max_size),
                             big if post_max_size is None else int(post_max_size), _stream())
    _lib.check(st, 'ff3d_rotate_nms')
    return ob, os_, ol, oc


def boxes_iou_bev(boxes_a, boxes_b):
    """mmdet3d `boxes_iou_bev`: rotated BEV IoU of (N, 5) x (M, 5) boxes (x1, y1, x2, y2, angle) -> (N, M)."""
    lib = _lib.load()
    N, M = boxes_a.shape[0], boxes_b.shape[0]
    out = torch.empty(N, M, device=boxes_a.device)
    if N == 0 or M == 0:
        return out
    st = lib.ff3d_boxes_iou_bev(_chk(boxes_a, name='boxes_a'), _chk(boxes_b, name='boxes_b'), _chk(out), N, M, _stream())
    _lib.check(st, 'ff3d_boxes_iou_bev')
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """mmdet3d ``BboxOverlaps3D(coordinate='lidar')``: 3-D IoU of (N, >=7) x (M, >=7) LiDAR boxes (x, y, z_bottom, dx, dy,
    dz, yaw, ...) -> (N, M).  The IoU term of HungarianAssigner3D's matching cost (hungarian_assigner.py:128-129)."""
    lib = _lib.load()
    N, M = boxes_a.shape[0], boxes_b.shape[0]
    out = torch.empty(N, M, device=boxes_a.device)
    if N == 0 or M == 0:
        return out
    st = lib.ff3d_boxes_iou3d(_chk(boxes_a, name='boxes_a'), _chk(boxes_b, name='boxes_b'), _chk(out), N, M,
                              boxes_a.shape[1], boxes_b.shape[1], _stream())
    _lib.check(st, 'ff3d_boxes_iou3d')
    return out


def gaussian_heatmap_targets(gt_boxes, gt_labels, num_classes, H, W, coder, gaussian_overlap, min_radius):
    """FD:1133-1158: dense heatmap training target (K, H, W) of one sample from its ground-truth boxes (m, >=7) with
    gravity-or-bottom z (unused), int64 labels; coder = (out_size_factor, voxel_x, voxel_y, pc_x, pc_y)."""
    lib = _lib.load()
    heat = torch.zeros(num_classes, H, W, device=gt_boxes.device)
    m = gt_boxes.shape[0]
    if m:
        st = lib.ff3d_gaussian_heatmap_targets(_chk(gt_boxes, name='gt_boxes'), _chk(gt_labels, torch.int64, 'gt_labels'),
                                               _chk(heat), m, gt_boxes.shape[1], num_classes, H, W, _floats(coder),
                                               float(gaussian_overlap), int(min_radius), _stream())
        _lib.check(st, 'ff3d_gaussian_heatmap_targets')
    return heat


def nms_bev(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """mmdet3d `nms_gpu`: rotated-IoU NMS of (n, 5) xyxyr boxes -> kept original indices (int64), best score first.
    The kept count is data dependent, so this op ends with one host read (as the reference's `keep[:num_out]`)."""
    lib = _lib.load()
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    keep = torch.empty(n, dtype=torch.int32, device=boxes.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    big = 1 << 30
    st = lib.ff3d_nms_bev(_chk(boxes, name='boxes'), _chk(scores, name='scores'), float(thresh),
                          big if pre_maxsize is None else int(pre_maxsize),
                          big if post_max_size is None else int(post_max_size), _chk(keep, torch.int32),
                          _chk(cnt, torch.int32), n, _stream())
    _lib.check(st, 'ff3d_nms_bev')
    return keep[:int(cnt.item())].long()


def lss_cells(rots, trans, xs, ys, ds, lower, dx, nx, post_rots_inv=None, post_trans=None, extra_rots=None,
              extra_trans=None):
    """Frustum geometry + voxel binning (lss.py:232-276, :324-337): rots (B, N, 3, 3), trans (B, N, 3), frustum axes
    xs (fW) / ys (fH) / ds (D); lower / dx / nx host triples (x, y, z).  -> int32 keys (B*N*fH*fW*D), entry order
    (((b*N + n)*fH + h)*fW + w)*D + d; key = ((b*nz + z)*nx + x)*ny + y, or the cell count for points outside the grid."""
    lib = _lib.load()
    B, N = rots.shape[:2]
    opt = lambda t, name: None if t is None else _chk(t, name=name)                      # noqa: E731
    keys = torch.empty(B * N * ys.numel() * xs.numel() * ds.numel(), dtype=torch.int32, device=rots.device)
    nxa = (C.c_int32 * 3)(*[int(v) for v in nx])
    st = lib.ff3d_lss_cells(_chk(rots, name='rots'), _chk(trans, name='trans'), opt(post_rots_inv, 'post_rots_inv'),
                            opt(post_trans, 'post_trans'), opt(extra_rots, 'extra_rots'), opt(extra_trans, 'extra_trans'),
                            _chk(xs, name='xs'), _chk(ys, name='ys'), _chk(ds, name='ds'), B, N, ds.numel(), ys.numel(),
                            xs.numel(), _floats(lower), _floats(dx), nxa, _chk(keys, torch.int32), _stream())
    _lib.check(st, 'ff3d_lss_cells')
    return keys


def lss_splat(feat, depth, src, cell_offsets, n_cells):
    """Fused lift-splat (lss.py:132-141 + :339-362): feat (P, C) row view with unit inner stride (column block of a
    wider GEMM output is fine), depth (P, D), entries ``src`` = pixel*D + d sorted by cell and the (n_cells + 1) offsets
    table -> (n_cells, C), empty cells zero."""
    lib = _lib.load()
    P, C_ = feat.shape
    if not (feat.is_cuda and feat.dtype == torch.float32 and feat.stride(1) == 1):
        raise RuntimeError('feat: expected a CUDA fp32 (P, C) view with unit inner stride')
    if cell_offsets.numel() != n_cells + 1:
        raise RuntimeError('cell_offsets: expected n_cells + 1 entries')
    out = torch.empty(n_cells, C_, device=feat.device)
    st = lib.ff3d_lss_splat(C.c_void_p(feat.data_ptr()), feat.stride(0), _chk(depth, name='depth'), depth.shape[1],
                            _chk(src, torch.int32, 'src'), _chk(cell_offsets, torch.int32, 'cell_offsets'), _chk(out), C_,
                            n_cells, _stream())
    _lib.check(st, 'ff3d_lss_splat')
    return out


# ------------------------------------------------------------------------------- split-fp16 dense layers (splitmm.hip)
class Pair(tuple):
    """A split-fp16 operand: ``(hi, lo')`` fp16 planes (unpacks / indexes like the 2-tuple it used to be) plus the
    range-normalisation scalars of include/ff3d.h:
      exp    device int32 [1]: x = 2^exp * (hi + lo'/2048), |x| * 2^-exp < 2^15       (None: exponent 0)
      bound  weights only, device fp32 [2]: {largest row sum of |W|, max|bias|} - the kernels derive the exponent of a
             pair OUTPUT from it (|out| <= 2^(exp_in+15) * L1 + max|bias|), no pass over the output needed."""

    def __new__(cls, hi, lo, exp=None, bound=None):
        self = super().__new__(cls, (hi, lo))
        self.exp, self.bound = exp, bound
        return self

    def __getnewargs__(self):                   # copy / pickle rebuild through __new__(cls, hi, lo, exp, bound)
        return (self[0], self[1], self.exp, self.bound)

    def __deepcopy__(self, memo):
        """Planes, exponent and bound are copied (the planes share one storage: torch's deepcopy keeps that, zero rows included);
        the cached ctypes argument tuples (_lin_args / _rows_args: raw device pointers) are NOT - they are rebuilt on first use."""
        import copy
        return Pair(copy.deepcopy(self[0], memo), copy.deepcopy(self[1], memo), copy.deepcopy(self.exp, memo),
                    copy.deepcopy(self.bound, memo))

    def map(self, fn):
        """Same exponent, both planes through ``fn`` (views / reshapes)."""
        return Pair(fn(self[0]), fn(self[1]), self.exp, self.bound)

    def view(self, *shape):
        return self.map(lambda t: t.view(*shape))

    def value(self):
        """fp32 tensor in real units (tests / debugging)."""
        v = self[0].float() + self[1].float() / 2048.0
        return v if self.exp is None else torch.ldexp(v, self.exp.to(v.device).expand(v.shape).contiguous())


def plane_fits(rows, cols):
    """The split-fp16 kernels address an operand plane with 32-bit byte offsets: (rows + 1 zero row) * cols fp16 values must
    stay below 4 GiB (ff3d.h ZERO-ROW CONTRACT).  Callers fall back to the fp32 vendor path beyond that."""
    return (rows + 1) * cols * 2 < (1 << 32)


def as_pair(p):
    return p if isinstance(p, Pair) else Pair(p[0], p[1])


def _new_exp(device):
    return torch.empty(1, dtype=torch.int32, device=device)


def new_hint(device):
    """Persistent exponent guess of one fp32 -> pair call site (ff3d_split_f16): {guess, max|x| bits, redo, -} + the 64
    maximum slots of the conversion pass (FF3D_SPLIT_HINT_INTS int32)."""
    return torch.zeros(65 * 64, dtype=torch.int32, device=device)


def _scale_struct(a=None, w=None, res=None, a2=None, want_out=False):
    """ff3d_scale_t for one launch as a ctypes struct (+ the tensors it points to, kept alive by the caller's references) ->
    (struct, out_exp tensor | None)."""
    a, w = (as_pair(a) if a is not None else None), (as_pair(w) if w is not None else None)
    res, a2 = (as_pair(res) if res is not None else None), (as_pair(a2) if a2 is not None else None)
    ptr = lambda t: C.c_void_p(0 if t is None else t.data_ptr())                     # noqa: E731
    out_exp = None
    if want_out and w is not None and w.bound is not None:
        out_exp = _new_exp(w[0].device)
    st = _lib.Scale(ptr(a.exp if a else None), ptr(a2.exp if a2 else None), ptr(w.exp if w else None),
                    ptr(w.bound if (w and out_exp is not None) else None), ptr(res.exp if res else None), ptr(out_exp))
    return st, out_exp


def _scale(a=None, w=None, res=None, a2=None, want_out=False):
    """ff3d_scale_t for one launch, by reference -> (byref(struct), out_exp tensor | None)."""
    st, out_exp = _scale_struct(a, w, res, a2, want_out)
    return C.byref(st), out_exp


def _split_planes(rows, cols, device):
    """(2, rows + 1, cols) fp16: the (hi, lo') planes of a split operand, each followed by the zero row the kernels read for
    padding / ragged tiles (ff3d.h: ZERO-ROW CONTRACT)."""
    buf = torch.empty(2, rows + 1, cols, dtype=torch.float16, device=device)
    buf[:, rows].zero_()
    return buf


def split_f16(x, to_nhwc=False, hint=None):
    """fp32 -> range-normalised (hi, lo') fp16 Pair (x = 2^exp * (hi + lo'/2048)), each plane followed by a zero row.
    to_nhwc: x (B, C, H, W) -> two (B, H, W, C) tensors; otherwise x (..., K) -> two tensors of the same shape (rows of K).
    ``hint`` = new_hint(): the call site's persistent exponent guess (ff3d.h: one conversion pass when the guess holds, a
    second one otherwise - decided on the device); None: a fresh guess of 0."""
    lib = _lib.load()
    if hint is None:
        hint = new_hint(x.device)
    exp = _new_exp(x.device)
    hp, ep = _chk(hint, torch.int32, 'hint'), _chk(exp, torch.int32)
    if to_nhwc:
        B, C_, H, W = x.shape
        buf = _split_planes(B * H * W, C_, x.device)
        st = lib.ff3d_split_f16(_chk(x), C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr()), B, C_, H * W, 1,
                                hp, ep, _stream())
        shape = (B, H, W, C_)
    else:
        K = x.shape[-1]
        buf = _split_planes(x.numel() // K, K, x.device)
        st = lib.ff3d_split_f16(_chk(x), C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr()), 1, 1, x.numel(), 0,
                                hp, ep, _stream())
        shape = tuple(x.shape)
    _lib.check(st, 'ff3d_split_f16')
    return Pair(buf[0, :-1].view(shape), buf[1, :-1].view(shape), exp)


def split_f16_nhwc_group(xs, hints):
    """split_f16(to_nhwc=True) of up to four (B, C, H, W) maps of one shape in one launch per pass (ff3d_split_f16_nhwc_group);
    ``hints``: one persistent exponent guess (new_hint) per map.  -> list of Pairs."""
    lib = _lib.load()
    n = len(xs)
    B, C_, H, W = xs[0].shape
    dev = xs[0].device
    bufs = [_split_planes(B * H * W, C_, dev) for _ in range(n)]
    exps = [_new_exp(dev) for _ in range(n)]
    for x in xs:
        if tuple(x.shape) != (B, C_, H, W):
            raise RuntimeError('split_f16_nhwc_group: the maps must share one shape')
        _chk(x, name='x')
    for h in hints:
        _chk(h, torch.int32, 'hint')
    st = lib.ff3d_split_f16_nhwc_group(n, _ptr_array(xs), _ptr_array([b[0] for b in bufs]), _ptr_array([b[1] for b in bufs]),
                                       B, C_, H * W, _ptr_array(hints), _ptr_array(exps), _stream())
    _lib.check(st, 'ff3d_split_f16_nhwc_group')
    return [Pair(b[0, :-1].view(B, H, W, C_), b[1, :-1].view(B, H, W, C_), e) for b, e in zip(bufs, exps)]


def split_weight_f16(w, pad_rows_to=None, bias=None):
    """Split of a weight, once per weight load (cached by the caller): conv (N, C, 3, 3) -> two (N, 3, 3, C); linear (N, K)
    as is; each plane followed by a zero row.  pad_rows_to: zero rows appended up to that many output channels
    (conv3x3_small_f16x3).  The exponent (max|w| * 2^-exp in [2^13, 2^14)) and the output bound {L1(W), max|bias|} are
    computed on the device - nothing is read back."""
    if w.dim() == 4:
        w = w.permute(0, 2, 3, 1)
    w = w.detach().contiguous().float()
    if pad_rows_to is not None and w.shape[0] < pad_rows_to:
        w = torch.cat((w, w.new_zeros(pad_rows_to - w.shape[0], *w.shape[1:])), 0)
    N = w.shape[0]
    flat = w.view(N, -1)
    absw = flat.abs()
    _, ex = torch.frexp(absw.max())                    # max = m * 2^ex, m in [0.5, 1)  (0 -> ex = 0)
    exp = (ex - 14).to(torch.int32).view(1)
    bmax = bias.detach().float().abs().max() if bias is not None else flat.new_zeros(())
    bound = torch.stack((absw.sum(1).max(), bmax)).float().contiguous()
    ws = torch.ldexp(flat, (-exp).expand(flat.shape))  # exact power-of-two scaling
    buf = torch.zeros(2, N + 1, flat.shape[1], dtype=torch.float16, device=w.device)
    hi = ws.half()
    buf[0, :N] = hi
    buf[1, :N] = ((ws - hi.float()) * 2048.0).half()
    return Pair(buf[0, :N].view(w.shape), buf[1, :N].view(w.shape), exp, bound)


def _dense_event_start():
    if DENSE_EVENTS is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    return ev


def _dense_event_end(ev, tag, flops):
    if ev is not None:
        ev[1].record()
        DENSE_EVENTS.append((ev[0], ev[1], tag, flops))


def _plane(t, name):
    """Device pointer of a split operand plane (a contiguous view whose storage continues with the zero row)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float16 and t.is_contiguous()):
        raise RuntimeError(f'{name}: expected a contiguous CUDA fp16 plane from split_f16 / split_weight_f16')
    need = (t.storage_offset() + t.numel() + t.shape[-1]) * 2
    if t.untyped_storage().nbytes() < need:
        raise RuntimeError(f'{name}: the plane is not followed by its zero row (use split_f16 / split_weight_f16)')
    return C.c_void_p(t.data_ptr())


# weight planes of the halo-tile conv in K-step tiles (ff3d_conv3x3_halo_f16x3_tiled; FF3D_HALO_W_TILED=0: row-major planes, rounds 3-5)
HALO_W_TILED = os.environ.get('FF3D_HALO_W_TILED', '1') != '0'


def _halo_tiled_weight(w_split, N, C_):
    """(9 C / 32, N + 1, 32) K-step tiles of both planes of a split conv weight (tap-major rows + zero row), made once per Pair object."""
    wt = getattr(w_split, '_halo_tiled', None)
    if wt is None:
        def tile(pl):                                    # pl: the (N, 3, 3, C) view of an (N + 1)-row plane; the tiles include the zero row
            _plane(pl, 'w')                              # (validates: contiguous, followed by its zero row)
            return torch.as_strided(pl, (N + 1, 9 * C_ // 32, 32), (9 * C_, 32, 1)).permute(1, 0, 2).contiguous()
        wt = (tile(w_split[0]), tile(w_split[1]))
        try:
            w_split._halo_tiled = wt
        except AttributeError:
            pass
    return wt


# round 6 (third session): K slices of the implicit-GEMM conv with an fp32 NCHW result (ff3d_conv3x3_f16x3_splitk).  A conv whose output is
# only a few 128 x 128 tiles leaves most of the chip idle while every block walks all 9 * C / 32 K-steps (the BEV pyramid's stride-2 convs
# at 1 - 4 frames: 128 / 32 tiles, 72 steps): slices become extra blocks, up to CONV_KSPLIT_BLOCKS of them with at least
# CONV_KSPLIT_MIN_STEPS K-steps each - 47 vs 94 us (180 x 180 -> 90 x 90, one frame), 30 vs 95 us (90 x 90 -> 45 x 45), 49 vs 101 us (the
# latter at four frames); from 257 tiles on the one-pass kernels are level or ahead (profiles/r06_ks1_conv_splitk_sweep.txt).
# FF3D_CONV_KSPLIT=0: never.
CONV_KSPLIT = os.environ.get('FF3D_CONV_KSPLIT', '1') != '0'
CONV_KSPLIT_MAX_TILES = int(os.environ.get('FF3D_CONV_KSPLIT_MAX_TILES', '256'))
CONV_KSPLIT_BLOCKS = int(os.environ.get('FF3D_CONV_KSPLIT_BLOCKS', '512'))
CONV_KSPLIT_MIN_STEPS = int(os.environ.get('FF3D_CONV_KSPLIT_MIN_STEPS', '12'))


def conv_ksplit(M, N, K):
    """Number of K slices conv3x3_f16x3 uses for an (M = B * Ho * Wo) x N x K implicit GEMM with an fp32 result (1: the one-pass kernels)."""
    forced = os.environ.get('FF3D_CONV_KSPLIT_FORCE')
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if forced:
        return max(1, min(int(forced), K // 32, 64))
    if not CONV_KSPLIT or tiles > CONV_KSPLIT_MAX_TILES:
        return 1
    return max(1, min(CONV_KSPLIT_BLOCKS // tiles, (K // 32) // CONV_KSPLIT_MIN_STEPS, 64))


def conv3x3_f16x3(x_split, w_split, bias=None, relu=False, stride=1, split_out=False, nhwc_out=False):
    """3x3 conv, padding 1, fp32-class accuracy on the fp16 matrix cores: x_split = split_f16(x, to_nhwc=True),
    w_split = split_weight_f16(weight[, bias=bias]) -> (B, N, Ho, Wo) fp32, or with split_out the (hi, lo') NHWC Pair
    (B, Ho, Wo, N) x 2 for a following split-fp16 layer.  The fp32 result carries its bound exponent as ``._ff3d_exp``
    (|out| < 2^(e+15)) when the weights carry a bound."""
    lib = _lib.load()
    xh, xl = x_split
    wh, wl = w_split
    B, H, W, C_ = xh.shape
    N = wh.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    sc, out_exp = _scale(x_split, w_split, want_out=True)
    halo_blocks = B * ((H + 3) // 4) * ((W + 63) // 64) * ((N + 127) // 128)     # one 512-thread block per CU at a time
    if nhwc_out:
        # round 5: the result as NHWC fp32 (B, H, W, N) - the halo kernel's transposed-tile epilogue writing fp32 (camera maps for the
        # projection sampler); only the halo form has it
        if not (stride == 1 and N >= 64 and not split_out and CONV_HALO != '0'):
            raise RuntimeError('conv3x3_f16x3(nhwc_out=True): needs the halo-tile form (stride 1, N >= 64)')
        out = torch.empty(B, H, W, N, device=xh.device)
        ev = _dense_event_start()
        if HALO_W_TILED:
            wt = _halo_tiled_weight(w_split, N, C_)
            z = C.c_void_p(0)
            st = lib.ff3d_conv3x3_halo_f16x3_tiled(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), C.c_void_p(wt[0].data_ptr()),
                                                   C.c_void_p(wt[1].data_ptr()), _opt(bias, name='bias'), int(relu), z, z, z, _chk(out),
                                                   B, C_, H, W, N, sc, _stream())
        else:
            st = lib.ff3d_conv3x3_halo_f16x3_nhwc(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                                  _opt(bias, name='bias'), int(relu), _chk(out), B, C_, H, W, N, sc, _stream())
        _dense_event_end(ev, f'conv3x3 {C_}->{N} s1 {H}x{W} B={B} nhwc', 2.0 * B * H * W * N * 9 * C_)
        _lib.check(st, 'ff3d_conv3x3_halo_f16x3_nhwc')
        out._ff3d_exp = out_exp
        return out
    if stride == 1 and N >= 64 and (CONV_HALO == '1' or (CONV_HALO == 'auto' and halo_blocks >= 1024)):
        # halo-tile form: activations staged once per channel chunk instead of once per tap (convhalo.hip); needs >= 4
        # rounds of blocks over the 256 CUs, below that the implicit GEMM's finer tiles fill the chip better
        buf = _split_planes(B * H * W, N, xh.device) if split_out else None
        out = None if split_out else torch.empty(B, N, H, W, device=xh.device)
        ev = _dense_event_start()
        if HALO_W_TILED:                                  # round 5: K-step-tiled weight planes (ff3d.h), bit-identical results
            wt = _halo_tiled_weight(w_split, N, C_)
            st = lib.ff3d_conv3x3_halo_f16x3_tiled(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), C.c_void_p(wt[0].data_ptr()),
                                                   C.c_void_p(wt[1].data_ptr()), _opt(bias, name='bias'), int(relu), _opt(out),
                                                   C.c_void_p(buf[0].data_ptr() if split_out else 0),
                                                   C.c_void_p(buf[1].data_ptr() if split_out else 0), C.c_void_p(0), B, C_, H, W, N, sc,
                                                   _stream())
        else:
            st = lib.ff3d_conv3x3_halo_f16x3(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                             _opt(bias, name='bias'), int(relu), _opt(out),
                                             C.c_void_p(buf[0].data_ptr() if split_out else 0),
                                             C.c_void_p(buf[1].data_ptr() if split_out else 0), B, C_, H, W, N, sc, _stream())
        _dense_event_end(ev, f'conv3x3 {C_}->{N} s1 {H}x{W} B={B}', 2.0 * B * H * W * N * 9 * C_)
        _lib.check(st, 'ff3d_conv3x3_halo_f16x3')
        if split_out:
            return Pair(buf[0, :-1].view(B, H, W, N), buf[1, :-1].view(B, H, W, N), out_exp)
        out._ff3d_exp = out_exp
        return out
    if split_out:
        buf = _split_planes(B * Ho * Wo, N, xh.device)
        ev = _dense_event_start()
        st = lib.ff3d_conv3x3_f16x3_split_out(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                              _opt(bias, name='bias'), int(relu), C.c_void_p(buf[0].data_ptr()),
                                              C.c_void_p(buf[1].data_ptr()), B, C_, H, W, N, stride, sc, _stream())
        _dense_event_end(ev, f'conv3x3 {C_}->{N} s{stride} {H}x{W} B={B}', 2.0 * B * Ho * Wo * N * 9 * C_)
        _lib.check(st, 'ff3d_conv3x3_f16x3_split_out')
        return Pair(buf[0, :-1].view(B, Ho, Wo, N), buf[1, :-1].view(B, Ho, Wo, N), out_exp)
    out = torch.empty(B, N, Ho, Wo, device=xh.device)
    ks = conv_ksplit(B * Ho * Wo, N, 9 * C_)
    if ks > 1:
        # round 6: few row tiles and a long K walk (the pyramid's stride-2 convs at 1 - 4 frames) - K slices as extra blocks + a reduce
        ws = torch.empty(ks, B * Ho * Wo, N, device=xh.device)
        ev = _dense_event_start()
        st = lib.ff3d_conv3x3_f16x3_splitk(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                           _opt(bias, name='bias'), int(relu), _chk(out), B, C_, H, W, N, stride, ks, _chk(ws), sc,
                                           _stream())
        _dense_event_end(ev, f'conv3x3 {C_}->{N} s{stride} {H}x{W} B={B} ksplit={ks}', 2.0 * B * Ho * Wo * N * 9 * C_)
        _lib.check(st, 'ff3d_conv3x3_f16x3_splitk')
        out._ff3d_exp = out_exp
        return out
    ev = _dense_event_start()
    st = lib.ff3d_conv3x3_f16x3(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                _opt(bias, name='bias'), int(relu), _chk(out), B, C_, H, W, N, stride, sc, _stream())
    _dense_event_end(ev, f'conv3x3 {C_}->{N} s{stride} {H}x{W} B={B}', 2.0 * B * Ho * Wo * N * 9 * C_)
    _lib.check(st, 'ff3d_conv3x3_f16x3')
    out._ff3d_exp = out_exp
    return out


# round 6: wide stride-1 3x3 convs that read an NCHW fp32 map take it as it is (ff3d_conv3x3_halo_f16x3_nchwsrc: the fp32 -> pair conversion
# pass folded into the halo staging, 3.16 - 3.22 ms against 3.28 - 3.47 ms for conversion + conv at 32 x 256 x 180 x 180; same bits).
# FF3D_HALO_NCHW_SRC=0: convert first, as before (A/B record).
HALO_NCHW_SRC = os.environ.get('FF3D_HALO_NCHW_SRC', '1') != '0'


def conv3x3_nchwsrc_ok(x, w_split):
    """Whether conv3x3_f16x3_nchwsrc applies to this map: the halo-tile form would run (conv3x3_f16x3's own rule) in its 4 x 64 geometry,
    and the frame fits the kernel's 31-bit per-lane offsets."""
    B, C_, H, W = x.shape
    N = w_split[0].shape[0]
    if not (HALO_NCHW_SRC and HALO_W_TILED and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C_ % 32 == 0 and N >= 64):
        return False
    halo_blocks = B * ((H + 3) // 4) * ((W + 63) // 64) * ((N + 127) // 128)
    if not (CONV_HALO == '1' or (CONV_HALO == 'auto' and halo_blocks >= 1024)):
        return False
    pad0, pad1 = ((H + 3) // 4 * 4) * ((W + 63) // 64 * 64), ((H + 7) // 8 * 8) * ((W + 31) // 32 * 32)
    return not (pad1 * 100 < pad0 * 99) and C_ * H * W * 4 < (1 << 31) and plane_fits(B * H * W, N)


def conv3x3_f16x3_nchwsrc(x, hint, w_split, bias=None, relu=False, split_out=False):
    """conv3x3_f16x3(split_f16(x, to_nhwc=True, hint=hint), w_split, ...) without the conversion pass: x (B, C, H, W) fp32 NCHW as the caller
    holds it, ``hint`` = new_hint() of the call site (the exponent record, same protocol as split_f16).  Bit-identical to that pair of
    calls.  Check conv3x3_nchwsrc_ok(x, w_split) first."""
    lib = _lib.load()
    wh, wl = w_split
    B, C_, H, W = x.shape
    N = wh.shape[0]
    st_, out_exp = _scale_struct(None, w_split, want_out=True)
    buf = _split_planes(B * H * W, N, x.device) if split_out else None
    out = None if split_out else torch.empty(B, N, H, W, device=x.device)
    wt = _halo_tiled_weight(w_split, N, C_)
    ev = _dense_event_start()
    st = lib.ff3d_conv3x3_halo_f16x3_nchwsrc(_chk(x, name='x'), _chk(hint, torch.int32, 'hint'), C.c_void_p(wt[0].data_ptr()),
                                             C.c_void_p(wt[1].data_ptr()), _opt(bias, name='bias'), int(relu), _opt(out),
                                             C.c_void_p(buf[0].data_ptr() if split_out else 0),
                                             C.c_void_p(buf[1].data_ptr() if split_out else 0), B, C_, H, W, N, C.byref(st_), _stream())
    _dense_event_end(ev, f'conv3x3 {C_}->{N} s1 {H}x{W} B={B}', 2.0 * B * H * W * N * 9 * C_)
    _lib.check(st, 'ff3d_conv3x3_halo_f16x3_nchwsrc')
    if split_out:
        return Pair(buf[0, :-1].view(B, H, W, N), buf[1, :-1].view(B, H, W, N), out_exp)
    out._ff3d_exp = out_exp
    return out


# weight planes of the heatmap heads' tail conv in chunk tiles (ff3d_conv3x3_small_f16x3_tiled).  Measured level with the row-major planes
# (1299.6 / 1294.4 vs 1302.2 / 1291.4 frames/s, profiles/r05_ai_*; bit-identical results): off by default, FF3D_TAIL_W_TILED=1 selects it
TAIL_W_TILED = os.environ.get('FF3D_TAIL_W_TILED', '0') == '1'


def conv3x3_small_f16x3(x_split, w_split, bias, K):
    """Heatmap-head tail (FD:213-220): conv3x3 (C -> K <= 16) + bias on the (hi, lo') NHWC pair of the preceding
    conv3x3_f16x3(split_out=True); w_split = split_weight_f16(weight, pad_rows_to=16) -> (B, K, H, W) fp32 logits."""
    lib = _lib.load()
    xh, xl = x_split
    wh, wl = w_split
    B, H, W, C_ = xh.shape
    if wh.shape[0] != 16:
        raise RuntimeError('conv3x3_small_f16x3: weights must be class-padded to 16 rows (split_weight_f16(w, pad_rows_to=16))')
    out = torch.empty(B, K, H, W, device=xh.device)
    sc, _ = _scale(x_split, w_split)
    if TAIL_W_TILED:                                      # round 5: (C / 32, 9, 16, 32) chunk tiles, made once per Pair object
        wt = getattr(w_split, '_tail_tiled', None)
        if wt is None:
            tile = lambda pl: pl.reshape(16, 9, C_ // 32, 32).permute(2, 1, 0, 3).contiguous()    # noqa: E731
            wt = (tile(wh), tile(wl))
            try:
                w_split._tail_tiled = wt
            except AttributeError:
                pass
        st = lib.ff3d_conv3x3_small_f16x3_tiled(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), C.c_void_p(wt[0].data_ptr()),
                                                C.c_void_p(wt[1].data_ptr()), _opt(bias, name='bias'), _chk(out), B, C_, H, W, K, sc,
                                                _stream())
    else:
        st = lib.ff3d_conv3x3_small_f16x3(_plane(xh, 'x_hi'), _plane(xl, 'x_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                          _opt(bias, name='bias'), _chk(out), B, C_, H, W, K, sc, _stream())
    _lib.check(st, 'ff3d_conv3x3_small_f16x3')
    return out


# grouped heatmap heads: smallest total grid (blocks of the halo form) for which the grouped launch is used
HEADS_GROUP_MIN_BLOCKS = int(os.environ.get('FF3D_HEADS_GROUP_MIN_BLOCKS', '768'))    # one frame: 3 x 270 = 810 blocks, 476 vs 461 frames/s


def _ptr_array(items):
    return (C.c_void_p * len(items))(*[0 if t is None else t.data_ptr() for t in items])


def heatmap_heads_group(x_splits, w1_splits, b1s, w2_splits, b2s, K):
    """The heatmap heads of a multi-stage head in TWO launches instead of 2 x S: S first convs (conv3x3 C -> C + shift + ReLU,
    pair output: ff3d_conv3x3_halo_f16x3_group) and S tail convs (C -> K <= 16: ff3d_conv3x3_small_f16x3_group), all inputs of one
    shape.  -> list of (B, K, H, W) fp32 logits, or None when the grouped halo form does not apply (the caller runs them one by
    one).  Grid arithmetic: at 4 frames one conv is 4.2 rounds of blocks on the 256 CUs (5 with the last 22 % full), three in one
    grid 12.7; the tail conv 1.08 rounds of the 512 resident blocks (2) against 3.2 (4)."""
    lib = _lib.load()
    n = len(x_splits)
    xh0 = x_splits[0][0]
    B, H, W, C_ = xh0.shape
    N = w1_splits[0][0].shape[0]
    if not (1 < n <= 4 and N >= 64 and N % 32 == 0 and CONV_HALO != '0'
            and all(tuple(x[0].shape) == (B, H, W, C_) for x in x_splits)
            and all(w[0].shape[0] == N for w in w1_splits) and all(w[0].shape[0] == 16 for w in w2_splits)):
        return None
    halo_blocks = n * B * ((H + 3) // 4) * ((W + 63) // 64) * ((N + 127) // 128)
    if CONV_HALO == 'auto' and halo_blocks < HEADS_GROUP_MIN_BLOCKS:
        return None
    bufs = [_split_planes(B * H * W, N, xh0.device) for _ in range(n)]
    scs = [_scale_struct(x, w, want_out=True) for x, w in zip(x_splits, w1_splits)]
    sarr = (C.POINTER(_lib.Scale) * n)(*[C.pointer(st) for st, _ in scs])
    for x, w in zip(x_splits, w1_splits):
        _plane(x[0], 'x_hi'), _plane(x[1], 'x_lo'), _plane(w[0], 'w_hi'), _plane(w[1], 'w_lo')
    ev = _dense_event_start()
    st = lib.ff3d_conv3x3_halo_f16x3_group(n, _ptr_array([x[0] for x in x_splits]), _ptr_array([x[1] for x in x_splits]),
                                           _ptr_array([w[0] for w in w1_splits]), _ptr_array([w[1] for w in w1_splits]),
                                           _ptr_array(b1s), 1, C.c_void_p(0), _ptr_array([b[0] for b in bufs]),
                                           _ptr_array([b[1] for b in bufs]), B, C_, H, W, N,
                                           C.cast(sarr, C.c_void_p), _stream())
    _dense_event_end(ev, f'conv3x3 {C_}->{N} s1 {H}x{W} B={B} x{n}', 2.0 * n * B * H * W * N * 9 * C_)
    _lib.check(st, 'ff3d_conv3x3_halo_f16x3_group')
    ys = [Pair(b[0, :-1].view(B, H, W, N), b[1, :-1].view(B, H, W, N), e) for b, (_, e) in zip(bufs, scs)]
    outs = [torch.empty(B, K, H, W, device=xh0.device) for _ in range(n)]
    scs2 = [_scale_struct(y, w) for y, w in zip(ys, w2_splits)]
    sarr2 = (C.POINTER(_lib.Scale) * n)(*[C.pointer(st) for st, _ in scs2])
    for w in w2_splits:
        _plane(w[0], 'w_hi'), _plane(w[1], 'w_lo')
    st = lib.ff3d_conv3x3_small_f16x3_group(n, _ptr_array([y[0] for y in ys]), _ptr_array([y[1] for y in ys]),
                                            _ptr_array([w[0] for w in w2_splits]), _ptr_array([w[1] for w in w2_splits]),
                                            _ptr_array(b2s), _ptr_array(outs), B, N, H, W, K, C.cast(sarr2, C.c_void_p), _stream())
    _lib.check(st, 'ff3d_conv3x3_small_f16x3_group')
    return outs


_KSPLIT_FORCE = int(os.environ.get('FF3D_GEMM_KSPLIT_FORCE', '0'))          # tuning hook: this many slices for every long-K GEMM


def gemm_ksplit(M, N, K):
    """K slices for gemm_f16x3: long-K GEMMs whose 128x128 tiles do not fill the 512 resident blocks of the chip
    (roi_mlp.0: K = 37 632, 20 tiles at batch 1) are cut so that tiles x slices is one full round of blocks.  A grid of a few
    rounds is cut so that its LAST round is full: 600 tiles (batch 32) are 1.17 rounds - 58 % of the slots busy over two rounds;
    x 2 = 2.34 rounds (78 %), x 5 = 5.86 rounds (97.7 %) for 0.4 GB of partial planes written and read back."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    nk = K // 32
    if not GEMM_KSPLIT or K < 2048:
        return 1
    if _KSPLIT_FORCE > 0:
        return min(_KSPLIT_FORCE, max(1, nk // 8))
    if tiles <= 256:
        return max(1, min(512 // tiles, nk // 8, 64))
    if tiles < 2048:
        # the slice count (<= 6) with the fullest last round; ties -> fewer slices (less partial-plane traffic)
        best, best_eff = 1, 0.0
        for ks in range(1, 7):
            if nk // ks < 64:
                break
            rounds = tiles * ks / 512.0
            eff = rounds / -(-tiles * ks // 512)
            if eff > best_eff + 0.02:
                best, best_eff = ks, eff
        return best
    return 1


def gemm_f16x3(a_split, w_split, bias=None, relu=False, ksplit=None):
    """out (M, N) = A (M, K) @ W (N, K)^T + bias with the split-fp16 scheme (deterministic split-K for long-K GEMMs with
    few tiles: slices write partial planes, a second kernel adds them in order with bias / ReLU fused)."""
    lib = _lib.load()
    ah, al = a_split
    wh, wl = w_split
    M, K = ah.shape
    N = wh.shape[0]
    ks = gemm_ksplit(M, N, K) if ksplit is None else int(ksplit)
    out = torch.empty(M, N, device=ah.device)
    ws = torch.empty(ks, M, N, device=ah.device) if ks > 1 else None
    sc, out_exp = _scale(a_split, w_split, want_out=True)
    ev = _dense_event_start()
    st = lib.ff3d_gemm_f16x3(_plane(ah, 'a_hi'), _plane(al, 'a_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                             _opt(bias, name='bias'), int(relu), _chk(out), M, N, K, ks, _opt(ws), sc, _stream())
    _dense_event_end(ev, f'gemm {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_gemm_f16x3')
    out._ff3d_exp = out_exp
    return out


def gemm_f16x3_rowbias(a_split, w_split, table, nbatch):
    """out (nbatch*rows, N) = A @ W^T + table[row within the frame] (table (rows, N) fp32, shared by the frames): every value
    projection of the decoder in one launch over the raw pyramid (ff3d.h)."""
    lib = _lib.load()
    ah, al = a_split
    wh, wl = w_split
    M, K = ah.shape
    N = wh.shape[0]
    rows = M // nbatch
    assert rows * nbatch == M and tuple(table.shape) == (rows, N)
    out = torch.empty(M, N, device=ah.device)
    sc, _ = _scale(a_split, w_split)
    ev = _dense_event_start()
    st = lib.ff3d_gemm_f16x3_rowbias(_plane(ah, 'a_hi'), _plane(al, 'a_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                     _chk(table, name='table'), _chk(out), nbatch, rows, N, K, sc, _stream())
    _dense_event_end(ev, f'gemm {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_gemm_f16x3_rowbias')
    return out


# ------------------------------------------------------------------------------- NHWC pair pipeline (nhwcpair.hip)
def gemm_f16x3_fused(a_split, w_split, bias=None, act=0, residual=None, pair_out=False):
    """1x1-conv layer on NHWC pairs: (M, K) pair @ (N, K) pair^T + bias (+ residual pair) with act 0 / 1 ReLU / 2 ReLU6 ->
    fp32 (M, N), or with pair_out the (hi, lo') Pair (exponent from the layer's bound, ff3d.h)."""
    lib = _lib.load()
    ah, al = a_split
    wh, wl = w_split
    M, K = ah.shape
    N = wh.shape[0]
    rh, rl = residual if residual is not None else (None, None)
    z = C.c_void_p(0)
    sc, out_exp = _scale(a_split, w_split, res=residual, want_out=pair_out)
    if pair_out:
        buf = _split_planes(M, N, ah.device)
        out, oh, ol = z, C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr())
    else:
        res = torch.empty(M, N, device=ah.device)
        out, oh, ol = _chk(res), z, z
    ev = _dense_event_start()
    st = lib.ff3d_gemm_f16x3_fused(_plane(ah, 'a_hi'), _plane(al, 'a_lo'), _plane(wh, 'w_hi'), _plane(wl, 'w_lo'),
                                   _opt(bias, name='bias'), int(act), z if rh is None else _plane(rh, 'res_hi'),
                                   z if rl is None else _plane(rl, 'res_lo'), out, oh, ol, M, N, K, 1, z, sc, _stream())
    _dense_event_end(ev, f'gemm {M}x{K}x{N}', 2.0 * M * N * K)
    _lib.check(st, 'ff3d_gemm_f16x3_fused')
    return Pair(buf[0, :-1], buf[1, :-1], out_exp) if pair_out else res


def dw_bound(weight, bias=None):
    """Output bound scalars of a depthwise 3x3 layer: device fp32 [2] {max_c sum|w_c|, max|bias|} (once per weight load)."""
    bmax = bias.detach().float().abs().max() if bias is not None else weight.new_zeros(())
    return torch.stack((weight.detach().float().abs().flatten(1).sum(1).max(), bmax)).float().contiguous()


def dwconv3x3_pair(x0, x1, weight, bias, act, B, H, W, bound=None):
    """Depthwise 3x3 (+ bias + act 0 / 1 / 2 = none / ReLU / ReLU6) over cat(x0, x1) (x1 may be None): NHWC pairs of rows
    B*H*W -> Pair (B*H*W, C0 + C1).  weight (C, 9) fp32; ``bound`` = dw_bound(weight, bias) (cached by the caller)."""
    lib = _lib.load()
    C0 = x0[0].shape[-1]
    C1 = 0 if x1 is None else x1[0].shape[-1]
    buf = _split_planes(B * H * W, C0 + C1, x0[0].device)
    z = C.c_void_p(0)
    if bound is None:
        bound = dw_bound(weight, bias)
    out_exp = _new_exp(x0[0].device)
    p0, p1 = as_pair(x0), (as_pair(x1) if x1 is not None else None)
    ptr = lambda t: C.c_void_p(0 if t is None else t.data_ptr())                     # noqa: E731
    sc = _lib.Scale(ptr(p0.exp), ptr(p1.exp if p1 is not None else None), z, ptr(bound), z, ptr(out_exp))
    st = lib.ff3d_dwconv3x3_pair(_plane(x0[0], 'x0_hi'), _plane(x0[1], 'x0_lo'), C0,
                                 z if x1 is None else _plane(x1[0], 'x1_hi'), z if x1 is None else _plane(x1[1], 'x1_lo'), C1,
                                 _chk(weight, name='weight'), _opt(bias, name='bias'), int(act),
                                 C.c_void_p(buf[0].data_ptr()), C.c_void_p(buf[1].data_ptr()), B, H, W, C.byref(sc), _stream())
    _lib.check(st, 'ff3d_dwconv3x3_pair')
    return Pair(buf[0, :-1], buf[1, :-1], out_exp)


def unsplit_f16(pair, B, H, W):
    """NHWC (hi, lo') pair with rows B*H*W -> fp32 NCHW (B, C, H, W) in real units."""
    lib = _lib.load()
    C_ = pair[0].shape[-1]
    out = torch.empty(B, C_, H, W, device=pair[0].device)
    exp = as_pair(pair).exp
    st = lib.ff3d_unsplit_f16(_plane(pair[0], 'hi'), _plane(pair[1], 'lo'), _opt(exp, torch.int32, 'exp'), _chk(out), B, C_,
                              H * W, _stream())
    _lib.check(st, 'ff3d_unsplit_f16')
    return out
