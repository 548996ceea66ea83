#!/usr/bin/env python
"""bench.py - decoder frames/s of the FocalFormer3D Hard-Instance-Probing head on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--channels C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of ``FocalDecoder.forward`` + ``get_bboxes`` over one batch of B synthetic frames per
GPU (features resident in HBM), plus - when N > 1 - the RCCL all-gather of the padded detections.
Workload = BASELINE.json configs[1]: FocalFormer3D_L, 3 HIP stages x 200 queries (Nq=600), 2 decoder
stages x 3 layers, RoI 7x7, 180x180x256 BEV.  Frames shard over ranks (weak scaling: B frames per GPU).
Rank 0 prints ONE JSON line (metric = BASELINE.json's).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MFMA_F16_PEAK_TF = 2500.0     # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='frames per GPU per step')
    ap.add_argument('--channels', type=int, default=256, help='BEV hidden width (BASELINE: 256; reference configs: 128)')
    ap.add_argument('--graph', action='store_true', help='replay the head from a captured hipGraph (launch-bound small batches)')
    ap.add_argument('--gemm-dtype', choices=['f32', 'bf16'], default='f32',
                    help="precision of the decoder's dense projections (f32 = parity path; bf16 = BASELINE config 5 mode)")
    ap.add_argument('--dense', choices=['default', 'f16x3', 'vendor'], default='default',
                    help="wide convs / large GEMMs: 'f16x3' = own split-fp16 MFMA kernels (fp32-class), 'vendor' = MIOpen / hipBLASLt fp32")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-frames', type=int, default=0, help='frames of the CPU-oracle sample (0 = auto, ~10-30 s)')
    return ap.parse_args()


def cpu_baseline(cfg, sd, C, budget_s=20.0, frames=0):
    """The CPU oracle (a port of the reference algorithm, oracle/ff3d_oracle.py) timed on the host cores on a
    bounded sample of the same workload: B=1 frames of the same shape, forward + get_bboxes."""
    from oracle import ff3d_oracle as O
    from focalformer3d_amd.synthetic import stage_features
    ocfg = O.head_config(
        num_proposals=cfg['num_proposals'], hidden_channel=C, num_classes=cfg['num_classes'],
        num_decoder_layers=cfg['num_decoder_layers'], nms_kernel_size=3, multiscale=True,
        multistage_heatmap=cfg['multistage_heatmap'], reuse_first_heatmap=True, extra_feat=True, bevpos=True,
        input_img=False, iterbev_wo_img=True, roi_feats=7, roi_expand_ratio=1.2, roi_based_reg=True,
        common_heads=cfg['common_heads'], voxel_size=tuple(cfg['bbox_coder']['voxel_size']))
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    inputs = stage_features(1, C, 180, 3, seed=123)
    cores = torch.get_num_threads()

    def one():
        with torch.no_grad():
            out, aux = O.focal_decoder_forward(sd, ocfg, inputs)
            O.focal_decoder_get_bboxes(out, aux, ocfg)
    t0 = time.perf_counter()
    one()                                   # warm-up (also sizes the sample)
    t_first = time.perf_counter() - t0
    n = frames or max(2, min(20, int(budget_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 4), unit='frames/s', cores=cores, kind='port',
                sample=f'{n} frames at batch 1 of the same workload (oracle/ff3d_oracle.py forward + get_bboxes, '
                       f'fp32, torch CPU, {cores} threads), after 1 warm-up frame')


def pmc_traffic(B, C):
    """HBM bytes per MSDA launch measured with rocprofv3 PMC counters for this exact configuration
    (profiles/pmc_msda.json, collected and corrected as MI355X_MICROARCH.md prescribes), or None."""
    try:
        for e in json.load(open(os.path.join(ROOT, 'profiles', 'pmc_msda.json')))['entries']:
            if e['batch'] == B and e['channels'] == C:
                return e['traffic_bytes']
    except Exception:
        pass
    return None


def pmc_dense(B, C):
    """PMC record of the dominant dense kernel for this configuration (profiles/pmc_dense.json), or None."""
    try:
        for e in json.load(open(os.path.join(ROOT, 'profiles', 'pmc_dense.json')))['entries']:
            if e['batch'] == B and e['channels'] == C:
                return e
    except Exception:
        pass
    return None


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus != world and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if a.gpus > 1 and world == 1:
        raise SystemExit('for --gpus > 1 launch through torch.distributed.run (one rank per GPU)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP decoder path has no CPU fallback')
    local_rank %= torch.cuda.device_count()                     # (only differs in the single-GPU rehearsal below)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  FF3D_BENCH_BACKEND=gloo exists only to rehearse the multi-rank control flow with
        # several ranks on ONE GPU (RCCL refuses duplicate devices); it is never used for reported numbers.
        backend = os.environ.get('FF3D_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from focalformer3d_amd import dist as fdist, ops
    from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features

    C, B = a.channels, a.batch
    cfg = focalformer3d_l_head_cfg(C=C, grid=180, num_proposals=200, stages=3, decoder_stages=2)
    head = build_head_from_cfg(cfg, seed=0, device=dev)
    if a.gemm_dtype == 'bf16':
        head.set_gemm_dtype(torch.bfloat16)
    if a.dense != 'default':
        head.set_dense_mode(a.dense)
    inputs = stage_features(B, C, 180, 3, seed=1 + rank, device=dev)
    metas = [{'box_type_3d': lambda t, box_dim=9: t}] * B

    graphed = None
    if a.graph:
        from focalformer3d_amd.runtime import GraphedHead
        graphed = GraphedHead(head, inputs)

    def step():
        if graphed is not None:
            boxes, scores, labels, count = graphed()                     # replay: inputs already in the static buffers
        else:
            preds = head(inputs, None, metas)
            boxes, scores, labels, count = head.get_bboxes_padded(preds)
        packed = fdist.gather_detections(boxes, scores, labels, count)   # RCCL all-gather when world > 1
        return packed, count

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync_all()
    ops.MSDA_EVENTS, ops.DENSE_EVENTS = [], []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        packed, count = step()
    counts = count.tolist()              # the host reads the detection counts of the last batch (get_bboxes' compaction)
    sync_all()
    elapsed = time.perf_counter() - t0
    events, ops.MSDA_EVENTS = ops.MSDA_EVENTS, None
    dense_events, ops.DENSE_EVENTS = ops.DENSE_EVENTS, None
    if not events:
        # graph replay hides the individual launches from the host: time the same launches (same tensors, same
        # stream) eagerly right after the timed region instead
        ops.MSDA_EVENTS, ops.DENSE_EVENTS = [], []
        for _ in range(max(2, min(a.steps, 5))):
            head.get_bboxes_padded(head(inputs, None, metas))
        torch.cuda.synchronize()
        events, ops.MSDA_EVENTS = ops.MSDA_EVENTS, None
        dense_events, ops.DENSE_EVENTS = ops.DENSE_EVENTS, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = [s.elapsed_time(e) for s, e, _ in events]
        avg_ms = sum(ms) / len(ms)
        alg_bytes = events[0][2]
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            'metric': 'decoder frames/sec @ 600 queries x 3 stages, 180x180 BEV',
            'value': round(world * B * a.steps / elapsed, 3),
            'unit': 'frames/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(elapsed / a.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32' + (' (wide convs / large GEMMs: fp32 operands as fp16 hi+lo pairs, 3 MFMA passes, f32 accumulate)'
                               if head.dense_mode == 'f16x3' else '')) if a.gemm_dtype == 'f32'
                     else 'bf16 decoder GEMMs + f32 heatmap/convs/gather accumulation',
            'data': 'synthetic',
            'config': {'workload': f'FocalFormer3D_L head: 3 HIP stages x 200 queries (Nq=600), 2 decoder stages x 3 '
                                   f'layers, RoI 7x7, 180x180x{C} BEV, K=10; FocalDecoder.forward + get_bboxes, features '
                                   f'resident in HBM (BASELINE.json configs[1])',
                       'frames_per_gpu_per_step': B, 'global_batch': B * world, 'channels': C,
                       'parallelism': f'frames sharded dp{world}' + (' + RCCL all-gather of padded detections' if world > 1 else ''),
                       'weights': 'random init of the reference architecture, BN statistics randomised',
                       'dense_layers': {'f16x3': 'wide 3x3 convs (+ large GEMMs) on own split-fp16 MFMA kernels: fp32 operands as '
                                                 '(hi, lo) fp16 pairs, 3 MFMA passes, fp32 accumulate; error vs fp64 = vendor fp32 path',
                                        'vendor': 'MIOpen / hipBLASLt fp32'}[head.dense_mode],
                       'execution': ('hipGraph replay' if a.graph else 'eager launches') +
                                    ', BEV positional embedding cached per weight load',
                       'detections_last_batch': counts},
            'roofline': {'kernel': f'msda_fwd_kernel (ff3d_msda_fused_fwd, {a.gemm_dtype} value)', 'bound': 'hbm',
                         'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(B, C),
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': round(avg_ms, 5),
                         'launches_timed': len(ms)},
        }
        if dense_events:
            # the kernel that now carries most of the step: split-fp16 implicit GEMM (3 MFMA passes per fp32 product)
            per = {}
            for s_, e_, tag, fl in dense_events:
                d_ = per.setdefault(tag, [0, 0.0, fl])
                d_[0] += 1
                d_[1] += s_.elapsed_time(e_)
            tag, (n_l, tot, fl) = max(per.items(), key=lambda kv: kv[1][1])
            avg = tot / n_l
            mfma_tf = 3.0 * fl / (avg * 1e-3) / 1e12
            out['roofline_dense'] = {
                'kernel': f'split-fp16 dense kernel, largest launch: {tag} ' + ('(conv3x3_halo_f16x3_kernel)' if ' s1 ' in tag and ops.CONV_HALO != '0' and B >= 16 else '(splitmm_kernel)'), 'bound': 'mfma', 'achieved': round(mfma_tf, 1), 'peak': MFMA_F16_PEAK_TF,
                'unit': 'TFLOP/s', 'frac': round(mfma_tf / MFMA_F16_PEAK_TF, 4),
                'traffic': (pmc_dense(B, C) or {}).get('traffic_bytes') if ' s1 ' in tag and B >= 16 else None,
                'mfma_busy_pmc': (pmc_dense(B, C) or {}).get('mfma_busy') if ' s1 ' in tag and B >= 16 else None,
                'executed_mfma_flops_per_launch': 3.0 * fl, 'algorithmic_fp32_flops_per_launch': fl,
                'fp32_equivalent_tflops': round(fl / (avg * 1e-3) / 1e12, 1), 'fp32_mfma_peak_tflops': 157.3,
                'avg_launch_ms': round(avg, 4), 'launches_timed': n_l,
                'all_dense_launches_ms_per_step': round(sum(v[1] for v in per.values()) / a.steps, 3)
                if len(events) == a.steps * 6 else None}
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, head.state_dict(), C, frames=a.cpu_frames)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
