"""Round 4 experiment: more than one batch in flight per GPU at small batch (the per-GPU share of BASELINE configs[3] is 4
frames per step; the 4-frame step leaves the chip partly idle during its ~100 launches of 5 - 30 us).

    python tools/experiments/exp_pipeline.py --mode MODE [--batch 4] [--steps 60] [--slots 2]

MODE:
  eager          head + get_bboxes_padded + pack, eager launches on one stream (the round-3 N > 1 form)
  graph          one captured graph (head + get_bboxes_padded + pack) replayed on one stream (the round-3 N = 1 form)
  graph2         --slots graphs, each with its own static buffers, replayed round-robin on --slots streams: consecutive batches
                 overlap on the GPU
  graph_cc       as graph, with the RCCL all-gather of the packed detections CAPTURED INSIDE the graph (1-rank group)
  graph2_cc      as graph2, collective captured inside every graph
Every mode runs in THIS process (start one process per mode: a replay fault takes the context with it).  Prints one JSON line.
The waiting discipline of runtime.py is kept: after the first replay the host waits on EVENTS only.
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='graph2')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--slots', type=int, default=2)
    ap.add_argument('--channels', type=int, default=256)
    ap.add_argument('--capture-mode', default='global', help="capture_error_mode of torch.cuda.graph ('thread_local': the RCCL "
                    "watchdog thread's event queries do not invalidate the capture)")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    cc = a.mode.endswith('_cc')
    if cc:
        import torch.distributed as dist
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
    from focalformer3d_amd import dist as fdist
    from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features
    B, C = a.batch, a.channels
    head = build_head_from_cfg(focalformer3d_l_head_cfg(C=C, grid=180, num_proposals=200, stages=3, decoder_stages=2), seed=0,
                               device=dev)
    n_slots = a.slots if a.mode.startswith('graph2') else 1
    import copy
    # one head per slot: the derived caches of a head (per-call-site exponent hints written by the guarded split, the
    # per-forward split memo) are per-forward device state and must not be shared by replays that overlap in time
    heads = [head] + [copy.deepcopy(head) for _ in range(n_slots - 1)]
    inputs = [stage_features(B, C, 180, 3, seed=1 + i, device=dev) for i in range(n_slots)]

    def eager_step(inp, packed, gathered, head=head):
        dets = head.get_bboxes_padded(head(inp, None, None))
        fdist.pack_detections(*dets, out=packed)
        if cc:
            torch.distributed.all_gather_into_tensor(gathered, packed)
        return dets

    packed = [torch.empty(B, 201, fdist.DET_COLS, device=dev) for _ in range(n_slots)]
    gathered = [torch.empty(B, 201, fdist.DET_COLS, device=dev) for _ in range(n_slots)]
    for i in range(3):                                               # warm-up: caches, lazy RCCL init
        for s in range(n_slots):
            eager_step(inputs[s], packed[s], gathered[s], heads[s])
    torch.cuda.synchronize()
    ref = [(gathered[s] if cc else packed[s]).clone() for s in range(n_slots)]

    if a.mode.startswith('eager'):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            eager_step(inputs[0], packed[0], gathered[0])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ok = True
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_slots)]
        graphs = []
        for s in range(n_slots):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                eager_step(inputs[s], packed[s], gathered[s], heads[s])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=a.capture_mode):
                eager_step(inputs[s], packed[s], gathered[s], heads[s])
            graphs.append(g)
        torch.cuda.synchronize()                                     # (last device-wide wait: no replay has run yet)
        done = [torch.cuda.Event() for _ in range(n_slots)]
        t0 = time.perf_counter()
        for i in range(a.steps):
            s = i % n_slots
            with torch.cuda.stream(streams[s]):
                graphs[s].replay()
                done[s].record()
        for e in done:
            e.synchronize()
        el = time.perf_counter() - t0
        got = [(gathered[s] if cc else packed[s]).cpu() for s in range(n_slots)]
        ok = all(torch.equal(g_, r_.cpu()) for g_, r_ in zip(got, ref))
    print(json.dumps({'mode': a.mode, 'capture_mode': a.capture_mode, 'batch': B, 'slots': n_slots, 'steps': a.steps, 'frames_per_s': round(B * a.steps / el, 1),
                      'ms_per_step': round(el / a.steps * 1e3, 4), 'results_equal_eager': ok}), flush=True)
    os._exit(0)                                                      # (no teardown after replays: see runtime.py)


if __name__ == '__main__':
    main()
