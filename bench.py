#!/usr/bin/env python
"""bench.py - decoder frames/s of the FocalFormer3D Hard-Instance-Probing head on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B | --global-batch G] [--channels C]

``--gpus N`` with N > 1 starts its own N ranks (one process per GPU, ``torch.distributed.run``, RCCL); when the driver
already launched the ranks (WORLD_SIZE set) the script runs as one of them.  This is the counterpart of the reference's
tools/dist_test.sh:9-11 + ``multi_gpu_test(..., gpu_collect)`` (tools/test.py:229-233).

One step = one pass of ``FocalDecoder.forward`` + ``get_bboxes`` over one batch of synthetic frames per GPU (features
resident in HBM) plus - when N > 1 - the RCCL all-gather of the padded detections, issued on a side stream so that it
overlaps the next batch.  Workload = BASELINE.json configs[1]: FocalFormer3D_L, 3 HIP stages x 200 queries (Nq=600),
2 decoder stages x 3 layers, RoI 7x7, 180x180x256 BEV.

Scaling modes:  default = WEAK (``--batch`` frames per GPU, 32);  ``--global-batch G`` = STRONG (G frames split over the
ranks: BASELINE.json configs[3] is ``--global-batch 32 --gpus 8``, 4 frames per GPU).  With N > 1 the weak-mode line
also carries a short measurement of the configs[3] mode (``configs3_strong``).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import signal
import socket
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MFMA_F16_PEAK_TF = 2500.0     # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = 'decoder frames/sec @ 600 queries x 3 stages, 180x180 BEV'
# untimed graph replays per slot between the eager warm-up steps and the timed region (graph upload, code objects, the two slots settling into
# their staggered steady state); FF3D_BENCH_WARM_REPLAYS overrides (A/B: profiles/r05_ae_*)
WARM_REPLAYS = int(os.environ.get('FF3D_BENCH_WARM_REPLAYS', '2'))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['l', 'lc', 'waymo'], default='l',
                    help="l = BASELINE configs[1] (FocalFormer3D_L head, the metric's configuration; default); lc = configs[2] "
                         "(6 x 256 x 232 x 400 camera maps -> I2P -> FocalEncoder 'bevfusion' -> head); waymo = configs[4] "
                         "(468 x 468 x 256 BEV, 1000 queries, bf16 decoder GEMMs)")
    ap.add_argument('--batch', type=int, default=0, help='frames per GPU per step (weak scaling); default 32 (l) / 8 (lc, waymo)')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling: total frames per step, split over the ranks (BASELINE configs[3]: 32)')
    ap.add_argument('--channels', type=int, default=256, help='BEV hidden width (BASELINE: 256; reference configs: 128)')
    ap.add_argument('--graph', choices=['auto', 'on', 'off'], default='auto',
                    help='replay the head (+ the detection packing + the RCCL all-gather when there is a process group) from captured '
                         'hipGraphs, --slots batches in flight (runtime.PipelinedHead).  auto: on whenever the head is fed directly '
                         '(every workload since round 5: lc captures neck + head as one unit).  A graphed step is ONE replay: the packing and the '
                         'collective are captured inside the graph (round 4).  (Rounds 2-5 also had to keep eager launches + host '
                         'synchronisations away from the replays - a GPU fault caused by memset nodes, fixed in round 6: DESIGN.md 5.3.)')
    ap.add_argument('--slots', type=int, default=0,
                    help='batches in flight per GPU in graph mode (0 = auto: 4 up to 8 frames per step, 2 above): consecutive steps are replayed round-robin from this many captured '
                         'graphs on as many streams and overlap on the device (profiles/r04_a_batches_in_flight_ab.txt)')
    ap.add_argument('--gemm-dtype', choices=['f32', 'bf16'], default=None,
                    help="precision of the decoder's dense projections (f32 = parity path; bf16 = BASELINE configs[4] mode, "
                         "the default of --workload waymo)")
    ap.add_argument('--dense', choices=['default', 'f16x3', 'vendor'], default='default',
                    help="wide convs / large GEMMs: 'f16x3' = own split-fp16 MFMA kernels (fp32-class), 'vendor' = MIOpen / hipBLASLt fp32")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=30.0,
                    help='seconds of CPU-oracle work for C2 (the benchmarked head; C1 gets a third): 1 warm-up + >= 5 timed frames')
    ap.add_argument('--cpu-full-protocol', action='store_true',
                    help='CPU baseline with the full protocol of tools/analysis_tools/benchmark.py:62-91 (5 warm-up + 20 timed)')
    ap.add_argument('--no-strong-probe', action='store_true', help='skip the configs[3] measurement appended when N > 1')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help="skip the compact records of --workload lc / waymo (BASELINE configs[2] / [4]) that the default N = 1 line "
                         "carries under 'other_workloads' (each measured in a child process, ~10 steps)")
    ap.add_argument('--watchdog', type=float, default=float(os.environ.get('FF3D_BENCH_WATCHDOG_S', '600')),
                    help='seconds a stage of the run may take before rank 0 prints a JSON line with "error" and every rank exits 3 '
                         '(0 = off): a hung replay / collective must not hang the driver')
    ap.add_argument('--fresh-inputs', type=int, default=0, metavar='N',
                    help='graph mode: rotate N >= 2 x slots DISTINCT batches (resident in HBM) through the slots INSIDE the timed region: '
                         "before every replay a producer stream writes the next batch into the slot's input buffers in place "
                         '(runtime.PipelinedHead.begin_fill / submit(filled=True)); the producer here is a device-side copy from the '
                         'pool, timed.  The default line carries this measurement as config.fresh_inputs (child process)')
    ap.add_argument('--latency-b1', action='store_true',
                    help="the reference's own protocol (tools/analysis_tools/benchmark.py:62-91): ONE frame per call, one call at a time, "
                         '5 warm-up + 20 timed calls, the host waits for every result.  Prints {"latency_b1_ms": ...}; the default line '
                         'carries it (child process)')
    ap.add_argument('--no-companions', action='store_true',
                    help='skip the fresh-input and batch-1 latency companions of the default N = 1 line (two child processes)')
    ap.add_argument('--value-mode', choices=['project_first', 'gather_first'], default='project_first',
                    help="cross-attention value path: 'project_first' (default, the form north_star names: value_proj over every BEV cell, "
                         "then the HBM-bound gather of projected head slices) or 'gather_first' (opt-in: the gather reads un-projected C-wide "
                         "rows per head, value_proj runs on the gathered rows; same operator).  The default line carries the opt-in "
                         "form's measurement as config.value_mode_gather_first (child process)")
    ap.add_argument('--scale-sweep', action='store_true',
                    help='one command, the whole scaling curve: for N = 1, 2, 4, 8 (as many as there are GPUs) run the weak line '
                         '(--batch frames per GPU) AND the strong line (--global-batch 32 = BASELINE configs[3]), each as its own '
                         '`bench.py --gpus N` job; prints every JSON line as it arrives and a final {"scale_sweep": ...} summary')
    ap.add_argument('--no-pin', action='store_true', help='do not pin the host threads of a rank to its own cores (N > 1)')
    ap.add_argument('--preflight-collective', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


class Watchdog:
    """No rank may hang the driver (VERDICT r04 #5c): every stage of the run (set-up + capture, warm-up, the timed replays, the
    verification, the per-kernel pass, the child measurements) re-arms a timer; a stage that does not finish within its
    allowance - a replay or collective that never completes, a rank that died and left the others in a barrier - ends the run:
    rank 0 prints ONE JSON line carrying ``"error"`` (metric / unit / n_gpus as in the normal line, ``value`` null) and every
    rank leaves with exit code 3.  The other ranks wait 20 s longer than rank 0 so that its line gets out before the launcher
    tears the job down; a SIGTERM from the launcher (another rank crashed) is answered the same way by a helper thread behind
    the interpreter's wake-up pipe - a Python-level handler would never run while the main thread sits inside a HIP / RCCL call."""

    def __init__(self, seconds, rank, world):
        self.seconds, self.rank, self.world = float(seconds), rank, world
        self.timer, self.name, self.done = None, 'start', False
        if self.seconds > 0 and threading.current_thread() is threading.main_thread():
            try:
                # the interpreter's C-level handler writes the signal number to the wake-up pipe at once, in whichever thread
                # the kernel picked; a helper thread reads it (a Python-level handler would wait for the main thread)
                self._rfd, wfd = os.pipe()
                os.set_blocking(wfd, False)
                signal.signal(signal.SIGTERM, lambda *_: None)
                signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
                threading.Thread(target=self._on_sigterm, daemon=True).start()
            except (ValueError, OSError):
                pass

    def _on_sigterm(self):
        while True:
            b = os.read(self._rfd, 1)
            if not b:
                return
            if b[0] == int(signal.SIGTERM):
                self._leave('terminated by the launcher (SIGTERM): another rank failed or the job was cancelled', 143)

    def stage(self, name, seconds=None):
        self.cancel()
        self.name = name
        if self.seconds > 0:
            allow = float(seconds if seconds is not None else self.seconds) + (0.0 if self.rank == 0 else 20.0)
            self.timer = threading.Timer(allow, self._leave, args=(
                f'watchdog: rank {self.rank} made no progress for {allow:.0f} s in stage "{name}"', 3))
            self.timer.daemon = True
            self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def finish(self):
        self.done = True
        self.cancel()

    def _leave(self, why, code):
        if self.done:
            return
        line = {'metric': METRIC, 'value': None, 'unit': 'frames/s', 'n_gpus': self.world, 'higher_is_better': True,
                'error': why, 'stage': self.name, 'rank': self.rank}
        if self.rank == 0:
            print(json.dumps(line), flush=True)
        print(f'bench.py: {why}', file=sys.stderr, flush=True)
        os._exit(code)


def self_launch(a):
    """``python bench.py --gpus N`` (no launcher around it): re-exec through torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL over xGMI needs it on this host driver)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def scale_sweep(a):
    """N = 1, 2, 4, 8 weak + strong from ONE command (one lease of a node yields the whole curve; the driver's own SCALE run calls
    `bench.py --gpus N` per N - this is the same job list, self-launched).  Each job is a child `bench.py --gpus N ...`; its JSON
    line is relayed verbatim.  The summary carries values only - efficiencies are the reader's to compute."""
    n_dev = torch.cuda.device_count()
    sizes = [n for n in (1, 2, 4, 8) if n <= n_dev]
    base = [sys.executable, os.path.abspath(__file__), '--steps', str(a.steps), '--warmup', str(a.warmup), '--channels', str(a.channels),
            '--no-cpu-baseline', '--no-strong-probe', '--no-other-workloads', '--no-companions', '--dense', a.dense, '--workload', a.workload]
    summary = {'devices_visible': n_dev, 'weak': {}, 'strong_global_batch_32': {}}
    for n in sizes:
        for mode, extra in (('weak', ['--batch', str(a.batch)] if a.batch else []), ('strong_global_batch_32', ['--global-batch', '32'])):
            if mode != 'weak' and (32 % n or a.workload != 'l'):
                continue
            try:
                r_ = subprocess.run(base + ['--gpus', str(n)] + extra, capture_output=True, text=True, timeout=900)
                line = [l for l in r_.stdout.splitlines() if l.startswith('{')][-1]
                d_ = json.loads(line)
                print(line, flush=True)
                summary[mode][str(n)] = {'value': d_.get('value'), 'ms_per_step': d_.get('ms_per_step'), 'verified': d_.get('verified'),
                                         'distinct_devices': (d_.get('config', {}).get('ranks') or {}).get('distinct_devices'),
                                         'error': d_.get('error')}
            except Exception as e:
                summary[mode][str(n)] = {'error': repr(e)[:300]}
    print(json.dumps({'scale_sweep': summary, 'metric': METRIC, 'unit': 'frames/s'}), flush=True)


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        return os.cpu_count()


def _time_oracle(fn, budget_s, full, min_timed=5):
    """Frames/s of ``fn`` (one frame): median, quartiles and extremes of the timed frames.  Full protocol: 5 warm-up + 20 timed
    (benchmark.py:62-91); default: the same protocol bounded to ``budget_s`` seconds of CPU work, never fewer than 1 warm-up +
    ``min_timed`` timed frames (round 4 timed 3 and moved 80 % between boxes)."""
    t0 = time.perf_counter()
    fn()
    first = time.perf_counter() - t0
    if full:
        n_warm, n = 4, 20
    else:
        n = max(min_timed, min(20, int(budget_s / max(first, 1e-3)) - 1))
        n_warm = 4 if (n + 5) * first <= budget_s else 0
    for _ in range(n_warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    q = statistics.quantiles(ts, n=4) if len(ts) >= 4 else [min(ts), statistics.median(ts), max(ts)]
    return {'fps': 1.0 / statistics.median(ts), 'warm': n_warm + 1, 'timed': n, 'min_max': (1.0 / max(ts), 1.0 / min(ts)),
            'iqr': (1.0 / q[2], 1.0 / q[0])}


def cpu_baseline(C, budget_s, full):
    """The CPU oracle (a port of the reference algorithm, oracle/ff3d_oracle.py) timed on the host's physical cores on a
    bounded sample of the same workload, SURVEY.md §8(d): C2 = the benchmarked head (BASELINE configs[1]) and C1 =
    DeformFormer3D_L (configs[0]), both at batch 1, forward + get_bboxes; median + inter-quartile range of >= 5 timed frames."""
    from oracle import ff3d_oracle as O
    from focalformer3d_amd.synthetic import (build_head_from_cfg, deformformer3d_l_head_cfg, focalformer3d_l_head_cfg,
                                             stage_features)
    cores = physical_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(cores)
    used = torch.get_num_threads()                      # what torch actually runs with (it may clamp the request)
    res = {}
    try:
        # C1 is ~3 x cheaper per frame than C2: a third of the budget keeps the default run inside a few minutes
        for tag, hc, n_maps, share in (('C2', focalformer3d_l_head_cfg(C=C, grid=180, num_proposals=200, stages=3, decoder_stages=2), 3, 1.0),
                                       ('C1', deformformer3d_l_head_cfg(C=C, grid=180, num_proposals=200), 1, 1.0 / 3.0)):
            sd = {k: v.detach().cpu() for k, v in build_head_from_cfg(hc, seed=0).state_dict().items()}
            ocfg = O.head_config(
                num_proposals=hc['num_proposals'], hidden_channel=C, num_classes=hc['num_classes'],
                num_decoder_layers=hc['num_decoder_layers'], nms_kernel_size=3, multiscale=True,
                multistage_heatmap=hc['multistage_heatmap'] or 0, reuse_first_heatmap=hc['reuse_first_heatmap'],
                extra_feat=hc['extra_feat'], bevpos=True, input_img=False, iterbev_wo_img=True,
                roi_feats=hc['roi_feats'], roi_expand_ratio=hc['roi_expand_ratio'], roi_based_reg=hc['roi_based_reg'],
                common_heads=hc['common_heads'], voxel_size=tuple(hc['bbox_coder']['voxel_size']))
            f = stage_features(1, C, 180, n_maps, seed=123)
            inputs = f if hc['multistage_heatmap'] else [f[0], f[1][0]]

            def one():
                with torch.no_grad():
                    out, aux = O.focal_decoder_forward(sd, ocfg, inputs)
                    O.focal_decoder_get_bboxes(out, aux, ocfg)
            res[tag] = _time_oracle(one, budget_s * share, full)
    finally:
        torch.set_num_threads(old)
    c2, c1 = res['C2'], res['C1']
    r4 = lambda vs: [round(v, 4) for v in vs]                                              # noqa: E731
    return dict(value=round(c2['fps'], 4), unit='frames/s', cores=cores, torch_threads=used, kind='port',
                protocol='full (5 warm-up + 20 timed)' if full else f'bounded (~{budget_s:.0f} s of CPU work, >= 5 timed frames)',
                timed_frames=c2['timed'], iqr=r4(c2['iqr']), spread_min_max=r4(c2['min_max']),
                sample=f'C2 = this workload at batch 1 (oracle/ff3d_oracle.py forward + get_bboxes, fp32, torch CPU, '
                       f'{used} threads; {cores} physical cores): median of {c2["timed"]} timed frames after {c2["warm"]} warm-up frames'
                       + ('' if full else f' (protocol of tools/analysis_tools/benchmark.py:62-91 bounded to ~{budget_s:.0f} s of CPU work)'),
                c1_deformformer3d_l={'value': round(c1['fps'], 4), 'unit': 'frames/s', 'timed_frames': c1['timed'],
                                     'iqr': r4(c1['iqr']), 'spread_min_max': r4(c1['min_max']),
                                     'sample': f'C1 = DeformFormer3D_L head (BASELINE configs[0]: 1 stage, 200 queries, 1 decoder '
                                               f'stage, no RoI) at batch 1, 180x180x{C}: median of {c1["timed"]} timed frames after '
                                               f'{c1["warm"]} warm-up'})


def other_workloads(a):
    """BASELINE configs[2] (`--workload lc`) and configs[4] (`--workload waymo`) next to the headline: each in a child process
    (own context, ~10 timed steps), reduced to frames/s, ms/step, dtype and the launch group that carries most of its step.
    The headline never depends on them: a failure is recorded as such."""
    res = {}
    for wl in ('lc', 'waymo'):
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', wl, '--steps', '10', '--warmup', '3', '--channels',
               str(a.channels), '--no-cpu-baseline', '--no-strong-probe', '--no-other-workloads', '--dense', a.dense]
        try:
            t0 = time.perf_counter()
            r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d_ = json.loads([l for l in r_.stdout.splitlines() if l.startswith('{')][-1])
            res[wl] = {'config': 'BASELINE.json configs[2]' if wl == 'lc' else 'BASELINE.json configs[4]',
                       'workload': d_['config']['workload'], 'value': d_['value'], 'unit': d_['unit'],
                       'ms_per_step': d_['ms_per_step'], 'frames_per_step': d_['config']['frames_per_gpu_per_step'],
                       'steps': d_['steps'], 'dtype': d_['dtype'], 'execution': d_['config']['execution'],
                       'top_kernel': d_.get('top_kernel'), 'child_wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:
            res[wl] = {'error': repr(e)[:300]}
    return res


def companions(a, static_value):
    """Two companions of the default line, each in a child process (own context; the headline never depends on them):
      * ``fresh_inputs``: the same step with 4 distinct batches rotated through the slots inside the timed region (--fresh-inputs 4);
      * ``latency_b1_ms``: one frame per call, one call at a time - the reference's own benchmark protocol (--latency-b1)."""
    base = [sys.executable, os.path.abspath(__file__), '--channels', str(a.channels), '--no-cpu-baseline', '--no-strong-probe',
            '--no-other-workloads', '--no-companions', '--dense', a.dense, '--gemm-dtype', a.gemm_dtype]
    fresh, lat = None, None
    try:
        r_ = subprocess.run(base + ['--fresh-inputs', '4', '--batch', str(a.batch), '--steps', str(a.steps), '--warmup', str(a.warmup),
                                    '--slots', str(a.slots)], capture_output=True, text=True, timeout=240)
        d_ = json.loads([l for l in r_.stdout.splitlines() if l.startswith('{')][-1])
        fresh = {'value': d_['value'], 'unit': d_['unit'], 'ms_per_step': d_['ms_per_step'], 'steps': d_['steps'],
                 'vs_static_replay': round(d_['value'] / static_value, 4), 'verified': d_['verified'],
                 'inputs': d_['config']['inputs'], 'execution': d_['config']['execution']}
    except Exception as e:
        fresh = {'error': repr(e)[:300]}
    try:
        r_ = subprocess.run(base + ['--latency-b1'], capture_output=True, text=True, timeout=240)
        lat = json.loads([l for l in r_.stdout.splitlines() if l.startswith('{')][-1])['latency_b1_ms']
    except Exception as e:
        lat = {'error': repr(e)[:300]}
    return fresh, lat


def gather_first_companion(a, default_value):
    """The opt-in value mode next to the default: the same command with --value-mode gather_first in a child process."""
    cmd = [sys.executable, os.path.abspath(__file__), '--channels', str(a.channels), '--no-cpu-baseline', '--no-strong-probe',
           '--no-other-workloads', '--no-companions', '--dense', a.dense, '--gemm-dtype', a.gemm_dtype, '--value-mode', 'gather_first',
           '--batch', str(a.batch), '--steps', str(a.steps), '--warmup', str(a.warmup), '--slots', str(a.slots)]
    try:
        r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        d_ = json.loads([l for l in r_.stdout.splitlines() if l.startswith('{')][-1])
        rf = d_['roofline']
        return {'value': d_['value'], 'unit': d_['unit'], 'ms_per_step': d_['ms_per_step'], 'steps': d_['steps'],
                'vs_default_mode': round(d_['value'] / default_value, 4), 'verified': d_['verified'],
                'gather': {'kernel': rf['kernel'], 'avg_launch_ms': rf['avg_launch_ms'], 'unique_bytes_per_launch': rf['algorithmic_bytes_per_launch'],
                           'hbm_frac_by_unique_bytes': rf['frac']},
                'note': 'opt-in (FocalDecoder.set_value_mode): value_proj moved behind the gather - the two value GEMMs of a step are gone, '
                        'the gather requests 8 x the bytes from L2 / MALL; NOT the default: north_star names the HBM-bound gather of '
                        'projected values, which the headline and `roofline` measure'}
    except Exception as e:
        return {'error': repr(e)[:300]}


def latency_b1(a, head, inputs, more_inputs, metas, dev, wd):
    """tools/analysis_tools/benchmark.py:62-91's protocol on the head path: samples one at a time (batch 1), the host waits for each
    result before it submits the next, 5 warm-up + 20 timed calls, mean.  Three forms of the same call: eager launches; one captured
    graph replayed over a frame COPIED into its input buffers per call (a caller with a fresh frame); device-side duration of that
    replay from HIP events on the replaying stream."""
    from focalformer3d_amd import dist as fdist
    from focalformer3d_amd.runtime import PipelinedHead
    wd.stage('latency, batch 1')
    frames = [inputs] + list(more_inputs or [])
    if len(frames) < 2:
        frames = frames * 2
    n_warm, n_timed = 5, 20

    def protocol(call):
        ts = []
        for i in range(n_warm + n_timed):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            call(i)
            torch.cuda.synchronize()
            if i >= n_warm:
                ts.append((time.perf_counter() - t0) * 1e3)
        return ts
    eager = protocol(lambda i: fdist.pack_detections(*head.get_bboxes_padded(head(frames[i % len(frames)], None, metas))))
    pipe = PipelinedHead(head, [frames[0]], slots=1, pack=True)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dev_ms = []

    def replay(i):
        s_ = pipe.submit(frames[i % len(frames)])           # copies the frame into the slot's buffers, replays
        pipe.wait(s_)
    graph = protocol(replay)

    def replay_events(i):
        with torch.cuda.stream(pipe.streams[0]):
            st.record()
        pipe.submit()
        with torch.cuda.stream(pipe.streams[0]):
            en.record()
        en.synchronize()
        if i >= n_warm:
            dev_ms.append(st.elapsed_time(en))
    protocol(replay_events)
    want = pipe.eager_reference(0)
    ok = bool(torch.equal(pipe.packed[0], want))
    rec = {'protocol': "tools/analysis_tools/benchmark.py:62-91: one frame per call, one call in flight, torch.cuda.synchronize() around "
                       'every call, 5 warm-up + 20 timed calls',
           'frames_per_call': head.num_frames(inputs) if hasattr(head, 'num_frames') else int(inputs[0].shape[0]),
           'graph_replay': {'mean': round(statistics.mean(graph), 4), 'median': round(statistics.median(graph), 4),
                            'min': round(min(graph), 4), 'max': round(max(graph), 4),
                            'what': 'copy of a fresh frame into the input buffers + ONE hipGraph replay (head + get_bboxes + packing), host-timed'},
           'graph_replay_device': {'mean': round(statistics.mean(dev_ms), 4), 'what': 'the replay alone, HIP events on its stream'},
           'eager': {'mean': round(statistics.mean(eager), 4), 'median': round(statistics.median(eager), 4),
                     'what': 'the same call as ~120 eager launches, host-timed'},
           'frames_per_s_at_batch_1': round(1e3 / statistics.mean(graph), 2),
           'verified': {'bit_identical': ok, 'against': 'eager launches over the frame of the last replay'}}
    wd.finish()
    print(json.dumps({'latency_b1_ms': rec, 'workload_key': a.workload, 'channels': a.channels}), flush=True)


def pmc_entry(name, B, C):
    """PMC record (rocprofv3 counter passes, profiles/<name>.json) for this exact configuration, or None.  These are
    labelled evidence measured at the commit the file names, not live measurements of this run."""
    try:
        doc = json.load(open(os.path.join(ROOT, 'profiles', name + '.json')))
        for e in doc['entries']:
            if e['batch'] == B and e['channels'] == C:
                return dict(e, measured_at=doc.get('measured_at', 'round 1'))
    except Exception:
        pass
    return None


def preflight_collective():
    """Child process of collective_capturable(): can THIS stack do what the pipelined step does with a process group of the same
    shape as the parent's - capture an RCCL all-gather (thread-local capture mode) behind some device work in TWO hipGraphs, each with
    its own communicator and stream, and replay them round-robin so that the graphs (and their collectives) overlap?  Exit code
    0 = yes.  A capture that fails leaves its process unusable for further GPU work (`operation failed due to a previous error
    during capture`, session k), and a collective that deadlocks never returns - hence a disposable process per rank, with its own
    rendezvous, under the parent's timeout."""
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    slots, rows = 2, 4 * 201
    groups = [dist.new_group(backend='nccl') for _ in range(slots)]
    streams = [torch.cuda.Stream() for _ in range(slots)]
    work = [torch.zeros(32 << 20, device=dev) for _ in range(slots)]               # 128 MB each: ~0.1 ms per pass over it
    tick = [torch.zeros(1, device=dev) for _ in range(slots)]
    x = [torch.zeros(rows, 11, device=dev) for _ in range(slots)]
    out = [torch.empty(world * rows, 11, device=dev) for _ in range(slots)]

    def run(s):
        tick[s].add_(1.0)
        for _ in range(4):                                                         # the batch's kernels ...
            work[s].add_(tick[s])
        x[s].copy_((work[s][:rows * 11] * 0 + tick[s] * (100 * s + rank + 1)).view(rows, 11))
        dist.all_gather_into_tensor(out[s], x[s], group=groups[s])                 # ... then its all-gather, in the same graph

    graphs = []
    for s in range(slots):
        with torch.cuda.stream(streams[s]):
            run(s)                                                                 # lazy communicator init outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            run(s)
        graphs.append(g)
    torch.cuda.synchronize()
    done = [torch.cuda.Event() for _ in range(slots)]
    n = 24
    for i in range(n * slots):
        s = i % slots
        with torch.cuda.stream(streams[s]):
            graphs[s].replay()
            done[s].record()
    ok = True
    for s in range(slots):
        done[s].synchronize()
        ticks = float(n + 1)                                                       # warm-up pass + n replays (a capture runs nothing)
        want = torch.cat([torch.full((rows, 11), ticks * (100 * s + r + 1)) for r in range(world)])
        ok = ok and torch.equal(out[s].cpu(), want)
    os._exit(0 if ok else 3)


def collective_capturable(world, rank, local_rank, dev, backend):
    """Decide - identically on every rank - whether the step's all-gather goes inside the captured graphs: one disposable child
    per rank tries it (preflight_collective) under its own rendezvous; any failure, timeout or a non-RCCL backend means eager
    launches + the side-stream gather."""
    if backend != 'nccl':
        return False
    import torch.distributed as dist
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        free_port = s_.getsockname()[1]
    base = int(os.environ.get('MASTER_PORT', '29500'))
    port = free_port if world == 1 else (base + 101 if base + 101 < 65000 else base - 101)     # the children's own rendezvous
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank),
               MASTER_ADDR=os.environ.get('MASTER_ADDR', '127.0.0.1'), MASTER_PORT=str(port))
    for k_ in [k_ for k_ in env if k_.startswith('TORCHELASTIC_') or k_ == 'FF3D_BENCH_FORCE_DIST']:
        env.pop(k_)          # (under torchrun rank 0 would otherwise look for the launcher's store on the child's port)
    try:
        ok = subprocess.run([sys.executable, os.path.abspath(__file__), '--preflight-collective'], env=env, capture_output=True,
                            timeout=180).returncode == 0
    except Exception:
        ok = False
    t = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


class Runner:
    """One rank's decoder loop over fixed batches.  Two execution forms:
      * eager launches on one stream + dist.AsyncDetectionGather (pack + RCCL all-gather on a side stream);
      * runtime.PipelinedHead: ``slots`` captured hipGraphs (head + get_bboxes_padded + pack + - when there is a process group -
        the RCCL all-gather, captured inside the graph) replayed round-robin on ``slots`` streams, so that consecutive batches
        overlap on the GPU.  A step is still one pass of the path over one batch."""

    def __init__(self, head, inputs, metas, use_graph, dev, neck=None, neck_inputs=None, slots=2, more_inputs=None,
                 collective=False, pool=None):
        from focalformer3d_amd import dist as fdist
        self.head, self.inputs, self.metas = head, inputs, metas
        self.neck, self.neck_inputs = neck, neck_inputs          # workload lc: the fusion neck produces the head's inputs
        self.pipe = None
        self.slots = 1
        # --fresh-inputs: distinct batches handed to the slots in turn; the producer (a device copy) runs on its own stream
        self.pool, self.fed, self.producer = pool, 0, (torch.cuda.Stream(device=dev) if pool else None)
        B = len(metas)
        self.gather = fdist.AsyncDetectionGather(B, 200, dev, force_collective=os.environ.get('FF3D_BENCH_FORCE_DIST') == '1')
        if use_graph:
            from focalformer3d_amd.runtime import PipelinedHead
            groups = None
            if collective:                                 # one communicator per slot (their all-gathers may overlap in time)
                import torch.distributed as dist
                groups = [dist.new_group(backend=dist.get_backend()) for _ in range(slots)]    # (nccl = RCCL; gloo only in the one-GPU rehearsal)
            examples = [inputs] + list(more_inputs or [])[:slots - 1]
            while len(examples) < slots:
                examples.append(inputs)
            self.pipe = PipelinedHead(head, examples, slots=slots, pack=True, collective=groups)
            self.slots = slots

    def warm_replays(self, n=None):
        """Replays before the timed region (graph upload, code-object loading, the slots settling into their staggered steady
        state).  (timed() still uses a host-side barrier before the contract's synchronise - a leftover of the rounds in which an
        eager launch between replays and a synchronise faulted the GPU; harmless, and it keeps the timed region free of launches.)"""
        if self.pipe is not None:
            n = WARM_REPLAYS if n is None else n
            for _ in range(n * self.slots):
                self.pipe.submit()
            self.pipe.wait()

    def step(self, warm=False):
        # warm: run the step eagerly even in graph mode (the same kernels, launched one by one)
        if self.pipe is not None and not warm and self.pool:
            # fresh frames: the producer writes batch i into the slot's input buffers IN PLACE (ordered after the slot's previous
            # replay by an event, before its next one by a stream wait), then the slot replays - no copy besides the producer's own
            with torch.cuda.stream(self.producer):
                s, bufs = self.pipe.begin_fill()
                for dst, src in zip(_flat(bufs), _flat(self.pool[self.fed % len(self.pool)])):
                    dst.copy_(src, non_blocking=True)
                self.fed += 1
                self.pipe.submit(filled=True)
            return self.pipe.dets[s][3]
        if self.pipe is not None and not warm:
            s = self.pipe.submit()                                    # replay: inputs already in the slot's static buffers
            return self.pipe.dets[s][3]
        inputs = self.inputs if self.neck is None else self.neck(*self.neck_inputs, self.metas)[1]
        dets = self.head.get_bboxes_padded(self.head(inputs, None, self.metas))
        self.gather.submit(*dets)                                     # pack (1 launch) + RCCL all-gather on the side stream
        return dets[3]

    def verify(self, rank=0):
        """After the LAST replay: every slot's packed detections (this rank's rows of the gathered record when there is a
        collective) against eager launches over that slot's own inputs (PipelinedHead.eager_reference) - the timed replays must
        have produced, bit for bit, what the parity-tested eager form produces (tests/test_bench_shape_gpu.py checks that form
        against the oracle at this size; tests/test_bench_shape_gpu.py::test_pipelined_replays_* does both in one test)."""
        if self.pipe is None:
            rec = {'slots': 0, 'note': 'eager launches: the timed steps are themselves the parity-tested form'}
            if self.gather.collective:
                # the exchange is verified even without a graph: this rank's rows of the gathered record are its
                # own packed detections, and every rank holds the same gathered record
                g_ = self.gather
                gathered, own = g_.result(), g_.packed[g_.i]
                rec.update(gathered_record_check(gathered, own, rank))
            return rec
        p = self.pipe
        B = p.packed[0].shape[0]
        same, worst = True, 0.0
        for s in range(p.slots):
            p.wait(s)
            got = p.packed[s] if p.gathered[s] is None else p.gathered[s][rank * B:(rank + 1) * B]
            want = p.eager_reference(s)
            if not torch.equal(got, want):
                same = False
                worst = max(worst, float((got - want).abs().nan_to_num(nan=float('inf')).max()))
        fresh = None
        if self.pool:
            # the rotation really happened: the slot of the LAST step holds the last batch fed, the one before it the previous one
            fresh = True
            for back in range(min(p.slots, self.fed)):
                s_ = (p.i - back) % p.slots
                want_b = self.pool[(self.fed - 1 - back) % len(self.pool)]
                fresh = fresh and all(torch.equal(a_, b_) for a_, b_ in zip(_flat(p.input_buffers(s_)), _flat(want_b)))
        rec = {'slots': p.slots, 'frames_compared': p.slots * B, 'bit_identical': same,
               'against': "eager launches over each slot's own inputs after the last timed replay (runtime.PipelinedHead.eager_reference)"}
        if not same:
            rec['max_abs_diff'] = worst
        if fresh is not None:
            rec['slots_hold_the_last_batches_fed'] = fresh
        if p.gathered[p.i] is not None:
            rec.update(gathered_record_check(p.gathered[p.i], p.packed[p.i], rank))
        return rec

    def finish(self, replayed=True):
        if self.pipe is not None and replayed:
            self.pipe.wait()
            return self.pipe.result()
        return self.gather.result()


def gathered_record_check(gathered, own, rank):
    """What the all-gather must have produced (tools/test.py:229-233's collect_results_gpu counterpart): rows [rank * B, (rank + 1) * B)
    of the gathered record are bit for bit this rank's own packed detections, and the whole record is the same on every rank (a 64-bit
    checksum of its bit pattern, MIN == MAX over the ranks).  Collective calls: every rank must get here."""
    import torch.distributed as dist
    B = own.shape[0]
    mine = bool(torch.equal(gathered[rank * B:(rank + 1) * B], own))
    rec = {'gathered_rows_of_this_rank_are_its_own': mine, 'gathered_frames': int(gathered.shape[0])}
    if dist.is_available() and dist.is_initialized():
        bits = gathered.contiguous().view(torch.int32).to(torch.int64)
        weights = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64).view_as(bits) % 8191 + 1
        chk = (bits * weights).sum().view(1)               # position-weighted: a permutation of frames changes it
        lo, hi, ok = chk.clone(), chk.clone(), torch.tensor([1 if mine else 0], device=chk.device if chk.is_cuda else None)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        rec.update({'gathered_record_identical_on_every_rank': bool(lo.item() == hi.item()),
                    'gathered_rows_of_this_rank_are_its_own': bool(ok.item()), 'gathered_checksum': int(chk.item()),
                    'ranks_in_the_check': dist.get_world_size()})
    return rec


def _flat(inputs):
    """[map, [maps...]] / [map, map] -> flat list of tensors."""
    return [inputs[0]] + (list(inputs[1]) if isinstance(inputs[1], (list, tuple)) else [inputs[1]])


_HOST_GROUP = {}


def host_barrier(world):
    """Barrier without a GPU launch (gloo): between graph replays and a device synchronise nothing may be launched eagerly."""
    if world > 1:
        import torch.distributed as dist
        if 'g' not in _HOST_GROUP:
            _HOST_GROUP['g'] = dist.new_group(backend='gloo')
        dist.barrier(group=_HOST_GROUP['g'])


def timed(runner, steps, warmup, world, dev):
    def sync_all():
        host_barrier(world)
        torch.cuda.synchronize()
    for _ in range(warmup):
        runner.step(warm=True)
    runner.finish(replayed=False)
    if runner.pipe is not None:
        torch.cuda.synchronize()
        runner.warm_replays()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        count = runner.step()
    packed = runner.finish()
    counts = count.tolist()          # the host reads the detection counts of the last batch (get_bboxes' compaction)
    sync_all()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.empty(world, device=dev, dtype=torch.float64)
        torch.distributed.all_gather_into_tensor(every, t)
        per_rank = every.tolist()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, counts, packed, per_rank


def _pci_address(props):
    from focalformer3d_amd import dist as fdist
    return fdist.gpu_pci_address(props=props)


def rank_records(world, dev, per_rank_s, steps, affinity=None):
    """What a SCALE run needs to verify that N ranks on N different devices took part: per rank its device (index, name, UUID,
    PCI bus id), host pid and its own wall time of the timed region; + the size of the RCCL group the all-gather ran in."""
    pr = torch.cuda.get_device_properties(dev)
    mine = {'rank': int(os.environ.get('RANK', 0)), 'pid': os.getpid(), 'device_index': dev.index, 'device_name': pr.name,
            'device_uuid': str(getattr(pr, 'uuid', '')), 'pci_bus_id': getattr(pr, 'pci_bus_id', None),
            'pci_address': _pci_address(pr),
            'host_cpus': affinity or {'cpus': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
                                      'pinned': False}}
    recs = [mine]
    if world > 1:
        recs = [None] * world
        torch.distributed.all_gather_object(recs, mine)
    for r_, t_ in zip(recs, per_rank_s):
        r_['ms_per_step'] = round(t_ / steps * 1e3, 4)
    initialised = torch.distributed.is_available() and torch.distributed.is_initialized()
    return {'rccl_world': torch.distributed.get_world_size() if initialised else 0,
            'backend': torch.distributed.get_backend() if initialised else None,
            'distinct_devices': len({r_['device_uuid'] or (r_['pid'], r_['device_index']) for r_ in recs}), 'ranks': recs}


def main():
    a = parse()
    if a.preflight_collective:
        preflight_collective()
    if a.scale_sweep and 'WORLD_SIZE' not in os.environ:
        scale_sweep(a)
        return
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(a)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus != world:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    wd = Watchdog(a.watchdog, rank, world)
    wd.stage('set-up: process group, host pinning, head, capture')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP decoder path has no CPU fallback')
    backend = os.environ.get('FF3D_BENCH_BACKEND', 'nccl')
    if world > torch.cuda.device_count() and backend == 'nccl':
        raise SystemExit(f'--gpus {world} but only {torch.cuda.device_count()} device(s) visible')
    local_rank %= torch.cuda.device_count()                     # (only differs in the single-GPU rehearsal below)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = world == 1 and os.environ.get('FF3D_BENCH_FORCE_DIST') == '1'
    if force_dist:
        # rehearsal on ONE GPU of everything the N > 1 path does: a 1-rank RCCL group, the all-gather on the side stream
        import torch.distributed as dist
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  FF3D_BENCH_BACKEND=gloo exists only to rehearse the multi-rank control flow with
        # several ranks on ONE GPU (RCCL refuses duplicate devices); it is never used for reported numbers.
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from focalformer3d_amd import dist as fdist, ops
    affinity = None
    if world > 1 and not a.no_pin:
        # one process per GPU: every rank's host threads on its own cores, next to its GPU's NUMA node where sysfs knows it
        import torch.distributed as dist
        buses = [None] * world
        dist.all_gather_object(buses, fdist.gpu_pci_address(dev))
        # (LOCAL_RANK as launched: in the one-GPU rehearsal every rank maps to device 0 but still gets its own cores)
        aff = fdist.pin_host_threads(int(os.environ.get('LOCAL_RANK', rank)), int(os.environ.get('LOCAL_WORLD_SIZE', world)), dev, buses)
        affinity = dict(aff, pinned=bool(aff))
    from focalformer3d_amd.synthetic import (build_head_from_cfg, build_neck_from_cfg, focalformer3d_l_head_cfg,
                                             focalformer3d_lc_cfgs, lc_inputs, stage_features, waymo_shape_head_cfg)

    C = a.channels
    if a.latency_b1:
        if world > 1:
            raise SystemExit('--latency-b1 is a one-GPU measurement')
        a.batch, a.global_batch = 1, 0
    if not a.batch:
        a.batch = 32 if a.workload == 'l' else 8
    if a.gemm_dtype is None:
        a.gemm_dtype = 'bf16' if a.workload == 'waymo' else 'f32'
    strong = a.global_batch > 0
    if strong:
        lo, hi = fdist.shard_range(a.global_batch, rank, world)
        B, total = hi - lo, a.global_batch
        if a.global_batch % world:
            raise SystemExit('--global-batch must be divisible by --gpus (fixed-shape all-gather)')
    else:
        B, total = a.batch, a.batch * world
    neck = neck_inputs = None
    metas = [{'box_type_3d': lambda t, box_dim=9: t}] * B
    if a.workload == 'l':
        cfg = focalformer3d_l_head_cfg(C=C, grid=180, num_proposals=200, stages=3, decoder_stages=2)
        inputs = stage_features(B, C, 180, 3, seed=1 + rank, device=dev)
        what = (f'FocalFormer3D_L head: 3 HIP stages x 200 queries (Nq=600), 2 decoder stages x 3 layers, RoI 7x7, '
                f'180x180x{C} BEV, K=10; FocalDecoder.forward + get_bboxes, features resident in HBM (BASELINE.json configs[1]'
                + ('; sharded as configs[3])' if strong else ')'))
    elif a.workload == 'waymo':
        cfg = waymo_shape_head_cfg(C=C)
        inputs = stage_features(B, C, 468, 4, seed=1 + rank, device=dev)
        what = (f'Waymo-shape head: 468x468x{C} BEV (Nv = 287 469), 1000 queries = 4 HIP stages x 250, K=3, 2 decoder stages x 3 '
                f'layers, RoI 7x7, {a.gemm_dtype} decoder GEMMs; FocalDecoder.forward + get_bboxes (BASELINE.json configs[4])')
    else:
        ncfg, cfg = focalformer3d_lc_cfgs(C=C)
        neck = build_neck_from_cfg(ncfg, seed=1, device=dev)
        img, pts, lc_metas, _ = lc_inputs(B, seed=1 + rank, device=dev)
        metas = [dict(m, box_type_3d=metas[0]['box_type_3d']) for m in lc_metas]
        neck_inputs, inputs = (img, pts), None
        what = (f'FocalFormer3D_LC_Proj chain: 6 x 256 x 232 x 400 camera maps + 180x180x512 LiDAR BEV -> FocalEncoder '
                f"('bevfusion': I2P camera-projection sampler, 9x9 local attention; {C} channels) -> head (3 HIP stages x 200 "
                f'queries, 2 decoder stages, RoI 7x7) -> get_bboxes (BASELINE.json configs[2])')
    head = build_head_from_cfg(cfg, seed=0, device=dev)
    if a.gemm_dtype == 'bf16':
        head.set_gemm_dtype(torch.bfloat16)
    if a.dense != 'default':
        head.set_dense_mode(a.dense)
    if a.value_mode != 'project_first':
        if a.workload == 'lc':
            raise SystemExit("--value-mode gather_first: workloads l / waymo")
        head.set_value_mode(a.value_mode)
    # Graph mode (round 4): every step is one replay of a captured graph that contains the whole step INCLUDING the RCCL
    # all-gather (captured in thread-local capture mode, one communicator per slot; profiles/r04_b_*): one launch per step for the
    # host, and the exchange overlaps the other slots' work.  FF3D_BENCH_DIST_MODE=eager restores eager launches + the side-stream
    # gather for N > 1.
    collective = world > 1 or force_dist
    if neck is not None:
        # round 5: neck + head as one capturable unit (runtime.NeckAndHead): the lc step is graph-replayed like the others (rounds 1-4
        # ran the neck eagerly, 400 dispatches per step); its inputs are [camera maps, [LiDAR BEV map]]
        from focalformer3d_amd.runtime import NeckAndHead
        base_head, head = head, NeckAndHead(neck, head, metas).eval()
        inputs, neck, neck_inputs = [neck_inputs[0], [neck_inputs[1]]], None, None
    use_graph = a.graph != 'off'
    if collective and use_graph:
        if os.environ.get('FF3D_BENCH_DIST_MODE') == 'eager':
            use_graph = False
        elif not collective_capturable(world, rank, local_rank, dev, backend if world > 1 else 'nccl'):
            # (decided in disposable child processes: a failed capture cannot be recovered from in-process, session k)
            if rank == 0:
                print('bench.py: an RCCL all-gather cannot be captured in a hipGraph on this stack (preflight failed); eager launches '
                      '+ side-stream gather', file=sys.stderr)
            use_graph = False
    # auto: four batches in flight while a batch is small (<= 8 frames of 180 x 180), two beyond (every slot owns a full set of
    # activations: at 468 x 468 x 8 frames that is ~30 GB per slot)
    grid_cells = {'l': 180 * 180, 'waymo': 468 * 468, 'lc': 180 * 180}[a.workload]
    slots = a.slots if a.slots > 0 else (4 if B * grid_cells <= 8 * 180 * 180 else 2)
    if a.slots <= 0 and a.workload == 'lc':
        slots = 2                            # (every slot owns the camera maps, their pairs and the conv output: ~15 GB at 8 frames)
    # Overlapping replays are only used while every dense launch of the step is one of this package's kernels (a vendor kernel that
    # spin-waits on workgroups of its own grid deadlocks beside another graph's kernels: profiles/r04_d_waymo_two_slots_hang.txt).
    # PipelinedHead decides that on what ran in its warm-up (ops.note_vendor) and raises; then: one graph.
    if a.slots <= 0 and head.dense_mode != 'f16x3':
        slots = 1
    more_inputs = None
    if use_graph and slots > 1 and a.workload in ('l', 'waymo'):          # every slot decodes its own frames
        grid, n_maps = (180, 3) if a.workload == 'l' else (468, 4)
        more_inputs = [stage_features(B, C, grid, n_maps, seed=1000 * i + 1 + rank, device=dev) for i in range(1, slots)]
    elif use_graph and slots > 1 and a.workload == 'lc':                  # other frames, the same camera rig (metas are part of the unit)
        more_inputs = []
        for i in range(1, slots):
            img_i, pts_i, _, _ = lc_inputs(B, seed=1000 * i + 1 + rank, device=dev)
            more_inputs.append([img_i, [pts_i]])
    if a.latency_b1:
        latency_b1(a, head, inputs, more_inputs, metas, dev, wd)
        return
    pool = None
    if a.fresh_inputs:
        if not use_graph:
            raise SystemExit('--fresh-inputs is a property of the graph-replay form (eager launches read whatever tensors they are given)')
        n_pool = max(a.fresh_inputs, 2 * slots)
        if a.workload == 'lc':
            pool = [inputs] + [[t_[0], [t_[1]]] for t_ in (lc_inputs(B, seed=2000 + 10 * i + rank, device=dev)[:2] for i in range(1, n_pool))]
        else:
            grid, n_maps = (180, 3) if a.workload == 'l' else (468, 4)
            pool = [inputs] + [stage_features(B, C, grid, n_maps, seed=2000 + 10 * i + rank, device=dev) for i in range(1, n_pool)]
    try:
        runner = Runner(head, inputs, metas, use_graph, dev, neck, neck_inputs, slots=slots, more_inputs=more_inputs,
                        collective=collective, pool=pool)
    except ValueError as e:
        if 'vendor' not in str(e) or a.slots > 0:
            raise
        print(f'bench.py: {e}', file=sys.stderr)
        runner = Runner(head, inputs, metas, use_graph, dev, neck, neck_inputs, slots=1, collective=collective)
    wd.stage('warm-up steps (eager launches)')
    for _ in range(a.warmup):
        runner.step(warm=True)
    # Live kernel timing: the MSDA gather (the roofline kernel) is bracketed by HIP events INSIDE the timed region; the ~60 dense
    # launches of a step are timed in a short eager pass right after it (same tensors, same stream) - two event records per
    # launch inside the timed region cost the host-bound small-batch steps ~0.4 ms.
    ops.MSDA_EVENTS, ops.DENSE_EVENTS = [], None
    gather_first = a.value_mode == 'gather_first'
    if gather_first:
        ops.MSDA_EVENTS, ops.GATHER_EVENTS = None, []
    wd.stage('timed region: warm replays, barrier, K steps, barrier')
    elapsed, counts, packed, per_rank_s = timed(runner, a.steps, 0, world, dev)
    wd.stage('rank records + verification of the replays against eager launches')
    ranks = rank_records(world, dev, per_rank_s, a.steps, affinity)
    if packed.shape[0] != total:
        raise SystemExit(f'bench.py: {packed.shape[0]} frames in the gathered detections, expected {total}')
    # The headline is self-verifying: the replays of the timed region against eager launches, every slot, every rank
    verified = runner.verify(rank)
    if world > 1:
        flag = torch.tensor([1 if verified.get('bit_identical', True) else 0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if 'bit_identical' in verified:
            verified['bit_identical'] = bool(flag.item())
            verified['ranks_checked'] = world
    events, ops.MSDA_EVENTS = ops.MSDA_EVENTS, None
    if gather_first:
        events, ops.GATHER_EVENTS = ops.GATHER_EVENTS, None
    # Like-for-like companion of the headline: the same step as EAGER launches on ONE stream, one batch at a time -
    # the protocol of rounds 1-3 and of the reference's per-sample benchmark - timed right after the replays (their last use)
    single = None
    if runner.pipe is not None and world == 1:
        wd.stage('single-stream eager companion measurement')
        n_e = max(3, min(a.steps, 6))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_e):
            fdist.pack_detections(*head.get_bboxes_padded(head(inputs, None, metas)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        single = {'value': round(B * n_e / dt, 3), 'unit': 'frames/s', 'ms_per_step': round(dt / n_e * 1e3, 4), 'steps': n_e,
                  'execution': 'eager launches, one stream, one batch in flight'}
    # (graph replay hides the individual launches from the host: then the MSDA events come from the eager pass as well)
    ops.MSDA_EVENTS, ops.DENSE_EVENTS = ([] if (not events and not gather_first) else None), []
    if gather_first and not events:
        ops.GATHER_EVENTS = []
    wd.stage('per-kernel event pass (eager launches)')
    n_pass = max(2, min(a.steps, 4))
    for _ in range(n_pass):
        head.get_bboxes_padded(head(inputs if neck is None else neck(*neck_inputs, metas)[1], None, metas))
    torch.cuda.synchronize()
    if not events:
        events = ops.GATHER_EVENTS if gather_first else ops.MSDA_EVENTS
    ops.GATHER_EVENTS = None
    dense_events, ops.MSDA_EVENTS, ops.DENSE_EVENTS = ops.DENSE_EVENTS, None, None

    # configs[3] (strong scaling: 32 frames sharded over the ranks) measured next to the weak-mode line.  N > 1: in this process,
    # right after the main measurement.  N = 1: the per-GPU share of configs[3] at 8 GPUs (4 frames per step) in a 1-rank RCCL
    # group, run in a child process (its hipGraph replays must not follow this process's synchronisations, runtime.py).
    wd.stage('configs[3] probe', max(a.watchdog, 300.0))
    probe = None
    if a.workload == 'l' and (world > 1 or force_dist) and not strong and not a.no_strong_probe and 32 % world == 0:
        Bs = 32 // world
        sub = [inputs[0][:Bs].contiguous(), [t[:Bs].contiguous() for t in inputs[1]]]
        r2 = Runner(head, sub, metas[:Bs], False, dev)      # (eager: no new capture after this process's replays and synchronisations)
        e2, _, p2, _ = timed(r2, max(a.steps, 20), 3, world, dev)
        probe = {'workload': 'BASELINE.json configs[3]: global batch 32 sharded over the ranks + RCCL all-gather of boxes',
                 'scaling': 'strong', 'frames_per_gpu_per_step': Bs, 'steps': max(a.steps, 20),
                 'value': round(32 * max(a.steps, 20) / e2, 3), 'unit': 'frames/s',
                 'ms_per_step': round(e2 / max(a.steps, 20) * 1e3, 4),
                 'execution': 'eager launches'}
    elif a.workload == 'l' and world == 1 and not strong and not a.no_strong_probe and rank == 0:
        env = dict(os.environ, FF3D_BENCH_FORCE_DIST='1')
        cmd = [sys.executable, os.path.abspath(__file__), '--batch', '4', '--steps', '40', '--warmup', '5', '--channels', str(C),
               '--no-cpu-baseline', '--no-strong-probe', '--graph', 'auto', '--dense', a.dense, '--gemm-dtype', a.gemm_dtype]
        try:
            r_ = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            d_ = json.loads([l for l in r_.stdout.splitlines() if l.startswith('{')][-1])
            probe = {'workload': 'the per-GPU share of BASELINE.json configs[3] (32 frames over 8 GPUs = 4 frames per step and GPU) '
                                 'with the collective active: 1-rank RCCL group on this GPU, the all-gather of the packed detections '
                                 'captured inside every replayed graph (eager mode: on a side stream).  north_star asks >= 6x at 8 '
                                 'GPUs, i.e. this rate >= 0.75 x the 1-GPU rate',
                     'scaling': 'strong (rehearsal on one GPU)', 'frames_per_gpu_per_step': 4, 'steps': d_['steps'],
                     'value': d_['value'], 'unit': 'frames/s per GPU', 'ms_per_step': d_['ms_per_step'],
                     'execution': d_['config']['execution'], 'projected_8gpu_frames_per_s': round(8 * d_['value'], 1)}
        except Exception as e:                                       # the headline line never depends on the probe
            probe = {'error': repr(e)[:300]}

    if rank == 0:
        ms = [ev_[0].elapsed_time(ev_[1]) for ev_ in events]
        avg_ms = sum(ms) / len(ms)
        # gather_first: priced on the UNIQUE bytes of the launch (the un-projected maps + the output): what HBM has to deliver; the
        # C-wide corner rows it REQUESTS (8 x the default mode's) are served by L2 / MALL
        alg_bytes = events[0][3] if gather_first else events[0][2]
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        pm = pmc_entry('pmc_msda', B, C) if (a.workload == 'l' and not gather_first) else None
        out = {
            'metric': METRIC,
            'value': round(total * a.steps / elapsed, 3),
            'unit': 'frames/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(elapsed / a.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': (('fp32-class (wide convs / large GEMMs / attention: fp32 operands as range-normalised 2 x fp16 split pairs = 22-bit '
                       'significand, 3 MFMA passes, f32 accumulate; everything else f32)' if head.dense_mode == 'f16x3' else 'f32')
                      if a.gemm_dtype == 'f32' else 'bf16 decoder GEMMs (bf16 operands and results, f32 accumulate) + fp32-class '
                      'heatmap / pyramid convs, f32 gather accumulation'),
            'data': 'synthetic',
            'verified': verified,
            'config': {'workload': what, 'workload_key': a.workload,
                       'frames_per_gpu_per_step': B, 'global_batch': total, 'channels': C,
                       'parallelism': f'frames sharded dp{world}' + (' + RCCL all-gather of padded detections on a side stream' if (world > 1 or force_dist) else ''),
                       'weights': 'random init of the reference architecture, BN statistics randomised',
                       'dense_layers': {'f16x3': 'wide 3x3 convs (+ large GEMMs) on own split-fp16 MFMA kernels: fp32 operands as '
                                                 '(hi, lo) fp16 pairs, 3 MFMA passes, fp32 accumulate; error vs fp64 = vendor fp32 path',
                                        'vendor': 'MIOpen / hipBLASLt fp32'}[head.dense_mode],
                       'execution': ((f'hipGraph replay, {runner.slots} batches in flight on {runner.slots} streams'
                                      + (', RCCL all-gather captured inside each graph' if collective else ''))
                                     if runner.pipe is not None else 'eager launches') +
                                    ', BEV positional embedding cached per weight load',
                       'batches_in_flight': runner.slots if runner.pipe is not None else 1,
                       'value_mode': a.value_mode,
                       'inputs': (f'{len(pool)} distinct batches resident in HBM, handed to the slots in turn INSIDE the timed region: a producer '
                                  "stream copies batch i into the slot's input buffers in place before its replay (timed)" if pool else
                                  'each slot replays over its own resident batch (static input buffers)'),
                       'single_stream_eager': single,
                       'detections_last_batch': counts, 'ranks': ranks},
            'roofline': {'kernel': (f'msda_fwd_kernel (ff3d_msda_fused_fwd, {a.gemm_dtype} value)' if not gather_first else
                                    'msda_fwd_kernel<SHARED> (ff3d_msda_gather_rows: un-projected C-wide rows per head; achieved = UNIQUE '
                                    f'bytes / time, requested bytes {events[0][2]} per launch = {events[0][2] / (avg_ms * 1e-3) / 1e12:.1f} TB/s from L2 / MALL)'),
                         'bound': 'hbm',
                         'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': pm['traffic_bytes'] if pm else None,
                         # the algorithmic count prices every bilinear corner as an HBM read; corners shared between queries
                         # are served by L2 / MALL, so the counter-based fraction (PMC bytes / this run's launch time) is lower
                         'frac_counter': round(pm['traffic_bytes'] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pm else None,
                         'traffic_measured_at': pm['measured_at'] if pm else None,
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': round(avg_ms, 5),
                         'launches_timed': len(ms),
                         # graph replay hides the individual launches from the host: then the HIP events bracket the launches of
                         # a short eager pass over the same tensors right after the timed region (--graph off: inside it)
                         'timed_in': 'the timed region (eager launches)' if runner.pipe is None else
                                     'an eager pass over the same tensors right after the timed region (the timed steps are graph replays)'},
        }
        if dense_events:
            # the kernel that carries most of the step: split-fp16 dense kernel (3 MFMA passes per fp32 product)
            per = {}
            for s_, e_, tag, fl in dense_events:
                d_ = per.setdefault(tag, [0, 0.0, fl])
                d_[0] += 1
                d_[1] += s_.elapsed_time(e_)
            tag, (n_l, tot, fl) = max(per.items(), key=lambda kv: kv[1][1])
            avg = tot / n_l
            # the launch group that carries most of a step (all launches of one (kernel, shape) tag), for the compact records
            out['top_kernel'] = {'name': tag, 'launches_per_step': round(n_l / n_pass, 2), 'ms_per_step': round(tot / n_pass, 4),
                                 'share_of_step': round(tot / n_pass / (elapsed / a.steps * 1e3), 4)}
            mfma_tf = 3.0 * fl / (avg * 1e-3) / 1e12
            pd = pmc_entry('pmc_dense', B, C) if (' s1 ' in tag and a.workload == 'l') else None
            out['roofline_dense'] = {
                'kernel': f'split-fp16 dense kernel, largest launch: {tag}', 'bound': 'mfma', 'achieved': round(mfma_tf, 1),
                'peak': MFMA_F16_PEAK_TF, 'unit': 'TFLOP/s', 'frac': round(mfma_tf / MFMA_F16_PEAK_TF, 4),
                'traffic': pd['traffic_bytes'] if pd else None, 'mfma_busy_pmc': pd.get('mfma_busy') if pd else None,
                'effective_clock_ghz_pmc': pd.get('effective_clock_ghz') if pd else None,
                'traffic_measured_at': pd['measured_at'] if pd else None,
                'executed_mfma_flops_per_launch': 3.0 * fl, 'algorithmic_fp32_flops_per_launch': fl,
                'fp32_equivalent_tflops': round(fl / (avg * 1e-3) / 1e12, 1), 'fp32_mfma_peak_tflops': 157.3,
                'avg_launch_ms': round(avg, 4), 'launches_timed': n_l,
                'dense_launches_ms': {k: round(v[1] / v[0], 4) for k, v in sorted(per.items())}}
        if probe is not None:
            if 'projected_8gpu_frames_per_s' in probe:
                probe['projected_speedup_8_vs_1'] = round(probe['projected_8gpu_frames_per_s'] / out['value'], 2)
            out['configs3_strong'] = probe
        if (world == 1 and a.workload == 'l' and not strong and not force_dist and not a.no_companions and not a.fresh_inputs
                and runner.pipe is not None):
            wd.stage('companions: fresh inputs, batch-1 latency (children)', 500.0)
            out['config']['fresh_inputs'], out['latency_b1_ms'] = companions(a, out['value'])
            if a.value_mode == 'project_first' and a.gemm_dtype == 'f32':
                out['config']['value_mode_gather_first'] = gather_first_companion(a, out['value'])
        if world == 1 and a.workload == 'l' and not strong and not force_dist and not a.no_other_workloads and a.batch == 32:
            wd.stage('other workloads (children)', 700.0)
            out['other_workloads'] = other_workloads(a)
        if world == 1 and not a.no_cpu_baseline and a.workload == 'l':
            wd.stage('cpu baseline', 3600.0 if a.cpu_full_protocol else max(a.watchdog, 20.0 * a.cpu_budget))
            out['cpu_baseline'] = cpu_baseline(C, a.cpu_budget, a.cpu_full_protocol)
        wd.finish()
        print(json.dumps(out), flush=True)
    wd.finish()
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
