#!/usr/bin/env python
"""Benchmark of the hot path BASELINE.json names: images/sec at 1333x800,
IoU-aware RetinaNet R-50-FPN (fp32), batch 8 per GPU on MI355X.

    python bench.py --gpus N --steps K --warmup W      (N > 1 without a launcher: bench.py
                                                         starts N ranks itself, see launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole inference path over one batch of synthetic
COCO-shaped input already resident in HBM: ResNet-50 + FPN + head convolutions, fp32,
channels-last (3x3 stride-1 convs: HIP Winograd F(4x4,3x3) transforms around hipBLASLt batched
GEMMs; 1x1 convs: hipBLASLt GEMMs with fused BN / residual / ReLU epilogues; the rest: MIOpen)
-> HIP row-max / top-k / gather+decode / batched NMS / final top-100 on the channels-last head
outputs -> (N>1) one RCCL all-gather of the per-image detections.
Weights are random-init (no checkpoints offline), data synthetic; images are
sharded data-parallel (weak scaling: 8 images per GPU per step).

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     -- the row-max kernel over the head logits (k_rowmax_nhwc for this channels-last
                  network, HBM bound).  `achieved` = the bytes THAT kernel moves per launch (class
                  + IoU logits read once, row maxima written: 66 124 800 B per image; the box
                  deltas are not touched by it) / its average launch duration, HIP events on the
                  launch stream inside the timed steps.  Sub-entries:
                    stage  SURVEY 8(d)'s decode-stage unit: 68 544 000 B per image (cls + reg + iou)
                           / the time of row-max + select + gather (events around all launches);
                    wino   the Winograd transforms (largest hand-written time of the step): their
                           algorithmic bytes / their summed durations, events in 2 extra steps
                           after the timed region (43 launches per step);
  cpu_baseline -- the same workload on the host cores (rank 0, N=1 only), one image: PyTorch-CPU
                  convolutions on all cores + the C oracle (oracle/, a port of the reference CPU
                  path, 1 thread); `single_thread` repeats it with one torch thread;
  train        -- BASELINE config 5 on this GPU (rank 0, N=1 only): R-50 training iterations at
                  batch 4 (forward, HIP target assignment + loss kernels, backward, grad clip,
                  SGD), img/s, measured after the inference timing.
  pipeline     -- image -> result, the reference's fps definition (MODEL_ZOO.md:31): uint8 HWC BGR
                  images resident on the device -> ia_image_transform -> network -> detections ->
                  D2H + bbox2result, img/s (rank 0, N=1 only).
  rccl_ranks   -- the size of the process group as the collective library itself counts it
                  (all-reduce of ones); n_gpus comes from the group, not from --gpus.
--config r101-bf16 / x101-64x4d run BASELINE configs 3 / 4 through the same path (sub-records
that belong to config 2 are skipped); --dry-run exercises launcher, process group, timed region
and the result exchange on CPU with the gloo backend and fake detections (no GPU work).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import iouaware  # noqa: E402
from iouaware import ops  # noqa: E402
from iouaware import dist as idist  # noqa: E402
from iouaware.config import ConfigDict  # noqa: E402

IMG_H, IMG_W, PAD_H, PAD_W = 800, 1333, 800, 1344
BATCH = 8
HEAD_BYTES_PER_IMAGE = 68544000          # cls+reg+iou logits, fp32, read once
# what k_rowmax[_nhwc] itself moves: cls (64 512 000) + iou (806 400) read, row maxima (806 400)
# written; the 3 225 600 B of box deltas are read by k_gather for the 4 693 candidates only
ROWMAX_BYTES_PER_IMAGE = 64512000 + 806400 + 806400
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
BF16_PEAK_TFLOPS = 2500.0                # MI355X_MICROARCH.md: dense bf16 MFMA peak (no 2:1 sparsity)
# what "parity" means for the bf16 configuration (there is no 1e-4 contract in bf16: 8 mantissa bits)
BF16_CONTRACT = ('bf16 storage, fp32 accumulation: post-conv path bit-exact against the oracle on the same '
                 'bf16-rounded logits (tests/test_gpu_e2e.py::test_config3_bf16_batch16_post_conv_path); network: '
                 'every stage output (C2..C5, P3..P7, 15 head outputs) no farther from the fp32 evaluation of the '
                 'bf16-rounded weights than 1.5 x torch\'s own bf16 path + 1e-3 and below 0.6 x 2^-8 x sqrt(convs in '
                 'front); detections: at least as many of the fp32 reference detections keep a twin (same class, IoU '
                 '> 0.7) as with torch\'s own bf16 evaluation (86 vs 82 of 100 at 800x1344).  The 1e-4 / bit-exact-'
                 'index contract of the north star is the fp32 path\'s')

MODEL = dict(
    type='RetinaNet', pretrained=None,
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                  frozen_stages=1, style='pytorch'),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
              add_extra_convs=True, num_outs=5),
    bbox_head=dict(type='IoUawareRetinaHead', num_classes=81, in_channels=256, stacked_convs=4,
                   feat_channels=256, octave_base_scale=4, scales_per_octave=3,
                   anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[8, 16, 32, 64, 128],
                   target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                   loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25,
                                 loss_weight=1.0),
                   loss_bbox=dict(type='SmoothL1Loss', beta=0.11, loss_weight=1.0)))
TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_thr=0.5),
                max_per_img=100)


def build_model(device, fuse=True, channels_last=False, winograd=True, backbone=None):
    torch.manual_seed(0)
    cfg = ConfigDict(MODEL)
    if backbone:
        cfg.backbone.update(backbone)
    model = iouaware.build_detector(cfg, train_cfg=None, test_cfg=ConfigDict(TEST_CFG))
    model = model.to(device).eval()
    if fuse:
        from iouaware.fuse import fuse_inference
        # conv epilogues (BN / bias / add / ReLU) -> one HIP pass each; the head's 3x3 convs ->
        # Winograd F(4x4,3x3) transforms (HIP) around one batched fp32 GEMM per layer
        fuse_inference(model, winograd=winograd and channels_last)
    if channels_last:
        # MIOpen's fp32 NHWC implicit-GEMM kernels beat the NCHW Winograd path on this net
        # (tools/try_layouts.py); the three head outputs are brought back to NCHW by
        # ops.level_ptrs (.contiguous()) for the HIP head kernels.
        model = model.to(memory_format=torch.channels_last)
    return model


def metas(batch):
    return [dict(ori_shape=(IMG_H, IMG_W, 3), img_shape=(IMG_H, IMG_W, 3),
                 pad_shape=(PAD_H, PAD_W, 3), scale_factor=1.0, flip=False) for _ in range(batch)]


class Stepper(object):
    """one benchmark step = the product path: network forward + ops.get_bboxes (one fused C-ABI
    call for the whole post-conv path) (+ the all-gather when N > 1).  No stage-wise launches
    inside the timed region.  With `stage_events(n)` armed, each of the next n steps hands the
    library a fresh pair of HIP events that ops.get_bboxes records on its stream in front of the
    decode stage's first launch and behind its last one (ia_profile_stage_events): the stage timed
    INSIDE the steps of the timed region (two event records per step, no synchronisation); the
    back-to-back figures of decode_stage_roofline() stay beside it."""
    events, ev_next = (), 0                  # (class defaults: subclasses with their own __init__)
    pending = results = None
    host_exchange, local_last = False, None

    def __init__(self, model, imgs, world):
        self.model, self.imgs, self.world = model, imgs, world
        self.metas = metas(imgs.shape[0])
        self.cfg = model.test_cfg
        self.last = None
        self.events, self.ev_next = [], 0
        # the step ends where the reference's simple_test ends (single_stage.py:90-96): the rank's
        # detections on the host as per-class arrays (bbox2result).  Batch i's device-to-host copy
        # (one pinned record, ~19 KB) is enqueued behind its detections and COLLECTED while batch
        # i + 1 runs on the device; drain() collects the last one inside the timed region.
        self.pending, self.results, self.host_buf, self.flip = None, None, [None, None], 0
        self.exchange_ms, self.exchange_ev = [], []

    def exchange_record(self, steps):
        """the all-gather of the last `steps` steps: mean / max duration (call after a device synchronisation)"""
        ms = self.exchange_ms if self.host_exchange else [a.elapsed_time(b) for a, b in self.exchange_ev]
        ms = ms[-steps:]
        if not ms:
            return None
        M = int(self.last[0].shape[1])
        return {'collective': 'gloo all_gather_into_tensor of host records (rehearsal)' if self.host_exchange else
                              'RCCL all_gather_into_tensor on the step\'s stream (HIP events around it)',
                'ms_mean': round(sum(ms) / len(ms), 4), 'ms_max': round(max(ms), 4), 'steps': len(ms),
                'bytes_per_rank': int(self.imgs.shape[0]) * (M * 6 + 1) * 4}

    def interleave_matches_part_list(self):
        """the gathered records of the last step against the reference's collect_results (tools/test.py:95-99):
        part_list = every rank's results in rank order; ordered = [res for tup in zip(*part_list) for res in tup]"""
        mine = [t.cpu().numpy() for t in self.local_last]
        part_list = [None] * self.world
        dist.all_gather_object(part_list, [tuple(a[i] for a in mine) for i in range(mine[0].shape[0])])
        ordered = [res for tup in zip(*part_list) for res in tup]
        got = [t.cpu().numpy() for t in self.last[:3]]
        return len(ordered) == got[0].shape[0] and all(
            np.array_equal(got[0][i], d) and np.array_equal(got[1][i], l) and int(got[2][i]) == int(n)
            for i, (d, l, n) in enumerate(ordered))

    def submit_results(self, dets, labels, num):
        from iouaware.detectors import PendingResults
        rec = idist.pack_detections(dets, labels, num)
        buf = self.host_buf[self.flip]
        if buf is None or buf.shape != rec.shape:
            buf = self.host_buf[self.flip] = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
        self.flip ^= 1
        buf.copy_(rec, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return PendingResults(buf, done, dets.shape[1], self.model.bbox_head.num_classes)

    def drain(self):
        """collect what is still on its way (end of the warm-up and of the timed region)"""
        if self.pending is not None:
            self.results, self.pending = self.pending.collect(), None

    def stage_events(self, n, skip=0):
        """arm n event pairs for the steps after the next `skip` ones (created by one record each:
        torch makes the HIP event lazily)"""
        self.events, self.ev_next = [], -skip
        for _ in range(n):
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
            pair[1].record()
            self.events.append(pair)
        torch.cuda.synchronize()

    def stage_times_ms(self):
        """-> the decode stage's duration in each step that used an event pair (call after a
        device synchronisation); switches the hook off"""
        ops.stage_events(None, None)
        return [a.elapsed_time(b) for a, b in self.events[:max(0, min(self.ev_next, len(self.events)))]]

    @torch.no_grad()
    def step(self, timed=False):
        """one benchmark step = this rank's detections (+ the all-gather when N > 1)"""
        if self.events and self.ev_next <= len(self.events):
            if 0 <= self.ev_next < len(self.events):
                ops.stage_events(*self.events[self.ev_next])
            elif self.ev_next == len(self.events):
                ops.stage_events(None, None)
            self.ev_next += 1
        dets, labels, num, cls, reg, iou = self.local_detections(timed)
        nxt = self.submit_results(dets, labels, num) if dets.is_cuda else None
        if self.world > 1:
            self.local_last = (dets, labels, num)
            if self.host_exchange or not dets.is_cuda:
                # --rehearsal (gloo): the records travel as host tensors; the copy waits for this rank's
                # detections, the exchange proper is timed behind it
                local = [t.cpu() for t in (dets, labels, num)]
                t0 = time.perf_counter()
                dets, labels, num = idist.all_gather_detections(*local)
                self.__dict__.setdefault('exchange_ms', []).append((time.perf_counter() - t0) * 1e3)
            else:
                # RCCL all-gather on the step's stream, between two events (read after the timed region)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                dets, labels, num = idist.all_gather_detections(dets, labels, num)
                ev[1].record()
                self.__dict__.setdefault('exchange_ev', []).append(ev)
        if self.pending is not None:
            self.results = self.pending.collect()          # the previous batch: host work under this batch's device work
        self.pending = nxt
        self.last = (dets, labels, num, cls, reg, iou)
        return dets

    def local_detections(self, timed=False):
        m = self.model
        cls, reg, iou = m.forward_head(self.imgs)
        geom = m.bbox_head.geometry([tuple(c.shape[-2:]) for c in cls], self.cfg.get('nms_pre', -1))
        shapes = [x['img_shape'] for x in self.metas]
        factors = [x['scale_factor'] for x in self.metas]
        dets, labels, rows, num = ops.get_bboxes(geom, cls, reg, iou, shapes, factors, True,
                                                 self.cfg.score_thr, self.cfg.nms.iou_thr,
                                                 self.cfg.max_per_img)
        return dets, labels, num, cls, reg, iou


@torch.no_grad()
def decode_stage_roofline(stepper, reps=10, rounds=5):
    """SURVEY 8(d)'s decode stage on the head outputs of the last step, after the timed region:
    the same kernels ops.get_bboxes launches (row-max with group maxima -> top-k filter -> top-k
    final -> gather / decode), `reps` passes back to back between ONE pair of HIP events on the
    launch stream (an event pair costs 3-7 us, and an event between two kernels delays the
    dependent launch), best of `rounds`; then the row-max kernel alone the same way.
    -> (ms per stage pass, ms per row-max launch, channels-last?)"""
    cls, reg, iou = stepper.last[3:6]
    m = stepper.model
    geom = m.bbox_head.geometry([tuple(c.shape[-2:]) for c in cls], stepper.cfg.get('nms_pre', -1))
    geom = ops.geometry_for(geom, cls, reg, iou)
    shapes = [x['img_shape'] for x in stepper.metas]
    factors = [x['scale_factor'] for x in stepper.metas]
    ws = ops.select_workspace(geom, cls[0].shape[0], cls[0].device)
    # one C-ABI call per pass (ia_decode_stage: the four launches ia_get_bboxes starts with, in
    # its workspace) -- the host stays far ahead of the device
    stage = ops.DecodeStage(geom, cls, reg, iou, shapes, factors, True).run

    def rowmax():
        return ops.decode_fuse_rowmax(geom, cls, reg, iou, ws)

    def best_of(fn):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps
            best = t if best is None or t < best else best
        return best
    return best_of(stage), best_of(rowmax), bool(geom.layout)


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _numa_cores():
    """physical cores (one hardware thread each) of NUMA node 0 and of the whole host, restricted
    to this process's affinity mask: [[cpu ids of node 0], [cpu ids of all nodes]]"""
    allowed = os.sched_getaffinity(0)

    def parse(txt):
        out = []
        for part in txt.strip().split(','):
            if '-' in part:
                lo, hi = part.split('-')
                out += list(range(int(lo), int(hi) + 1))
            elif part:
                out.append(int(part))
        return out

    def first_threads(cpus):
        seen, out = set(), []
        for c in sorted(cpus):
            try:
                with open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c) as f:
                    sib = tuple(sorted(parse(f.read())))
            except OSError:
                sib = (c,)
            if sib not in seen:
                seen.add(sib)
                out.append(c)
        return out
    try:
        with open('/sys/devices/system/node/node0/cpulist') as f:
            node0 = [c for c in parse(f.read()) if c in allowed]
    except OSError:
        node0 = sorted(allowed)
    return first_threads(node0 or sorted(allowed)), first_threads(sorted(allowed))


def cpu_worker(spec):
    """--cpu-worker: one point of the CPU sweep in its own process, so that the OpenMP runtime
    starts with the binding of THIS point (OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES are read
    once, at its initialisation).  Convolutions: PyTorch-CPU (oneDNN) on the plain modules, random
    init seed 0 like the GPU model; post-conv path: the C oracle, one image per thread."""
    spec = json.loads(spec)
    threads, batch = int(spec['threads']), int(spec['batch'])
    if spec.get('cpus'):
        os.sched_setaffinity(0, set(spec['cpus']))
    torch.set_num_threads(threads)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle
    oracle.build()
    model = build_model(torch.device('cpu'), fuse=False)
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(batch, 3, PAD_H, PAD_W, generator=g)
    if spec.get('channels_last'):
        model = model.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        # one untimed pass: oneDNN creates its primitives / reorders the weights at the first call
        # of every shape, which is not steady-state throughput
        model.forward_head(img)
        t0 = time.time()
        cls, reg, iou = model.forward_head(img)
        t_conv = time.time() - t0
    head = model.bbox_head
    base = np.stack([a.base_anchors.numpy() for a in head.anchor_generators])
    cls, reg, iou = ([t.contiguous().numpy() for t in x] for x in (cls, reg, iou))

    def post(b):
        return oracle.get_bboxes_single([c[b] for c in cls], [r[b] for r in reg], [i[b] for i in iou],
                                        head.anchor_strides, base, (IMG_H, IMG_W), 1.0, True,
                                        TEST_CFG['nms_pre'], TEST_CFG['score_thr'],
                                        TEST_CFG['nms']['iou_thr'], TEST_CFG['max_per_img'])
    t0 = time.time()
    if batch == 1:
        res = [post(0)]
    else:
        # the reference's NMS is serial per image; images are independent -> one per thread
        # (ctypes drops the GIL inside the C call)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(batch, threads)) as ex:
            res = list(ex.map(post, range(batch)))
    t_post = time.time() - t0
    print(json.dumps(dict(threads=threads, batch=batch, conv_s=round(t_conv, 3), post_s=round(t_post, 3),
                          img_per_s=round(batch / (t_conv + t_post), 4),
                          into_nms=int((res[0]['mlvl_scores'] > TEST_CFG['score_thr']).sum()),
                          dets=int(res[0]['num_det']))))


def cpu_baseline(model=None, stepper=None, budget_s=150.0):
    """The same workload on the host cores: a thread SWEEP, each point in its own
    process bound to the first n physical cores of NUMA node 0 (OMP_PROC_BIND=close,
    OMP_PLACES=cores, affinity mask), one warm-up pass, one timed pass:
      latency-configured    ONE image at a time (the reference asserts batch 1 at test time,
                            base.py:96-98) on 1 / 8 / 16 / 32 / 64 / all cores of the socket;
                            `value` = the best point, the table is reported;
      throughput-configured batch 8 in one forward on the socket's cores (and on all sockets), the
                            eight post-conv problems on eight threads.
    Convolutions: PyTorch-CPU; post-conv path: the C oracle (oracle/, a port of the reference CPU
    path pinned against it by tests/golden; serial per image by construction)."""
    node0, allc = _numa_cores()
    t_start = time.time()

    def run(threads, batch, cpus, timeout):
        env = dict(os.environ)
        env.update(OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND='close',
                   OMP_PLACES='cores', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        spec = json.dumps(dict(threads=threads, batch=batch, cpus=cpus))
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', spec], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            return json.loads(out.stdout.decode().strip().splitlines()[-1])
        except Exception as exc:
            return dict(threads=threads, batch=batch, error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
    points = sorted(set(n for n in (1, 8, 16, 32, 64, len(node0)) if n <= len(node0)))
    table = []
    for n in points:
        if time.time() - t_start > budget_s * 0.6:
            table.append(dict(threads=n, batch=1, skipped='time budget'))
            continue
        table.append(run(n, 1, node0[:n], 120))
    ok = [r for r in table if 'img_per_s' in r]
    best = max(ok, key=lambda r: r['img_per_s']) if ok else None
    one = next((r for r in ok if r['threads'] == 1), None)
    thr = []
    for n, cpus, what in ((len(node0), node0, 'one socket (NUMA node 0)'), (len(allc), allc, 'all sockets')):
        if (what == 'all sockets' and len(allc) == len(node0)) or time.time() - t_start > budget_s:
            continue
        r = run(n, BATCH, cpus, 240)
        r['cores_of'] = what
        thr.append(r)
    thr_ok = [r for r in thr if 'img_per_s' in r]
    tbest = max(thr_ok, key=lambda r: r['img_per_s']) if thr_ok else None
    name = cpu_model_name()
    return dict(value=best['img_per_s'] if best else None, unit='img/s',
                cores=best['threads'] if best else None, kind='port', cpu=name, host_cores=os.cpu_count(),
                physical_cores_node0=len(node0), physical_cores_host=len(allc),
                configuration='latency-configured: ONE image at a time (the reference asserts batch 1 at test '
                              'time, base.py:96-98); best point of a thread sweep, every point in its own process '
                              'bound to the first n physical cores of NUMA node 0 (OMP_PROC_BIND=close, '
                              'OMP_PLACES=cores), one warm-up pass then one timed pass',
                sample='1 image (3x800x1344, random-init R-50): PyTorch-CPU convs %.2f s on %d threads + C oracle '
                       'get_bboxes %.2f s on 1 thread (%d boxes into NMS); host: %s, %d hardware threads'
                       % (best['conv_s'], best['threads'], best['post_s'], best['into_nms'], name,
                          os.cpu_count()) if best else 'no point of the sweep finished',
                sweep=table,
                single_thread=dict(value=one['img_per_s'], unit='img/s', cores=1,
                                   sample='convs %.2f s + C oracle %.2f s' % (one['conv_s'], one['post_s']))
                if one else None,
                throughput=dict(value=tbest['img_per_s'], unit='img/s', cores=tbest['threads'], batch=BATCH,
                                configuration='throughput-configured: batch 8 in one forward, the 8 post-conv '
                                              'problems on 8 threads (serial per image by construction)',
                                runs=thr) if tbest else dict(value=None, runs=thr),
                seconds=round(time.time() - t_start, 1))


TRAIN_CFG = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4,
                               min_pos_iou=0, ignore_iof_thr=-1),
                 allowed_border=-1, pos_weight=-1, debug=False)
TRAIN_BATCH = 4                           # imgs_per_gpu of the reference's 4-GPU config (:78)


def train_state(device, find=False, channels_last=False, fuse=True, seed=0):
    """model, optimizer and one synthetic batch of a BASELINE config 5 training iteration"""
    from iouaware.train import build_optimizer
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import synth
    # MIOpen immediate mode: find mode times every candidate of every forward / backward-data /
    # backward-weight convolution in a fresh process (~8 minutes here) for ~7 % more img/s
    torch.backends.cudnn.benchmark = bool(find)
    torch.manual_seed(0)                      # the same initial weights on every rank
    model = iouaware.build_detector(ConfigDict(MODEL), train_cfg=ConfigDict(TRAIN_CFG),
                                    test_cfg=ConfigDict(TEST_CFG)).to(device).train()
    opt = build_optimizer(model, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001))
    g = torch.Generator(device=device).manual_seed(7 + seed)
    img = torch.randn(TRAIN_BATCH, 3, PAD_H, PAD_W, device=device, generator=g)
    if fuse:
        # bottlenecks / FPN: one autograd node per convolution on the GEMM / Winograd kernels,
        # eval-mode BatchNorm folded differentiably; frozen stem + stage 1 on the inference kernels
        from iouaware.fuse import fuse_inference
        fuse_inference(model, winograd=True, train=True)
        channels_last = True
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)
    gts, gls = synth.train_targets(5 + seed, TRAIN_BATCH, IMG_H, IMG_W, max_gt=20)
    gtb = [torch.from_numpy(x).to(device) for x in gts]
    gtl = [torch.from_numpy(x).to(device) for x in gls]
    return model, opt, img, metas(TRAIN_BATCH), gtb, gtl


def train_record(device, iters=5, warmup=3, loss_part=True, find=False, channels_last=False,
                 fuse=True):
    """BASELINE config 5, one GPU: whole training iterations of R-50 IoU-aware RetinaNet at
    800x1344, batch 4, fp32 -- forward, device target assignment + all-levels loss kernels,
    backward, gradient clipping, SGD (reference mmdet/apis/train.py:38-45, optimizer_config of
    the configs).  Gradient all-reduce is not part of a single-GPU measurement."""
    from iouaware.train import train_step
    model, opt, img, ms, gtb, gtl = train_state(device, find=find, channels_last=channels_last, fuse=fuse)
    clip = dict(max_norm=35, norm_type=2)
    for _ in range(warmup):
        lv = train_step(model, opt, img, ms, gtb, gtl, grad_clip=clip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        lv = train_step(model, opt, img, ms, gtb, gtl, grad_clip=clip)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # loss part alone on the last head outputs: HIP events around targets + losses fwd + bwd
    with torch.no_grad():
        outs = model.bbox_head(model.extract_feat(img))
    outs = [[t.detach().requires_grad_(True) for t in o] for o in outs]
    from iouaware.train import parse_losses
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run_loss_part():
        for o in outs:
            for t in o:
                t.grad = None
        loss, _ = parse_losses(model.bbox_head.loss(*outs, gtb, gtl, ms, model.train_cfg))
        loss.backward()
    for _ in range(3 if loss_part else 0):
        run_loss_part()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10 if loss_part else 0):
        run_loss_part()
    e1.record()
    torch.cuda.synchronize()
    return dict(value=round(TRAIN_BATCH / dt, 2), unit='img/s', ms_per_iter=round(dt * 1e3, 2),
                batch=TRAIN_BATCH, dtype='fp32', iters=iters, warmup=warmup,
                loss_part_ms=round(e0.elapsed_time(e1) / 10, 3) if loss_part else None,
                miopen_find_mode=bool(find),
                loss=round(float(lv['loss']), 4),
                workload='R-50 IoU-aware RetinaNet training iteration, 3x800x1344, batch 4 on one '
                         'GPU: fwd + HIP targets / losses + bwd + clip + SGD (random init, synthetic)')


PMC_PROFILE = 'r04_head_pmc.json'
WINO_PMC_PROFILE = 'r03_wino_pmc.json'


def wino_roofline(stepper, steps=2):
    """events around every Winograd transform launch of `steps` extra (untimed) steps"""
    from iouaware import winograd
    winograd.TIMING = []
    try:
        for _ in range(steps):
            # rank 0 alone runs these extra passes: no collective in here (stepper.step would
            # all-gather and hang the other ranks)
            with torch.no_grad():
                stepper.local_detections()
        torch.cuda.synchronize()
        rec = [(k, e0.elapsed_time(e1), b) for k, e0, e1, b in winograd.TIMING]
    finally:
        winograd.TIMING = None
    if not rec:
        return None
    ms = sum(r[1] for r in rec)
    nbytes = sum(r[2] for r in rec)
    achieved = nbytes / (ms * 1e-3) / 1e9
    per = {}
    for kind in ('in', 'out'):
        sel = [r for r in rec if r[0] == kind]
        if sel:
            per[kind] = dict(launches_per_step=len(sel) // steps,
                             ms_per_step=round(sum(r[1] for r in sel) / steps, 3),
                             achieved=round(sum(r[2] for r in sel) / (sum(r[1] for r in sel) * 1e-3)
                                            / 1e9, 1))
    traffic = None
    try:                         # measured HBM bytes of the head-layer launches (tools/collect_wino_pmc.sh)
        with open(os.path.join(ROOT, 'profiles', WINO_PMC_PROFILE)) as f:
            prof = json.load(f)
        traffic = dict(launch='head layer, batch 8, both towers (11440 tiles x 512 channels)',
                       algorithmic_bytes=int(prof['algorithmic_bytes_per_launch']),
                       k_wino_in=int(prof['kernels']['ia::k_wino_in']['traffic_bytes_per_launch']),
                       k_wino_out=int(prof['kernels']['ia::k_wino_out']['traffic_bytes_per_launch']),
                       source='profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate '
                              'passes, 2 x FETCH_SIZE)' % WINO_PMC_PROFILE)
    except Exception:
        pass
    return dict(bound='hbm', kernel='k_wino_in + k_wino_out', achieved=round(achieved, 1),
                peak=HBM_PEAK_GBS, unit='GB/s', frac=round(achieved / HBM_PEAK_GBS, 4),
                launches_per_step=len(rec) // steps, ms_per_step=round(ms / steps, 3),
                bytes_per_step=nbytes // steps, by_kernel=per, traffic=traffic,
                note='algorithmic bytes: input transform 16 + 36, output transform 36 + 16 fp32 '
                     'values per tile and channel; HIP events in %d steps after the timed region'
                     % steps)


def timed_region(step, steps, warmup, world, sync, barrier, device, drain=None, info=None):
    """the driver's timing contract: W untimed warm-up steps, then EXACTLY K steps between
    barrier + device synchronisation on both sides; the MAX over ranks of the elapsed time.
    drain: finishes whatever the last step left in flight on the host side (the last batch's
    result collection) -- called before the clock starts and before it stops, so the K steps'
    results are complete inside the timed region."""
    for _ in range(warmup):
        step()
    if drain is not None:
        drain()
    if world > 1:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if drain is not None:
        drain()
    if info is not None:
        sync()
        info['local_busy_s'] = time.perf_counter() - t0    # this rank's steps done (before it waits for the others)
    if world > 1:
        barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if info is not None:
        info['local_elapsed_s'] = elapsed                  # this rank's own clock (the line reports every rank's)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rowmax_traffic(kernel):
    """HBM bytes per k_rowmax launch from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 correction 2 x FETCH_SIZE; tools/collect_pmc.sh).  It is a
    batch-8 launch like the benchmark's; None when the profile is missing."""
    path = os.path.join(ROOT, 'profiles', PMC_PROFILE)
    try:
        with open(path) as f:
            prof = json.load(f)
        if prof.get('batch') != BATCH:
            return None
        return int(prof['kernels'][kernel]['traffic_bytes_per_launch'])
    except Exception:
        return None


def live_stage_traffic(timeout_s=150):
    """HBM bytes of the decode stage's kernels MEASURED IN THIS RUN (the traffic
    figure used to come from a committed profile only): two child processes under rocprofv3 --
    `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes, `--kernel-trace` only, as
    MI355X_MICROARCH.md prescribes -- run tools/time_head.py (the same batch-8 launches of the head
    path on random-init-like logits) after the timed region; traffic = 2 * FETCH_SIZE * 1024 +
    WRITE_SIZE * 1024 (gfx950: FETCH_SIZE counts half of a wide coalesced stream).  Bounded: a
    child that exceeds `timeout_s` is killed with its process group and the committed profile is
    reported instead.  Returns {'stage': bytes, 'rowmax': bytes, 'kernels': {...}} or None."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    script = os.path.join(ROOT, 'tools', 'time_head.py')
    if exe is None or not os.path.exists(script):
        return None
    per = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        tmp = tempfile.mkdtemp(prefix='ia_pmc_', dir='/tmp')
        env = dict(os.environ, TMPDIR='/tmp')
        proc = None
        try:
            proc = subprocess.Popen([exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', tmp,
                                     '--', sys.executable, script, str(BATCH), 'D', '3'], cwd='/tmp', env=env,
                                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            proc.wait(timeout=timeout_s)
            files = glob.glob(os.path.join(tmp, '*', '*counter_collection.csv'))
            if proc.returncode != 0 or not files:
                return None
            agg = {}
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if 'ia::' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
                        agg.setdefault(k, []).append(float(r['Counter_Value']))
            for k, v in agg.items():
                per.setdefault(k, {})[counter] = sum(v) / len(v)
        except Exception:
            if proc is not None and proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)      # the exact group started above
                except OSError:
                    pass
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    byts = {k: int(2 * d.get('FETCH_SIZE', 0.0) * 1024 + d.get('WRITE_SIZE', 0.0) * 1024) for k, d in per.items()}
    stage = [k for k in byts if k.startswith(('ia::k_rowmax_filter', 'ia::k_sel_final', 'ia::k_gather_nhwc'))]
    rowmax = [k for k in byts if k.startswith('ia::k_rowmax_nhwc')]
    if len(stage) < 3:
        return None
    return dict(stage=sum(byts[k] for k in stage), rowmax=byts[rowmax[0]] if rowmax else None,
                kernels={k: byts[k] for k in stage + rowmax})


CONFIGS = {
    # name -> (BASELINE config, backbone overrides, images per GPU per step, torch dtype name)
    'r50': ('config 2: IoU-aware RetinaNet R-50-FPN fp32, batch 8', {}, 8, 'float32'),
    'r101-bf16': ('config 3: IoU-aware RetinaNet R-101-FPN bf16, batch 16 per GPU',
                  dict(depth=101), 16, 'bfloat16'),
    'x101-64x4d': ('config 4: IoU-aware RetinaNet X-101-64x4d-FPN fp32, batch 8, grouped 3x3 '
                   'convolutions on the MFMA kernel (csrc/gconv.hip)',
                   dict(type='ResNeXt', depth=101, groups=64, base_width=4), 8, 'float32'),
    'r50-train': ('config 5: IoU-aware RetinaNet R-50-FPN training step (HIP target / loss kernels), '
                  'batch 4 per GPU, data-parallel', {}, 4, 'float32'),
}


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(n, argv, script=None):
    """Start n ranks of this script, one per GPU, the way the reference's launcher does
    (tools/dist_test.sh:7-10: `python -m torch.distributed.launch --nproc_per_node=$GPUS ...`,
    mmdet/apis/env.py:26-31): `torch.distributed.run` sets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* and the ranks rendezvous on 127.0.0.1.  Returns the launcher's exit code; rank 0's
    JSON line goes to this process's stdout."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC for RCCL (host driver)
    env.setdefault('OMP_NUM_THREADS', '8')
    env['IA_BENCH_LAUNCHED_BY'] = 'bench.py --gpus %d (self-launch)' % n
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           script or os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class DryRunStepper(Stepper):
    """--dry-run: per-rank fake detections (deterministic in the dataset index) instead of the
    network -- launcher, process group, timed region, the all-gather and its rank interleave on
    CPU / gloo, which is everything of the N > 1 path that is not GPU work."""
    B, M = 3, 100

    def __init__(self, rank, world):
        self.rank, self.world, self.last, self.calls = rank, world, None, 0

    @staticmethod
    def fake(i, M=100):
        rs = np.random.RandomState(1000 + i)
        k = int(rs.randint(0, M + 1))
        dets = np.zeros((M, 5), np.float32)
        dets[:k] = rs.uniform(0, 1000, (k, 5)).astype(np.float32)
        labels = np.full(M, -1, np.int32)
        labels[:k] = rs.randint(0, 80, k)
        return dets, labels, k

    def local_detections(self, timed=False):
        self.calls += 1
        d, l, n = zip(*[self.fake(j * self.world + self.rank, self.M) for j in range(self.B)])
        return (torch.from_numpy(np.stack(d)), torch.from_numpy(np.stack(l)),
                torch.tensor(n, dtype=torch.int32), None, None, None)

    def check(self):
        D, L, N = self.last[:3]
        ok = D.shape[0] == self.world * self.B
        for i in range(self.world * self.B):
            ed, el, ek = self.fake(i, self.M)
            ok = ok and bool(np.array_equal(D[i].numpy(), ed) and np.array_equal(L[i].numpy(), el)
                             and int(N[i]) == ek)
        return bool(ok)


def group_ranks(device):
    """the size of the process group as the collective library counts it: an all-reduce of ones
    (RCCL on GPUs), plus the distinct devices behind the ranks"""
    one = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(one)
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, (os.environ.get('LOCAL_RANK', '0'), str(device)))
    return int(one.item()), sorted(set(ids))


def pipeline_record(model, device, batch, steps=5, warmup=2):
    """image -> result, the reference's fps definition (MODEL_ZOO.md:31 "overall time including
    data loading, network forwarding and post processing" minus the disk): uint8 HWC BGR images
    resident on the device -> ia_image_transform (resize to (1333, 800) keep-ratio, normalise,
    pad to /32: mmdet/datasets/transforms.py:31-50) -> network -> detections -> one D2H copy ->
    bbox2result (core/bbox/transforms.py:148-166)."""
    from iouaware.preprocess import ImageTransform
    g = torch.Generator(device=device).manual_seed(99)
    # 480 x 800 sources -> scale 1.66625 -> img_shape (800, 1333), pad_shape (800, 1344)
    raws = [torch.randint(0, 256, (480, 800, 3), dtype=torch.uint8, device=device, generator=g)
            for _ in range(batch)]
    tf = ImageTransform(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True,
                        size_divisor=32)
    dtype = next(model.parameters()).dtype

    @torch.no_grad()
    def one():
        img, ms = tf.batch(raws, (1333, 800), keep_ratio=True, channels_last=True)
        if dtype != torch.float32:
            img = img.to(dtype)
        return model.simple_test_batch(img, ms, rescale=True), ms
    for _ in range(warmup):
        res, ms = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res, ms = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps

    # the same work as a serving loop: batch i+1 is submitted before batch i is collected, so
    # the host's bbox2result overlaps the device (SingleStageDetector.simple_test_batch_submit)
    @torch.no_grad()
    def submit():
        img, ms = tf.batch(raws, (1333, 800), keep_ratio=True, channels_last=True)
        if dtype != torch.float32:
            img = img.to(dtype)
        return model.simple_test_batch_submit(img, ms, rescale=True)
    pend = submit()
    for _ in range(warmup):
        nxt = submit(); res2 = pend.collect(); pend = nxt
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nxt = submit(); res2 = pend.collect(); pend = nxt
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / steps
    pend.collect()
    same = all(a.shape == b.shape and (a.size == 0 or float(np.abs(a - b).max()) < 1e-4)
               for ra, rb in zip(res, res2) for a, b in zip(ra, rb))
    return dict(value=round(batch / dt, 2), unit='img/s', ms_per_step=round(dt * 1e3, 3),
                overlapped=dict(value=round(batch / dt2, 2), unit='img/s',
                                ms_per_step=round(dt2 * 1e3, 3), matches_synchronous_results=bool(same),
                                note='batch i+1 submitted before batch i is collected: the '
                                     'host post-processing overlaps the device'),
                batch=batch, steps=steps, warmup=warmup,
                img_shape=list(ms[0]['img_shape']), pad_shape=list(ms[0]['pad_shape']),
                dets_image0=int(sum(r.shape[0] for r in res[0])),
                workload='uint8 480x800x3 BGR images on the device -> ia_image_transform '
                         '(keep-ratio resize to 800x1333, normalise, pad to 800x1344) -> network -> '
                         'get_bboxes -> one D2H -> bbox2result (80 per-class arrays per image)')


STAGE_PREFIXES = ('ia::k_rowmax', 'ia::k_sel', 'ia::k_gather', 'ia::k_dec')


@torch.no_grad()
def conv3x3_bf16_rate(device, batch=16, reps=20):
    """k_conv3x3_bf16 on the head-tower shape of config 3 (256 -> 256, 100 x 168, batch 16, bias +
    ReLU): TFLOP/s from HIP events around `reps` back-to-back launches in this run."""
    H, W, Cn = 100, 168, 256
    x = torch.randn(batch, Cn, H, W, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cn, Cn, 3, 3, device=device) * 0.03).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cn, device=device)
    wp = ops.conv3x3_bf16_pack(w)
    for _ in range(3):
        ops.conv3x3_bf16(x, wp, b, Cn, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.conv3x3_bf16(x, wp, b, Cn, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * batch * H * W * Cn * Cn * 9 / (ms * 1e-3) / 1e12
    return {'kernel': 'k_conv3x3_bf16', 'shape': '256 -> 256, 100 x 168, batch %d, bias + ReLU' % batch,
            'avg_launch_ms': round(ms, 4), 'achieved': round(tf, 1), 'peak': BF16_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(tf / BF16_PEAK_TFLOPS, 4),
            'timing': 'HIP events around %d back-to-back launches in this run' % reps}


OTHER_CONFIGS = (
    # (record name, BASELINE config it stands for, backbone overrides, batch, dtype, steps, warmup)
    ('config1_r50_fp32_batch1', 'config 1 on the GPU: IoU-aware RetinaNet R-50-FPN fp32, ONE 1333x800 image per step '
     '(the reference CPU run of config 1 is cpu_baseline)', {}, 1, 'float32', 10, 3),
    ('config3_r101_bf16_batch16', CONFIGS['r101-bf16'][0], dict(depth=101), 16, 'bfloat16', 5, 2),
    ('config4_x101_64x4d_fp32_batch8', CONFIGS['x101-64x4d'][0],
     dict(type='ResNeXt', depth=101, groups=64, base_width=4), 8, 'float32', 4, 2),
)


def other_configs_record(device, budget_s=60.0):
    """BASELINE configs 1, 3, 4 as bounded sub-records of the default run: the
    same whole inference path (network -> ops.get_bboxes) on one GPU, a few steps each between a
    device synchronisation on both sides; config 3 also carries the bf16 3x3 kernel's rate."""
    out, t_start = {}, time.time()
    for name, what, backbone, batch, dtype_name, steps, warmup in OTHER_CONFIGS:
        if time.time() - t_start > budget_s:
            out[name] = {'skipped': 'time budget of %.0f s used up' % budget_s}
            continue
        try:
            dtype = getattr(torch, dtype_name)
            model = build_model(device, fuse=True, channels_last=True, winograd=True, backbone=backbone)
            g = torch.Generator(device=device).manual_seed(4321)
            imgs = torch.randn(batch, 3, PAD_H, PAD_W, device=device, generator=g)
            if dtype != torch.float32:
                model, imgs = model.to(dtype), imgs.to(dtype)
            imgs = imgs.contiguous(memory_format=torch.channels_last)
            st = Stepper(model, imgs, 1)
            for _ in range(warmup):
                st.step()
            st.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step()
            st.drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rec = {'workload': what + '; 3x800x1344, random-init weights, whole inference path incl. NMS',
                   'batch': batch, 'dtype': 'fp32' if dtype == torch.float32 else 'bf16',
                   'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 3),
                   'value': round(batch * steps / dt, 2), 'unit': 'img/s',
                   'dets_per_image': int(st.last[2].float().mean().item())}
            if dtype == torch.bfloat16:
                rec['parity_contract'] = BF16_CONTRACT
                rec['mfma'] = conv3x3_bf16_rate(device, batch)
            out[name] = rec
            del st, model, imgs
            torch.cuda.empty_cache()
        except Exception as exc:                            # the headline number must still print
            out[name] = {'error': '%s: %s' % (type(exc).__name__, exc)}
    out['wall_s'] = round(time.time() - t_start, 1)
    return out


def stage_traffic():
    """HBM bytes per decode-stage pass (all of its kernels) from the committed PMC profile"""
    path = os.path.join(ROOT, 'profiles', PMC_PROFILE)
    try:
        with open(path) as f:
            prof = json.load(f)
        if prof.get('batch') != BATCH:
            return None
        names = prof.get('stage_kernels') or [k for k in prof['kernels']
                                              if k.startswith(STAGE_PREFIXES)]
        return int(sum(prof['kernels'][k]['traffic_bytes_per_launch'] for k in names))
    except Exception:
        return None


def main_train(args, cfg_name, world, rank, local_rank, device, sync, barrier, rccl_ranks,
               rank_devices, launched_by):
    """--config r50-train: BASELINE config 5 as the measured job -- whole training iterations
    (forward, device target assignment + all-levels loss kernels, backward, gradient averaging,
    clipping, SGD), one rank per GPU, every rank its own synthetic batch, gradients averaged by
    DistributedDataParallel's bucketed all-reduce on RCCL overlapped with backward
    (iouaware/train.py; reference mmdet/apis/train.py:38-45, core/utils/dist_utils.py:46-57)."""
    from iouaware.train import train_step, wrap_ddp
    model, opt, img, ms, gtb, gtl = train_state(device, find=args.train_find, seed=rank)
    net = wrap_ddp(model, device_ids=[local_rank]) if world > 1 else model
    clip = dict(max_norm=35, norm_type=2)
    last = {}

    def step():
        last['log'] = train_step(net, opt, img, ms, gtb, gtl, grad_clip=clip)
    elapsed = timed_region(step, args.steps, args.warmup, world, sync, barrier, device)
    if rank == 0:
        print(json.dumps({
            'metric': 'training images/sec at 1333x800, IoU-aware RetinaNet R-50-FPN',
            'value': round(TRAIN_BATCH * world * args.steps / elapsed, 3), 'unit': 'img/s',
            'n_gpus': world, 'rccl_ranks': rccl_ranks, 'rank_devices': rank_devices,
            'launched_by': launched_by, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'fp32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE %s; 3x800x1344, random-init weights, fwd + HIP targets / '
                                   'losses + bwd + gradient all-reduce (DDP buckets on RCCL) + clip + '
                                   'SGD' % cfg_name,
                       'global_batch': TRAIN_BATCH * world, 'parallelism': 'dp%d' % world,
                       'miopen_find_mode': bool(args.train_find),
                       'loss': round(float(last['log']['loss']), 4)},
            'roofline': None, 'cpu_baseline': None,
            'note': 'roofline of the loss kernels: DESIGN.md section 4 / profiles/r03_train_pmc.json '
                    '(k_focal_nhwc fwd 0.70, bwd 0.81 of the HBM peak)'}))
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None,
                    help='ranks = GPUs of this node; default: WORLD_SIZE of the launcher, else 1')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='r50',
                    help='r50 = BASELINE config 2 (the headline); r101-bf16 = config 3; '
                         'x101-64x4d = config 4')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-live-pmc', action='store_true',
                    help='do not measure the decode stage HBM traffic with rocprofv3 child processes')
    ap.add_argument('--cpu-worker', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='print the cpu_baseline record alone (no GPU needed)')
    ap.add_argument('--no-train', action='store_true', help='skip the training sub-record')
    ap.add_argument('--no-pipeline', action='store_true', help='skip the image -> result sub-record')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the BASELINE config 1 / 3 / 4 sub-records of the default run')
    ap.add_argument('--train-find', action='store_true',
                    help='MIOpen find mode for the training sub-record (adds ~8 minutes)')
    ap.add_argument('--miopen-find', action='store_true',
                    help='MIOpen find mode (times the convolution algorithms per process: not '
                         'reproducible run to run; default: immediate mode)')
    ap.add_argument('--gemm-tune', default='frozen', choices=['frozen', 'heuristic', 'all'],
                    help="'frozen' (default): committed tuning table, nothing timed at run time")
    ap.add_argument('--no-fuse', action='store_true', help='keep the eager BN/ReLU/add kernels')
    ap.add_argument('--nchw', action='store_true', help='run the convolutions in NCHW')
    ap.add_argument('--no-winograd', action='store_true',
                    help='head 3x3 convolutions through MIOpen instead of the Winograd path')
    ap.add_argument('--rehearsal', action='store_true',
                    help='N > 1 without N GPUs: every rank drives GPU 0 (oversubscribed), process group on '
                         'gloo, records exchanged as host tensors -- launcher, pinning, the real step, the '
                         'all-gather, its rank interleave and the per-rank / exchange fields of the line, i.e. '
                         'everything of `--gpus 8` except RCCL itself.  Not a scaling measurement.')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU: fake detections on CPU, gloo backend -- launcher, process group, '
                         'timed region and result exchange only')
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return

    launched = 'WORLD_SIZE' in os.environ
    launched_by = os.environ.get('IA_BENCH_LAUNCHED_BY',
                                 'external launcher' if launched else 'single process')
    if not launched and (args.gpus or 1) > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks here
        if not args.dry_run and not args.rehearsal and torch.cuda.device_count() < args.gpus:
            raise SystemExit('bench.py --gpus %d: this node shows %d GPU(s)'
                             % (args.gpus, torch.cuda.device_count()))
        raise SystemExit(launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit('bench.py --gpus %d inside a launcher with WORLD_SIZE=%d'
                         % (args.gpus, world))
    if args.dry_run:
        device = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
        dev_index = 0 if args.rehearsal else local_rank
        torch.cuda.set_device(dev_index)
        device = torch.device('cuda', dev_index)
    host_group = args.dry_run or args.rehearsal            # gloo: collectives on host tensors
    rccl_ranks, rank_devices = 1, None
    # every rank on the cores of its GPU's NUMA node, a disjoint share each (no-op for one rank)
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    affinity = idist.pin_rank(local_rank, local_world, device_count=0 if args.dry_run else None,
                               same_device=args.rehearsal)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo' if host_group else 'nccl')
        if dist.get_world_size() != world:
            raise SystemExit('process group of %d ranks, launcher said %d'
                             % (dist.get_world_size(), world))
        rccl_ranks, rank_devices = group_ranks(torch.device('cpu') if args.rehearsal else device)
        if rccl_ranks != world or (not host_group and len(rank_devices) != world):
            raise SystemExit('collective sees %d ranks on devices %s, expected %d distinct'
                             % (rccl_ranks, rank_devices, world))
    sync = (lambda: None) if args.dry_run else torch.cuda.synchronize
    barrier = dist.barrier if host_group else (lambda: dist.barrier(device_ids=[local_rank]))
    reduce_device = torch.device('cpu') if args.rehearsal else device

    if args.dry_run:
        stepper = DryRunStepper(rank, world)
        elapsed = timed_region(lambda: stepper.step(timed=True), args.steps, args.warmup, world,
                               sync, barrier, device)
        ok = stepper.check() and stepper.calls == args.steps + args.warmup
        if rank == 0:
            print(json.dumps({'metric': 'dry run (no GPU work): launcher + process group + timed '
                                        'region + result exchange', 'dry_run': True,
                              'value': round(stepper.B * world * args.steps / elapsed, 3),
                              'unit': 'fake img/s', 'n_gpus': world, 'rccl_ranks': rccl_ranks,
                              'backend': 'gloo', 'steps': args.steps, 'warmup': args.warmup,
                              'exchange_ok': ok, 'launched_by': launched_by,
                              'affinity_rank0': affinity}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if not ok:
            raise SystemExit(1)
        return

    # Deterministic kernel selection: nothing is picked by timing at run time.
    #   library GEMMs  -> the committed tuning table (iouaware/tuning/hipblaslt_gfx950.json, found
    #                     offline by tools/tune_gemm.py), else the library heuristic's first result;
    #   MIOpen convs   -> immediate mode (the library's own ranking, no find-mode timing race).
    # --miopen-find / --gemm-tune bring the old pick-by-timing behaviour back for comparisons.
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    # (MIOpen's "deterministic" attribute is not an option: it leaves only the naive kernel, 25-64 ms
    # per convolution.  The strided convolutions whose fast library kernels add split-K partial
    # sums with atomics run as im2col / strided-batched GEMMs instead, csrc/im2col.hip; what is
    # left on MIOpen in the fp32 path is the 7x7 stem, whose immediate-mode kernel does not split.)
    from iouaware import ops
    ops.gemm_tuning(args.gemm_tune)

    cfg_name, backbone, batch, dtype_name = CONFIGS[args.config]
    if args.config == 'r50-train':
        return main_train(args, cfg_name, world, rank, local_rank, device, sync, barrier,
                          rccl_ranks, rank_devices, launched_by)
    dtype = getattr(torch, dtype_name)
    headline = args.config == 'r50'
    model = build_model(device, fuse=not args.no_fuse, channels_last=not args.nchw,
                        winograd=not args.no_winograd, backbone=backbone)
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    imgs = torch.randn(batch, 3, PAD_H, PAD_W, device=device, generator=g)
    if dtype != torch.float32:
        model, imgs = model.to(dtype), imgs.to(dtype)
    if not args.nchw:
        imgs = imgs.contiguous(memory_format=torch.channels_last)
    stepper = Stepper(model, imgs, world)
    stepper.host_exchange = bool(args.rehearsal)

    def step():
        stepper.step(timed=True)

    if rank == 0:
        stepper.stage_events(args.steps, skip=args.warmup)
    info = {}
    elapsed = timed_region(step, args.steps, args.warmup, world, sync, barrier, reduce_device, drain=stepper.drain,
                           info=info)
    in_step = stepper.stage_times_ms() if rank == 0 else []
    host_results = stepper.results
    multi = None
    if world > 1:
        # what a scaling run is read by: every rank's own clock, the exchange step, the rank interleave
        clocks = [None] * world
        dist.all_gather_object(clocks, (rank, info['local_busy_s'], info['local_elapsed_s']))
        multi = {'per_rank_img_s': [round(batch * args.steps / c[1], 2) for c in sorted(clocks)],
                 'per_rank_busy_s': [round(c[1], 4) for c in sorted(clocks)],
                 'exchange': stepper.exchange_record(args.steps),
                 'interleave_equals_zip_part_list': bool(stepper.interleave_matches_part_list())}

    wino = wino_roofline(stepper) if rank == 0 and dtype == torch.float32 else None
    if rank == 0:
        esz = 4 if dtype == torch.float32 else 2
        n_img = batch * world * args.steps
        ms_stage, ms_rowmax, nhwc = decode_stage_roofline(stepper)
        tname = 'float' if esz == 4 else 'unsigned short'
        rm_kernel = ('ia::k_rowmax_nhwc<%s, %d>' % (tname, 80 * esz // 16) if nhwc
                     else 'ia::k_rowmax<%s>' % tname)
        rowmax_bytes = ((64512000 + 806400) * esz // 4 + 806400) * batch
        stage_bytes = HEAD_BYTES_PER_IMAGE * esz // 4 * batch
        achieved = rowmax_bytes / (ms_rowmax * 1e-3) / 1e9
        b2b = stage_bytes / (ms_stage * 1e-3) / 1e9
        # the headline roofline figure: the stage's average duration INSIDE the timed steps
        ms_in_step = sum(in_step) / len(in_step) if in_step else ms_stage
        stage = stage_bytes / (ms_in_step * 1e-3) / 1e9
        out = {
            'metric': 'images/sec at 1333x800, IoU-aware RetinaNet R-50-FPN' if headline else
                      'images/sec at 1333x800, IoU-aware RetinaNet (%s)' % args.config,
            'value': round(n_img / elapsed, 3), 'unit': 'img/s', 'n_gpus': 1 if args.rehearsal else world,
            'rccl_ranks': 0 if args.rehearsal else rccl_ranks, 'rank_devices': rank_devices,
            'ranks': world, 'multi_rank': multi,
            'rehearsal': ('%d ranks oversubscribing ONE MI355X, process group on gloo, records exchanged as host '
                          'tensors: the N > 1 code path, NOT a scaling measurement' % world) if args.rehearsal else None,
            'launched_by': launched_by, 'affinity_rank0': affinity,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'fp32' if esz == 4 else 'bf16', 'data': 'synthetic',
            'kernel_selection': {'gemm': dict(mode=args.gemm_tune, **ops.gemm_table_stats()),
                                 'miopen': 'find mode' if args.miopen_find else 'immediate mode (stem only)'},
            'config': {'workload': ('IoU-aware RetinaNet R-50-FPN fp32, batch 8 per GPU, '
                                    '3x800x1344 (1333x800 padded to /32), random-init weights, '
                                    'whole inference path incl. NMS, device-to-host copy and bbox2result '
                                    '(the reference\'s simple_test: per-class arrays on the host; batch i\'s '
                                    'host part runs under batch i+1\'s device part, the last one inside '
                                    'the timed region)') if headline else
                                   ('BASELINE %s; 3x800x1344, random-init weights, whole '
                                    'inference path incl. NMS, D2H and bbox2result' % cfg_name),
                       'host_results': {'images': len(host_results) if host_results else 0,
                                        'classes_per_image': len(host_results[0]) if host_results else 0,
                                        'dets_image0': int(sum(r.shape[0] for r in host_results[0])) if host_results else 0},
                       'global_batch': batch * world, 'parallelism': 'dp%d' % world,
                       'dets_per_image': int(stepper.last[2].float().mean().item())},
            # SURVEY 8(d)'s unit: the decode stage (row-max + top-k + gather), cls + reg + iou
            # logits read once = 68 544 000 B per fp32 image; the row-max kernel alone below it
            'roofline': {'bound': 'hbm',
                         'kernel': 'decode stage: k_rowmax + k_sel_filter + k_sel_final + k_gather '
                                   '(SURVEY 8d unit)',
                         'timing': ('HIP events recorded by the library on the launch stream in front '
                                    'of the stage\'s first launch and behind its last one, in every '
                                    'step of the timed region (ia_profile_stage_events); average '
                                    'over the %d timed steps' % len(in_step)) if in_step else
                                   'back-to-back passes after the timed region (see back_to_back)',
                         'achieved': round(stage, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(stage / HBM_PEAK_GBS, 4),
                         'in_step_ms': {'mean': round(ms_in_step, 4),
                                        'min': round(min(in_step), 4), 'max': round(max(in_step), 4),
                                        'steps': len(in_step)} if in_step else None,
                         'back_to_back': {'timing': 'HIP events around 10 back-to-back passes of the '
                                                    'stage after the timed region, best of 5 '
                                                    '(inter-kernel gaps included)',
                                          'avg_launch_ms': round(ms_stage, 4),
                                          'achieved': round(b2b, 1),
                                          'frac': round(b2b / HBM_PEAK_GBS, 4)},
                         'traffic': stage_traffic() if headline else None,
                         'traffic_source': 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                           'separate passes, same batch-8 launches)' % PMC_PROFILE,
                         'bytes_per_launch': stage_bytes, 'avg_launch_ms': round(ms_in_step, 4),
                         'rowmax': {'kernel': rm_kernel.split('::')[1].split('<')[0],
                                    'bytes_per_launch': rowmax_bytes,
                                    'avg_launch_ms': round(ms_rowmax, 4),
                                    'achieved': round(achieved, 1),
                                    'frac': round(achieved / HBM_PEAK_GBS, 4),
                                    'traffic': rowmax_traffic(rm_kernel) if headline else None},
                         'wino': wino},
        }
        # (with the CPU baseline, i.e. in the full default run only: the profiling scripts under tools/
        # pass --no-cpu-baseline and run this file under rocprofv3 themselves -- never nest profilers)
        under_profiler = any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
        if world == 1 and headline and not args.no_live_pmc and not args.no_cpu_baseline and not under_profiler:
            live = live_stage_traffic()
            if live is not None:
                rf = out['roofline']
                rf['traffic_committed_profile'] = rf['traffic']
                rf['traffic'] = live['stage']
                rf['traffic_source'] = ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child '
                                        'processes (separate passes, --kernel-trace only) on the same batch-8 '
                                        'launches (tools/time_head.py), 2 x FETCH_SIZE + WRITE_SIZE')
                rf['traffic_kernels'] = live['kernels']
                if live['rowmax'] is not None:
                    rf['rowmax']['traffic'] = live['rowmax']
            else:
                out['roofline']['traffic_source'] += ' -- the live rocprofv3 pass of this run was not available'
        if world == 1 and headline and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
        if world == 1 and not args.no_pipeline:
            try:
                out['pipeline'] = pipeline_record(model, device, batch)
            except Exception as exc:
                out['pipeline'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        else:
            out['pipeline'] = None
        freed = False
        if world == 1 and headline and not args.no_train:
            del stepper, model, imgs
            freed = True
            torch.cuda.empty_cache()
            try:
                out['train'] = train_record(device, find=args.train_find)
            except Exception as exc:                     # the headline number must still print
                out['train'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        else:
            out['train'] = None
        if world == 1 and headline and not args.no_other_configs:
            if not freed:
                del stepper, model, imgs
            torch.cuda.empty_cache()
            out['other_configs'] = other_configs_record(device)
        else:
            out['other_configs'] = None
        print(json.dumps(out))
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
