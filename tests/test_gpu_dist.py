"""GPU: the result exchange on the RCCL backend itself (`backend='nccl'` on ROCm), as far as a
one-GPU box allows: a one-rank process group, the collective forced -- RCCL accepts the record
buffer's shape / dtype, the interleave and unpacking are right on device tensors.  World sizes
> 1 are covered on CPU with gloo (tests/test_dist_gloo.py); N > 1 on hardware is the driver's."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(180)
def test_all_gather_detections_on_rccl_one_rank():
    from iouaware import dist as idist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0',
                      WORLD_SIZE='1', LOCAL_RANK='0')
    rank, world = idist.init_dist('pytorch', backend='nccl')
    try:
        assert (rank, world) == (0, 1) and dist.get_backend() == 'nccl'
        g = torch.Generator(device='cuda').manual_seed(0)
        dets = torch.rand(8, 100, 5, device='cuda', generator=g) * 1000
        labels = torch.randint(0, 80, (8, 100), device='cuda', generator=g, dtype=torch.int32)
        num = torch.randint(0, 101, (8,), device='cuda', generator=g, dtype=torch.int32)
        D, L, N = idist.all_gather_detections(dets, labels, num, force_collective=True)
        torch.cuda.synchronize()
        assert torch.equal(D, dets) and torch.equal(L, labels) and torch.equal(N, num)
        D2, _, _ = idist.all_gather_detections(dets, labels, num, num_samples=5, force_collective=True)
        assert D2.shape[0] == 5
        # the gradient all-reduce of the training path on RCCL (one rank: identity)
        from iouaware.train import allreduce_grads
        net = torch.nn.Linear(7, 3).cuda()
        net(torch.ones(2, 7, device='cuda')).sum().backward()
        before = [p.grad.clone() for p in net.parameters()]
        allreduce_grads(net)
        assert all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
        dist.barrier(device_ids=[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_over_the_fused_training_route_on_rccl_one_rank():
    """DistributedDataParallel (bucketed all-reduce on RCCL, overlapped with backward) around a
    detector on the fused training route: its reducer hooks fire from the library's autograd
    nodes, and with one rank the gradients are those of the bare model"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    import bench
    import iouaware
    import synth
    from iouaware import dist as idist
    from iouaware.config import ConfigDict
    from iouaware.fuse import fuse_inference
    from iouaware.train import parse_losses
    from torch.nn.parallel import DistributedDataParallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0',
                      WORLD_SIZE='1', LOCAL_RANK='0')
    idist.init_dist('pytorch', backend='nccl')
    try:
        gts, gls = synth.train_targets(11, 2, 250, 317, max_gt=6)
        gtb = [torch.from_numpy(x).cuda() for x in gts]
        gtl = [torch.from_numpy(x).cuda() for x in gls]
        metas = [synth.img_meta(250, 317, 256, 320) for _ in range(2)]
        img = torch.from_numpy(synth.e2e_image(3, 2, 256, 320, 250, 317)).cuda() \
            .contiguous(memory_format=torch.channels_last)
        grads = {}
        for mode in ('bare', 'ddp'):
            torch.manual_seed(0)
            model = iouaware.build_detector(ConfigDict(bench.MODEL), train_cfg=ConfigDict(bench.TRAIN_CFG),
                                            test_cfg=ConfigDict(bench.TEST_CFG))
            state = model.state_dict()
            synth.e2e_fill_state(state, 7)
            model.load_state_dict(state)
            model = model.cuda().train()
            fuse_inference(model, winograd=True, train=True)
            model = model.to(memory_format=torch.channels_last)
            net = DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False) \
                if mode == 'ddp' else model
            loss, _ = parse_losses(net(img, metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl))
            loss.backward()
            torch.cuda.synchronize()
            grads[mode] = {k: p.grad.clone() for k, p in model.named_parameters() if p.requires_grad}
            assert all(g is not None for g in grads[mode].values())
        assert set(grads['bare']) == set(grads['ddp'])
        a = torch.cat([grads['ddp'][k].flatten() for k in grads['bare']])
        b = torch.cat([g.flatten() for g in grads['bare'].values()])
        # (norm-wise: the GEMM kernels are chosen by timing per process state, ReLU masks may flip)
        assert float((a - b).norm() / b.norm()) < 5e-3
    finally:
        dist.destroy_process_group()


def _run_ranks(world, extra_env=None, iters=25, timeout=420, attempts=3):
    """(a rendezvous port that was free a moment ago can be taken by the time rank 0 binds it:
    EADDRINUSE is retried on another port)"""
    for k in range(attempts):
        try:
            return _run_ranks_once(world, extra_env, iters, timeout)
        except AssertionError as exc:
            if 'EADDRINUSE' not in str(exc) or k == attempts - 1:
                raise


def _run_ranks_once(world, extra_env, iters, timeout):
    import json
    import subprocess
    import sys
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(r), IA_OVERSUB_ITERS=str(iters))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), 'oversub_worker.py')],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    recs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        if p.returncode != 0:
            for q in procs:
                q.kill()
        assert p.returncode == 0, (out[-1000:], err[-3000:])
        recs.append(json.loads([l for l in out.splitlines() if l.startswith('{')][-1]))
    return recs


@pytest.mark.timeout(600)
def test_ranks_share_one_gpu():
    """three ranks oversubscribe ONE MI355X -- each runs the whole post-conv
    path (fused row-max / filter launch with its inter-workgroup flags, two-stream decode stage,
    lazy NMS) on its own full-size batch while the other ranks' kernels are resident, and
    exchanges its records (gloo: RCCL refuses two ranks on one device).  Every call returns the
    bits the rank computed alone; the number of calls whose fused launch gave up waiting and fell
    back to the dense selection is reported (a correct, slower call -- not an error)."""
    recs = _run_ranks(3)
    assert [r['rank'] for r in recs] == [0, 1, 2]
    assert all(r['mismatches'] == 0 for r in recs), recs
    print('\\n[oversubscription] 3 ranks on one GPU, %d calls each: fused-launch fallbacks per rank %s'
          % (recs[0]['iterations'], [r['fused_fallbacks'] for r in recs]))
    out = os.path.join(os.path.dirname(__file__), '..', 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'oversubscription_report.txt'), 'a') as fh:
            fh.write('3 ranks on one GPU: %s\\n' % recs)


@pytest.mark.timeout(600)
def test_fused_launch_on_a_quarter_of_the_chip():
    """the same workers with the process restricted to 64 of the 256 CUs (HSA_CU_MASK): the filter
    workgroups of the fused launch wait on flags written by row-max workgroups that now queue for a
    quarter of the wavefront slots -- real starvation instead of a forced timeout.  Results
    unchanged; fallbacks reported."""
    mask = {'HSA_CU_MASK': '0:0-63'}
    recs = _run_ranks(2, extra_env=mask, iters=15)
    assert all(r['mismatches'] == 0 for r in recs), recs
    print('\\n[cu mask] 2 ranks on 64 CUs: fused-launch fallbacks per rank %s' % [r['fused_fallbacks'] for r in recs])
    out = os.path.join(os.path.dirname(__file__), '..', 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'oversubscription_report.txt'), 'a') as fh:
            fh.write('2 ranks, HSA_CU_MASK=0:0-63: %s\\n' % recs)


@pytest.mark.timeout(900)
def test_bench_gpus_8_rehearsal_on_one_gpu():
    """`bench.py --gpus 8` itself -- its launcher (torch.distributed.run, 8 ranks), the per-rank
    core pinning, the real step (network + post-conv path, batch 8 per rank), the all-gather of the records and the
    rank interleave -- with the 8 ranks oversubscribing the ONE GPU of this box (--rehearsal: gloo, host records;
    RCCL refuses two ranks on a device).  The gathered records of the last step equal the reference's
    `[res for tup in zip(*part_list) for res in tup]` (tools/test.py:95-99), and the line carries what a scaling
    run is read by: every rank's own img/s, the exchange time, the number of ranks."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), '..')
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--rehearsal', '--steps', '3',
                        '--warmup', '1', '--no-cpu-baseline', '--no-live-pmc', '--no-train', '--no-pipeline',
                        '--no-other-configs'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=850)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert p.returncode == 0 and lines, p.stderr.decode()[-3000:]
    rec = json.loads(lines[-1])
    assert rec['ranks'] == 8 and rec['n_gpus'] == 1 and rec['rehearsal']
    assert rec['launched_by'].startswith('bench.py --gpus 8')
    m = rec['multi_rank']
    assert m['interleave_equals_zip_part_list'] is True
    assert len(m['per_rank_img_s']) == 8 and all(v > 0 for v in m['per_rank_img_s'])
    assert m['exchange']['steps'] == 3 and m['exchange']['ms_mean'] > 0
    assert m['exchange']['bytes_per_rank'] == 8 * 601 * 4          # 8 images x (100 x 6 + 1) words
    assert rec['config']['global_batch'] == 64 and rec['config']['parallelism'] == 'dp8'
    out = os.path.join(root, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'rehearsal_gpus8.json'), 'w') as fh:
            fh.write(lines[-1] + '\n')


def test_pin_rank_reads_the_real_topology():
    """the rank -> cores plan on THIS host's sysfs (not applied): the GPU's PCI address resolves to a
    NUMA node (or -1 in a VM), the plan is a non-empty subset of the allowed cores, two ranks that
    would share the GPU's node get disjoint shares"""
    from iouaware import dist as idist
    allowed = os.sched_getaffinity(0)
    r0 = idist.pin_rank(0, 2, device_count=1, apply=False)
    r1 = idist.pin_rank(1, 2, device_count=1, apply=False)
    assert 'error' not in r0 and 'error' not in r1, (r0, r1)
    assert r0['pinned'] is False and r0['cpus'] >= 1 and r0['cpus'] <= len(allowed)
    assert isinstance(r0['numa_node'], int) and r0['numa_node'] >= -1
    print('\n[pin_rank] rank 0 of 2: %s' % r0)
    print('[pin_rank] rank 1 of 2: %s' % r1)
