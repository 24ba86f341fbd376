"""CPU, world_size 2, gloo: the N>1 result exchange.  Each rank owns
the images rank, rank+W, ... (DistributedSampler order); one all_gather of
fixed-size records returns all detections in dataset order on every rank."""
import os
import json
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG = os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_dets(i, M=100):
    """deterministic per-image detections"""
    rs = np.random.RandomState(1000 + i)
    k = int(rs.randint(0, M + 1))
    dets = np.zeros((M, 5), np.float32)
    dets[:k] = rs.uniform(0, 1000, (k, 5)).astype(np.float32)
    labels = np.full(M, -1, np.int32)
    labels[:k] = rs.randint(0, 80, k)
    return dets, labels, k


def _worker(rank, world, port, num_samples, ret):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from iouaware import dist as idist
    r, w = idist.init_dist('pytorch', backend='gloo')
    assert (r, w) == (rank, world) and idist.get_dist_info() == (rank, world)
    mine = idist.shard_indices(num_samples, rank, world)
    d, l, n = zip(*[_fake_dets(i) for i in mine])
    dets = torch.from_numpy(np.stack(d))
    labels = torch.from_numpy(np.stack(l))
    num = torch.tensor(n, dtype=torch.int32)
    D, L, N = idist.all_gather_detections(dets, labels, num, num_samples=num_samples)
    ok = D.shape[0] == num_samples
    for i in range(num_samples):
        ed, el, ek = _fake_dets(i)
        ok &= bool(np.array_equal(D[i].numpy(), ed) and np.array_equal(L[i].numpy(), el)
                   and int(N[i]) == ek)
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_wrap_around():
    sys.path.insert(0, PKG)
    from iouaware import dist as idist
    assert idist.shard_indices(5, 0, 2) == [0, 2, 4]
    assert idist.shard_indices(5, 1, 2) == [1, 3, 0]        # padded by wrap-around
    assert idist.shard_indices(8, 3, 8) == [3]


def test_all_gather_detections_world2_gloo():
    world, num_samples = 2, 5
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_samples, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _alias_worker(rank, world, port, ret):
    """B = 1 per rank, two calls: the first result must survive the second call (the
    rank interleave of a (W, 1, rec) buffer is a view of the persistent receive buffer)"""
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from iouaware import dist as idist
    idist.init_dist('pytorch', backend='gloo')
    outs = []
    for call in range(2):
        d, l, n = _fake_dets(10 * call + rank)
        outs.append(idist.all_gather_detections(torch.from_numpy(d[None]), torch.from_numpy(l[None]),
                                                torch.tensor([n], dtype=torch.int32)))
    ok = True
    for call in range(2):
        D, L, N = outs[call]
        for r in range(world):
            ed, el, ek = _fake_dets(10 * call + r)
            ok &= bool(np.array_equal(D[r].numpy(), ed) and np.array_equal(L[r].numpy(), el)
                       and int(N[r]) == ek)
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_result_does_not_alias_the_receive_buffer():
    world, port = 2, _free_port()
    ret = mp.get_context('spawn').Manager().dict()
    mp.spawn(_alias_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_pack_unpack_roundtrip():
    sys.path.insert(0, PKG)
    from iouaware import dist as idist
    d, l, k = zip(*[_fake_dets(i, 7) for i in range(3)])
    dets, labels = torch.from_numpy(np.stack(d)), torch.from_numpy(np.stack(l))
    num = torch.tensor(k, dtype=torch.int32)
    D, L, N = idist.unpack_detections(idist.pack_detections(dets, labels, num), 7)
    assert torch.equal(D, dets) and torch.equal(L, labels) and torch.equal(N, num)


def _grad_worker(rank, world, port, ret):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from iouaware import dist as idist
    from iouaware.train import allreduce_grads
    idist.init_dist('pytorch', backend='gloo')
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    net[0].bias.requires_grad_(False)                       # frozen parameters are skipped
    ok = True
    for kw in (dict(), dict(bucket_size_mb=1), dict(coalesce=False)):
        for p in net.parameters():
            p.grad = None
        x = torch.full((4, 7), float(rank + 1))
        net(x).sum().backward()
        mine = [p.grad.clone() for p in net.parameters() if p.grad is not None]
        allreduce_grads(net, **kw)
        # gradients are linear in x here only through the first layer; compare with an explicit
        # gather of every rank's gradient
        gathered = [[torch.zeros_like(g) for _ in range(world)] for g in mine]
        for g, lst in zip(mine, gathered):
            dist.all_gather(lst, g)
        got = [p.grad for p in net.parameters() if p.grad is not None]
        for g, lst in zip(got, gathered):
            ok &= bool(torch.allclose(g, sum(lst) / world, rtol=1e-6, atol=1e-7))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_grads_world2():
    """T5: gradient averaging over ranks (reference dist_utils.py:9-43), gloo world size 2"""
    world, port = 2, _free_port()
    ret = mp.get_context('spawn').Manager().dict()
    mp.spawn(_grad_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


# ------------------------------------------------------------------ bench.py's own N > 1 path
def _bench_worker(rank, world, port, ret):
    """drives bench.py's Stepper / timed_region (barriers, MAX over ranks of the elapsed time,
    the per-step all-gather and its rank interleave) with the gloo backend on fake per-rank
    detections -- the code the driver runs on 2/4/8 GPUs, executed before it ever does"""
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import time
    import bench
    from iouaware import dist as idist
    idist.init_dist('pytorch', backend='gloo')
    B = 3

    class FakeStepper(bench.Stepper):
        def __init__(self):
            self.world, self.calls, self.last = world, 0, None

        def local_detections(self, timed=False):
            # image j of this rank is dataset index j * world + rank (DistributedSampler order)
            self.calls += 1
            if rank == 1:
                time.sleep(0.02)                  # the slow rank sets the step time
            d, l, n = zip(*[_fake_dets(j * world + rank) for j in range(B)])
            return (torch.from_numpy(np.stack(d)), torch.from_numpy(np.stack(l)),
                    torch.tensor(n, dtype=torch.int32), None, None, None)

    st = FakeStepper()
    steps, warmup = 4, 2
    elapsed = bench.timed_region(lambda: st.step(timed=True), steps, warmup, world, lambda: None,
                                 dist.barrier, torch.device('cpu'))
    ok = st.calls == steps + warmup
    D, L, N = st.last[:3]
    ok &= D.shape[0] == world * B
    for i in range(world * B):
        ed, el, ek = _fake_dets(i)
        ok &= bool(np.array_equal(D[i].numpy(), ed) and np.array_equal(L[i].numpy(), el)
                   and int(N[i]) == ek)
    ok &= elapsed >= steps * 0.02                  # MAX over ranks: rank 0 reports rank 1's time too
    ret[rank] = (ok, round(elapsed, 6))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timed_region_and_exchange_world2():
    world, port = 2, _free_port()
    ret = mp.get_context('spawn').Manager().dict()
    mp.spawn(_bench_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world)), dict(ret)
    assert ret[0][1] == ret[1][1]                  # every rank holds the same (maximum) time


def _run_bench(argv, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()


def test_bench_gpus_8_dry_run():
    """the node size the driver's scaling run uses: eight ranks, rank interleave of the gathered
    records over 8 shards"""
    rc, rec, err = _run_bench(['--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert rc == 0, err[-2000:]
    assert rec['n_gpus'] == 8 and rec['rccl_ranks'] == 8 and rec['exchange_ok'] is True


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no launcher) must start 2 ranks -- the
    reference's launcher takes the GPU count and spawns (tools/dist_test.sh:7-10).  --dry-run:
    gloo backend, fake detections; the launcher, the group-size checks, timed_region and the
    exchange are the code a real N-GPU run executes."""
    rc, rec, err = _run_bench(['--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1'])
    assert rc == 0, err[-2000:]
    assert rec['n_gpus'] == 2 and rec['rccl_ranks'] == 2 and rec['exchange_ok'] is True
    assert rec['launched_by'].startswith('bench.py --gpus 2')
    assert rec['steps'] == 3 and rec['warmup'] == 1


def test_bench_under_the_drivers_launcher_and_flag_mismatch():
    """the driver's form: torch.distributed.run ... bench.py --gpus N; a --gpus that disagrees
    with the launcher's WORLD_SIZE is an error, not a silent single-rank run"""
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
            '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
            os.path.join(ROOT, 'bench.py'), '--dry-run', '--steps', '2', '--warmup', '1']
    p = subprocess.run(base + ['--gpus', '2'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    import json
    rec = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert p.returncode == 0 and rec['n_gpus'] == 2 and rec['launched_by'] == 'external launcher'
    base[base.index('--master-port') + 1] = str(_free_port())
    p = subprocess.run(base + ['--gpus', '4'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert 'WORLD_SIZE=2' in p.stderr.decode() + p.stdout.decode()


def test_bench_single_process_dry_run_unchanged():
    rc, rec, err = _run_bench(['--dry-run', '--steps', '2', '--warmup', '1'])
    assert rc == 0 and rec['n_gpus'] == 1 and rec['launched_by'] == 'single process', err[-500:]


def _none_loss_worker(rank, world, port, ret):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from iouaware import dist as idist
    from iouaware.train import train_step
    idist.init_dist('pytorch', backend='gloo')

    class NoLoss(torch.nn.Module):                 # head.loss returned None (reference :362-363)
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

        def forward(self, img, img_meta, return_loss=True, **kw):
            return None
    net = NoLoss()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    try:
        train_step(net, opt, torch.zeros(1, 3, 8, 8), [{}], [None], [None], allreduce=True)
        ret[rank] = 'returned'
    except RuntimeError as exc:
        ret[rank] = 'raised' if 'distributed step' in str(exc) else 'other: %s' % exc
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_none_loss_raises_instead_of_hanging_the_collective():
    """returning early on one rank would leave the others blocked in the all-reduce"""
    world, port = 2, _free_port()
    ret = mp.get_context('spawn').Manager().dict()
    mp.spawn(_none_loss_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 'raised', 1: 'raised'}
    # single process: None propagates like in the reference's runner
    sys.path.insert(0, PKG)
    from iouaware.train import train_step

    class NoLoss(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

        def forward(self, img, img_meta, return_loss=True, **kw):
            return None
    net = NoLoss()
    assert train_step(net, torch.optim.SGD(net.parameters(), lr=0.1), torch.zeros(1, 3, 8, 8),
                      [{}], [None], [None]) is None


def _fake_topology(root, nodes=2, cores_per_node=8, gpus=8):
    """a sysfs tree: `nodes` NUMA nodes of `cores_per_node` physical cores with two hardware
    threads each (cpu c and c + nodes * cores_per_node), GPU g on node g // (gpus / nodes)"""
    ncore = nodes * cores_per_node
    for n in range(nodes):
        d = os.path.join(root, 'devices/system/node/node%d' % n)
        os.makedirs(d)
        lo = n * cores_per_node
        with open(os.path.join(d, 'cpulist'), 'w') as f:
            f.write('%d-%d,%d-%d\n' % (lo, lo + cores_per_node - 1, ncore + lo, ncore + lo + cores_per_node - 1))
    for c in range(2 * ncore):
        d = os.path.join(root, 'devices/system/cpu/cpu%d/topology' % c)
        os.makedirs(d)
        with open(os.path.join(d, 'thread_siblings_list'), 'w') as f:
            f.write('%d,%d\n' % (c % ncore, c % ncore + ncore))
    bus = []
    for g in range(gpus):
        name = '0000:%02x:00.0' % (0x10 + g)
        d = os.path.join(root, 'bus/pci/devices', name)
        os.makedirs(d)
        with open(os.path.join(d, 'numa_node'), 'w') as f:
            f.write('%d\n' % (g // (gpus // nodes)))
        bus.append(name)
    return bus


def test_rank_cpu_plan_splits_the_numa_node_of_each_gpu(tmp_path):
    """eight ranks on a two-socket host -- every rank on the cores of ITS GPU's
    NUMA node, the four ranks of a node on disjoint shares, SMT siblings kept together"""
    from iouaware import dist as idist
    root = str(tmp_path)
    bus = _fake_topology(root)
    nodes = [idist.gpu_numa_node(b, root) for b in bus]
    assert nodes == [0, 0, 0, 0, 1, 1, 1, 1]
    assert idist.gpu_numa_node('0000:ff:00.0', root) == -1 and idist.gpu_numa_node(None, root) == -1
    allowed = set(range(32))
    plans = [idist.rank_cpu_plan(r, 8, nodes, allowed, root)[0] for r in range(8)]
    assert plans[0] == [0, 1, 16, 17] and plans[3] == [6, 7, 22, 23]
    assert plans[4] == [8, 9, 24, 25] and plans[7] == [14, 15, 30, 31]
    flat = [c for p in plans for c in p]
    assert len(flat) == len(set(flat)) == 32                      # disjoint, everything used
    # a launcher that restricted the affinity mask: shares come out of the allowed cores only
    cpus, how = idist.rank_cpu_plan(1, 2, [0, 0], {0, 1, 2, 3, 16, 17, 18, 19}, root)
    assert cpus == [2, 3, 18, 19] and 'NUMA node 0' in how
    # unknown NUMA node (-1, e.g. a VM): an equal share of all allowed cores
    cpus, how = idist.rank_cpu_plan(1, 2, [-1, -1], allowed, root)
    assert cpus == list(range(8, 16)) + list(range(24, 32)) and 'unknown' in how
    # pin_rank never raises and leaves a single rank alone
    rec = idist.pin_rank(0, 1, device_count=0, sysfs=root)
    assert rec['pinned'] is False and 'error' not in rec


def test_bench_dry_run_ranks_are_pinned_to_disjoint_cores():
    """two gloo ranks of `bench.py --dry-run`: rank 0 reports its share of the host's cores"""
    if len(os.sched_getaffinity(0)) < 4:
        pytest.skip('needs at least four cores')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run',
                          '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    aff = rec['affinity_rank0']
    assert aff['pinned'] is True and aff['cpus'] >= 1 and 'share 1 of 2' in aff['plan'], aff
    assert aff['cpus'] <= len(os.sched_getaffinity(0)) // 2 + 1
