"""Process-group launch contract and the multi-GPU result exchange.

One process per GPU (reference tools/dist_test.sh:9-10, mmdet/apis/env.py:13-50);
`backend='nccl'` on PyTorch-ROCm IS RCCL over xGMI, so the reference's
`dist_params = dict(backend='nccl')` keeps working.  Images shard by rank the
way DistributedSampler(shuffle=False) does (rank r takes indices r, r+W, ...;
reference mmdet/datasets/loader/sampler.py:18-35).  The reference gathers
results through pickle files on a shared filesystem plus barriers
(tools/test.py:63-102); here each rank contributes a fixed-size record per
image -- max_per_img x (x1,y1,x2,y2,score) fp32 + label + count -- to ONE
all_gather (about 2.4 KB per image, latency bound), and rank order is undone
with the same interleave `zip(*part_list)` the reference uses.
"""
import os

import torch
import torch.distributed as dist


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """RANK / WORLD_SIZE / MASTER_* come from the launcher (torch.distributed.run)."""
    if launcher != 'pytorch':
        raise ValueError('Invalid launcher type: {} (only "pytorch" is built)'.format(launcher))
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    if backend == 'nccl':
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)
    return rank, dist.get_world_size()


# ------------------------------------------------------------------ rank -> cores of the GPU's NUMA node
def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(','):
        if '-' in part:
            lo, hi = part.split('-')
            out += list(range(int(lo), int(hi) + 1))
        elif part:
            out.append(int(part))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa_node(pci_bus_id, sysfs='/sys'):
    """NUMA node the GPU with this PCI address ('0000:c1:00.0') hangs off, or -1"""
    if not pci_bus_id:
        return -1
    txt = _read(os.path.join(sysfs, 'bus/pci/devices', pci_bus_id.lower(), 'numa_node'))
    try:
        return int(txt)
    except (TypeError, ValueError):
        return -1


def _physical_cores(cpus, sysfs='/sys'):
    """one hardware thread per physical core, ascending"""
    seen, out = set(), []
    for c in sorted(cpus):
        sib = _read(os.path.join(sysfs, 'devices/system/cpu/cpu%d/topology/thread_siblings_list' % c))
        key = tuple(sorted(_parse_cpulist(sib))) if sib else (c,)
        if key not in seen:
            seen.add(key)
            out.append(c)
    return out


def rank_cpu_plan(local_rank, local_world, numa_nodes, allowed, sysfs='/sys'):
    """The cores rank `local_rank` of `local_world` ranks on this node should run on.
    numa_nodes[r] = NUMA node of rank r's GPU (-1: unknown).  Ranks whose GPUs share a NUMA node
    split that node's physical cores (restricted to `allowed`, the launcher's affinity mask) into
    equal contiguous shares; a rank with an unknown node takes the same share of ALL allowed
    cores.  -> (sorted cpu ids incl. the SMT siblings of the chosen cores, description)"""
    allowed = set(allowed)
    node = numa_nodes[local_rank] if 0 <= local_rank < len(numa_nodes) else -1
    cpus = None
    if node >= 0:
        txt = _read(os.path.join(sysfs, 'devices/system/node/node%d/cpulist' % node))
        if txt:
            cpus = [c for c in _parse_cpulist(txt) if c in allowed]
    if cpus:
        peers = [r for r in range(local_world) if r < len(numa_nodes) and numa_nodes[r] == node]
        how = 'NUMA node %d' % node
    else:
        cpus, peers, how = sorted(allowed), list(range(local_world)), 'all allowed cores (GPU NUMA node unknown)'
    cores = _physical_cores(cpus, sysfs)
    share = max(1, len(cores) // max(1, len(peers)))
    k = peers.index(local_rank) if local_rank in peers else 0
    mine = cores[k * share:(k + 1) * share] or cores[-share:]
    out = set()
    for c in mine:
        sib = _read(os.path.join(sysfs, 'devices/system/cpu/cpu%d/topology/thread_siblings_list' % c))
        out.update(x for x in (_parse_cpulist(sib) if sib else [c]) if x in allowed)
    return sorted(out), '%d of %d physical cores of %s (share %d of %d)' % (len(mine), len(cores), how,
                                                                       k + 1, len(peers))


def pin_rank(local_rank, local_world, device_count=None, sysfs='/sys', apply=True, same_device=False):
    """Bind this process (and the threads it starts later) to the cores of the NUMA node its GPU
    hangs off, a disjoint share per rank, and size the host thread pools to it.  One Python feeder per
    GPU issues ~140 launches per 20 ms step; eight unpinned feeders on a two-socket host migrate
    between sockets and share cores with each other's helper threads.  The
    reference leaves placement to the launcher (tools/dist_test.sh:9-10).  -> record for the
    benchmark line; never raises (an unreadable topology leaves the affinity alone).
    same_device: every rank drives device 0 (bench.py --rehearsal): the ranks share that GPU's NUMA node."""
    rec = {'local_rank': int(local_rank), 'pinned': False}
    try:
        allowed = os.sched_getaffinity(0)
        nodes = []
        n = device_count if device_count is not None else (torch.cuda.device_count() if torch.cuda.is_available() else 0)
        for r in range(local_world):
            bus = None
            d = 0 if same_device else r
            if d < n:
                pr = torch.cuda.get_device_properties(d)
                if hasattr(pr, 'pci_bus_id'):
                    bus = '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id,
                                                getattr(pr, 'pci_device_id', 0))
            nodes.append(gpu_numa_node(bus, sysfs))
        cpus, how = rank_cpu_plan(local_rank, local_world, nodes, allowed, sysfs)
        rec.update(numa_node=nodes[local_rank] if local_rank < len(nodes) else -1, cpus=len(cpus),
                   first_cpu=cpus[0] if cpus else None, plan=how)
        if apply and cpus and local_world > 1:
            os.sched_setaffinity(0, cpus)
            threads = max(1, min(8, len(cpus) // 2 or 1))
            os.environ['OMP_NUM_THREADS'] = str(threads)
            torch.set_num_threads(threads)
            rec.update(pinned=True, host_threads=threads)
    except Exception as exc:                                 # topology files differ between hosts
        rec['error'] = '%s: %s' % (type(exc).__name__, exc)
    return rec


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(num_samples, rank, world_size):
    """indices of this rank, padded by wrap-around so every rank has the same count."""
    per = (num_samples + world_size - 1) // world_size
    total = per * world_size
    idx = list(range(num_samples))
    idx += idx[:total - num_samples]
    return idx[rank:total:world_size]


def pack_detections(dets, labels, num):
    """(B,M,5) f32, (B,M) i32, (B,) i32 -> (B, M*6+1) f32 records.  Labels and counts
    are exactly representable in fp32 (< 2^24)."""
    B, M, _ = dets.shape
    rec = torch.empty((B, M * 6 + 1), dtype=torch.float32, device=dets.device)
    rec[:, :M * 5] = dets.reshape(B, M * 5)
    rec[:, M * 5:M * 6] = labels.to(torch.float32)
    rec[:, M * 6] = num.to(torch.float32)
    return rec


def unpack_detections(rec, max_per_img):
    M = max_per_img
    B = rec.shape[0]
    dets = rec[:, :M * 5].reshape(B, M, 5)
    labels = rec[:, M * 5:M * 6].to(torch.int32)
    num = rec[:, M * 6].to(torch.int32)
    return dets, labels, num


_GATHER_BUF = {}


def _gather_buffer(world, rec):
    """persistent (world, B, rec_len) receive buffer: one allocation per record shape, not one
    list of `world` tensors per step"""
    key = (world, tuple(rec.shape), rec.dtype, rec.device)
    buf = _GATHER_BUF.get(key)
    if buf is None:
        buf = _GATHER_BUF[key] = torch.empty((world,) + tuple(rec.shape), dtype=rec.dtype,
                                             device=rec.device)
    return buf


def all_gather_detections(dets, labels, num, num_samples=None, force_collective=False):
    """Gather every rank's per-image detections on every rank, in dataset order.

    Each rank passes its local batch (same B on every rank).  Returns
    (dets (W*B,M,5), labels, num) interleaved rank-major -> dataset order, truncated to
    num_samples when given.  ONE collective per call: `all_gather_into_tensor` into a
    pre-allocated (W, B, M*6+1) buffer (RCCL on GPUs: a single ring / direct all-gather of
    W x B x 2.4 KB; gloo on CPU), falling back to the list form where a backend lacks it.
    The result never aliases the persistent receive buffer (the rank interleave is a copy when
    B > 1, an explicit clone otherwise), so it stays valid across later calls.
    force_collective: issue the collective even in a one-rank group (tests: RCCL on one GPU).
    """
    rank, world = get_dist_info()
    M = dets.shape[1]
    rec = pack_detections(dets, labels, num).contiguous()
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        allrec = rec
    else:
        buf = _gather_buffer(world, rec)
        try:
            dist.all_gather_into_tensor(buf, rec)
        except (RuntimeError, NotImplementedError, AttributeError):
            dist.all_gather(list(buf.unbind(0)), rec)
        # sample i of rank r is dataset index i*world + r (reference tools/test.py:95-99)
        allrec = buf.transpose(0, 1).reshape(world * rec.shape[0], rec.shape[1])
        if allrec.data_ptr() == buf.data_ptr():
            # B == 1 per rank (the reference's distributed testing, imgs_per_gpu=1) or a one-rank
            # group: the "interleave" is a view of the persistent buffer, which the next call
            # overwrites while the caller may still hold these detections
            allrec = allrec.clone()
    if num_samples is not None:
        allrec = allrec[:num_samples]
    return unpack_detections(allrec, M)
