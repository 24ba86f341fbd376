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
