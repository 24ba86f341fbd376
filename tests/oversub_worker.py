"""One rank of tests/test_gpu_dist.py::test_ranks_share_one_gpu: several processes drive the SAME
MI355X at once (the situation of N > 1 from one GPU's point of view: foreign kernels -- another
rank's streams, a collective's kernel -- resident while the fused row-max / filter launch waits on
its flags).  Process group on gloo (RCCL refuses two ranks on one device), records exchanged as
host tensors; everything else is the product path.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', 'iou-aware-single-stage-object-detector_amd'))
sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))       # gpu_util checks the anchor generator against the oracle's
import synth  # noqa: E402
import gpu_util as G  # noqa: E402


def main():
    from iouaware import ops, dist as idist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    iters = int(os.environ.get('IA_OVERSUB_ITERS', '25'))
    dist.init_process_group(backend='gloo')
    torch.cuda.set_device(0)
    ph, pw, B = 800, 1344, 4
    geom, _ = G.geometry(ph, pw, 1000)
    cls, reg, iou = synth.head_outputs(100 + rank, B, ph, pw, 'A')
    dev = [[t.contiguous(memory_format=torch.channels_last) for t in G.to_dev(x)] for x in (cls, reg, iou)]
    shapes, sfs = [(800, 1333, 3)] * B, [1.0] * B

    def run():
        return ops.get_bboxes(geom, *dev, shapes, sfs, True, 0.05, 0.5, 100)
    # every rank's expected result, computed while the others wait at the barrier
    want = None
    for r in range(world):
        dist.barrier()
        if r == rank:
            want = [t.clone() for t in run()]
            torch.cuda.synchronize()
    dist.barrier()
    bad = 0
    for it in range(iters):                     # all ranks at once from here on
        got = run()
        bad += int(not all(torch.equal(a, b) for a, b in zip(want, got)))
        if it % 5 == 4:                         # the exchange step, rank interleave included
            D, L, N = idist.all_gather_detections(got[0].cpu(), got[1].cpu(), got[3].cpu())
            bad += int(D.shape[0] != world * B or not torch.equal(D[rank::world], got[0].cpu())
                       or not torch.equal(N[rank::world], got[3].cpu()))
    torch.cuda.synchronize()
    g2, b2, ws = ops.state_workspace_for(geom, *dev)
    last_id, fallbacks = ops.get_bboxes_status(g2, b2, ws)
    print(json.dumps({'rank': rank, 'world': world, 'iterations': iters, 'mismatches': bad,
                      'fused_fallbacks': fallbacks}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
