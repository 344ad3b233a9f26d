"""``torch.distributed`` (gloo) helpers of the CPU tests of the sharding logic -- test scaffolding, not product code
(the product's collectives are ``aind_smartspim_destripe_amd.distributed.RankGroup``: RCCL through the C ABI, no torch)."""

import numpy as np


def broadcast_array(dist, array, src=0, device=None):
    """Broadcast a NumPy array from ``src`` (shape and dtype must already agree on all ranks).

    ``device=None`` uses a host tensor (gloo); ``device='cuda'`` stages through a device tensor so
    that the transfer is an RCCL broadcast over xGMI.
    """
    import torch

    a = np.ascontiguousarray(array)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    out = t.cpu().numpy().view(a.dtype).reshape(a.shape)
    return out


def broadcast_shading(dist, flatfield, darkfield, shape_flat, shape_dark, src=0, device=None):
    """Rank ``src`` holds the retrospective flat / dark planes of a tile; every rank gets a copy."""
    rank = dist.get_rank()
    flat = np.asarray(flatfield, dtype=np.float32) if rank == src else np.empty(shape_flat, np.float32)
    dark = np.asarray(darkfield, dtype=np.float32) if rank == src else np.empty(shape_dark, np.float32)
    return broadcast_array(dist, flat, src, device), broadcast_array(dist, dark, src, device)


def reduce_counters(dist, slices_done, seconds):
    """Sum of slices and max of elapsed time over ranks -> whole-job slices/s."""
    import torch

    t = torch.tensor([float(slices_done)], dtype=torch.float64)
    m = torch.tensor([float(seconds)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(t[0]), float(m[0])
