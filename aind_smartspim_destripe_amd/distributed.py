"""Multi-GPU plumbing of the destripe path: one process per GPU, z-slices shard embarrassingly.

The reference parallelises over Zarr chunks with OS processes and a queue
(``zarr_destriper.py:1138-1172``); every plane is filtered independently
(``zarr_destriper.py:319-327``, overlap ``(0, 0, 0)`` at ``:1018-1022``).  Here rank == GPU, each
rank owns a contiguous z-range aligned to the Zarr z-chunk, and the only collective is one
broadcast (root 0) of the constant blob -- filter tables and, when shading is on, the flat / dark
planes -- before the data path starts.  Nothing on the data path communicates.

``torch.distributed`` is the transport (backend ``nccl`` == RCCL over xGMI for device tensors,
``gloo`` for host tensors / CPU tests); it is passed in, this module does not import torch itself.
"""

import numpy as np


def z_shard(n_slices, world_size, rank, z_chunk=64):
    """Contiguous ``[start, stop)`` z-range of ``rank``: whole z-chunks, spread as evenly as possible.

    Chunk alignment keeps every output Zarr chunk ``(1, 1, 64, 128, 128)``
    (``zarr_destriper.py:1066-1074``) written by exactly one rank, so no locking is needed -- the same
    argument that makes the reference's concurrent consumers safe.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    n_chunks = (n_slices + z_chunk - 1) // z_chunk
    base, extra = divmod(n_chunks, world_size)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    start = min(first * z_chunk, n_slices)
    stop = min((first + count) * z_chunk, n_slices)
    return start, stop


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
