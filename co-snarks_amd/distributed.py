"""Split-MSM exchange step: one MSM whose points are partitioned by contiguous ranges over the
ranks of one node. Every rank reduces its range to per-window partial sums on its GPU
(`csh_msm_partial_dev`), the partial buffers (a few KiB each) are exchanged with ONE all-gather
(`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), and every rank folds the
gathered partials on the host (`csh_msm_fold_partials`: window-wise Horner + point additions + one inversion).
RCCL has no elliptic-curve reduction op, so an all-reduce cannot be used; the payload is latency-bound."""
from __future__ import annotations

import numpy as np

from . import bindings as B


def allgather_and_fold(partial, curve: int, group: int, world: int, dist=None):
    """partial: uint8 torch tensor of msm_partial_bytes(curve, group) bytes (device or CPU).
    Returns the Jacobian (X, Y, Z) limbs of the full MSM on every rank."""
    import torch
    if world == 1:
        host = partial.cpu().numpy()
    else:
        gathered = torch.empty(world * partial.numel(), dtype=torch.uint8, device=partial.device)
        dist.all_gather_into_tensor(gathered, partial)
        host = gathered.cpu().numpy()
    return B.msm_fold_partials(curve, group, np.ascontiguousarray(host), world)
