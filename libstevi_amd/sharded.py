"""Disparity-sharded census (+ SGM) over the GPUs of one node: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI) for the single exchange of the pipeline.

    rank r:  keys_r = svh_census_shard_keys(shard r of the disparity range)      (H, W, 2) int32, 8 B / pixel
             all_reduce(keys, MIN)                                              RCCL
             disp   = svh_census_shard_finish(keys)                             replicated, bit-identical to 1 GPU

Why one int32 MIN all-reduce is enough: in the Cost branch as the reference computes it a pass couples
the disparities of a pixel only through min_d [c + (c [+Pout])]; with integer census costs that minimum and the winner
are both functions of the two regional minima (cost, last index) the keys carry.
"""
import torch.distributed as dist

from . import correlation as _c


def shard_range(total, rank, world):
    """Contiguous split of `total` disparities; the first total % world ranks get one more."""
    base, rem = divmod(int(total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def stereoMatchSharded(img_l, img_r, h_radius, v_radius, disp_width, group=None, **kw):
    """Census + SGM with the disparity range split over the ranks of `group`.  Every rank passes the same images
    (resident on its own GPU) and gets the same disparity map back.  kw: dDir, sgmDirections, P1, P2, Pout, margins,
    refineKernel, refine_h_radius, refine_v_radius."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    _, D = _c._search_range(disp_width)
    shard = shard_range(D, rank, world)
    if shard[1] == 0:
        raise ValueError("more ranks than disparities")
    keys_kw = {k: v for k, v in kw.items() if k in ("dDir", "sgmDirections", "P1", "P2", "Pout", "margins", "matchFunc")}
    keys = _c.censusShardKeys(img_l, img_r, h_radius, v_radius, disp_width, shard, **keys_kw)
    if world > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
    return _c.censusShardFinish(img_l, img_r, keys, h_radius, v_radius, disp_width, **kw)
