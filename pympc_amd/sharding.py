"""Sharding of a batch of independent MPC instances over the GPUs of one node.

The instances are independent (no coupling, no reduction -- SURVEY.md section 8e): rank r owns the contiguous block
[r*B, (r+1)*B).  Collectives (RCCL on GPUs, gloo in the CPU tests) are used only where the path has a real exchange:
  * ``scatter_instances``: the problem data (Ad, Bd, x0, ...) is generated on rank 0 and scattered once, at setup;
  * ``gather_inputs``:     the first optimal inputs u* of every shard are all-gathered after each solve;
  * ``gather_trajectory``: the applied inputs of a whole device-loop launch are all-gathered at once.
One process per GPU, launched by torch.distributed.run; without a process group everything degenerates to plain copies.  (With a process
group of ONE rank the collectives still run -- a communicator of one: how the RCCL path is exercised on a single-GPU box.)
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _no_group():
    return not (dist.is_available() and dist.is_initialized())


def shard_range(total, rank, world_size):
    """Contiguous block of instance indices owned by ``rank`` (``total`` must divide evenly)."""
    if total % world_size:
        raise ValueError('the number of instances must be a multiple of the number of ranks')
    per = total // world_size
    return rank * per, (rank + 1) * per


def scatter_instances(full, shapes, per_rank, device, dtype=torch.float64, src=0):
    """``full``: dict name -> tensor [world*per_rank, ...] on rank ``src`` (None elsewhere);
    ``shapes``: dict name -> per-instance shape.  Returns dict name -> local shard [per_rank, ...]."""
    rank, ws = world()
    out = {}
    for name, shp in shapes.items():
        loc = torch.empty((per_rank,) + tuple(shp), dtype=dtype, device=device)
        if _no_group():
            loc.copy_(full[name])
        else:
            chunks = list(full[name].to(device=device, dtype=dtype).contiguous().split(per_rank)) if rank == src else None
            dist.scatter(loc, chunks, src=src)
        out[name] = loc
    return out


def gather_inputs(u_local, out=None):
    """All-gather the [per_rank, nu] first inputs of every rank into [world*per_rank, nu] (instance order)."""
    rank, ws = world()
    if _no_group():
        return u_local
    if out is None:
        out = torch.empty((ws * u_local.shape[0],) + tuple(u_local.shape[1:]), dtype=u_local.dtype, device=u_local.device)
    dist.all_gather_into_tensor(out, u_local.contiguous())
    return out


def gather_trajectory(u_traj, out=None):
    """All-gather a [steps, per_rank, nu] input trajectory of every rank into [world, steps, per_rank, nu]
    (device loop: one exchange per launch instead of one per step).  ``out``, if given, is [world*steps, per_rank, nu]
    (the concatenated form every backend accepts); the returned tensor is a view of it."""
    rank, ws = world()
    if _no_group():
        return u_traj.unsqueeze(0)
    steps = u_traj.shape[0]
    if out is None:
        out = torch.empty((ws * steps,) + tuple(u_traj.shape[1:]), dtype=u_traj.dtype, device=u_traj.device)
    dist.all_gather_into_tensor(out, u_traj.contiguous())
    return out.view((ws, steps) + tuple(u_traj.shape[1:]))
