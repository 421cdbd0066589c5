"""Sharding of a batch of independent MPC instances over the GPUs of one node.

The instances are independent (no coupling, no reduction -- SURVEY.md section 8e): rank r owns the contiguous block
[r*per, min((r+1)*per, total)), per = ceil(total / world) -- the last rank(s) may be short, or empty, when the total does not divide.
Collectives (RCCL on GPUs, gloo in the CPU tests) are used only where the path has a real exchange:
  * ``scatter_instances``: the problem data (Ad, Bd, x0, ...) is generated on rank 0, PACKED into one [world*per, width] buffer and
                           scattered once, at setup -- ONE collective whatever the number of arrays (17 per-array scatters before);
  * ``broadcast_model``:   ONE model for every instance of every rank (last paragraph: instances that share (Ad, Bd, Q*, bounds) and differ only in
                           x0 -- the reference's one-controller-many-states caller, test_scripts/example_mpc_function.py:105-111): the model is packed and
                           broadcast once, ``scatter_instances`` then carries the states alone; set up with the model and ONE common state on every
                           rank, the shard's instances share one KKT factor (mpcqp_share_factor), and ``update`` scatters the states;
  * ``gather_inputs``:     the first optimal inputs u* of every shard are all-gathered after each solve;
  * ``gather_trajectory``: the applied inputs of a whole device-loop launch are all-gathered at once.
Short shards are padded to ``per`` rows for the collectives (all_gather_into_tensor wants equal pieces) and trimmed afterwards.
One process per GPU, launched by torch.distributed.run; without a process group everything degenerates to plain copies.  (With a process
group of ONE rank the collectives still run -- a communicator of one: how the RCCL path is exercised on a single-GPU box.)
"""
import math

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _no_group():
    return not (dist.is_available() and dist.is_initialized())


def shard_rows(total, world_size):
    """Rows of the padded per-rank block: ceil(total / world_size)."""
    return -(-int(total) // int(world_size))


def shard_range(total, rank, world_size):
    """Contiguous block [lo, hi) of instance indices owned by ``rank``: ``shard_rows`` each, the last rank(s) short (or empty)."""
    per = shard_rows(total, world_size)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def scatter_instances(full, shapes, per_rank=None, device=None, dtype=torch.float64, src=0, total=None):
    """``full``: dict name -> tensor [total, ...] on rank ``src`` (None elsewhere); ``shapes``: dict name -> per-instance shape.
    ``total`` instances in all (default: world * per_rank, the even case).  Returns dict name -> local shard [count, ...] with
    count = this rank's ``shard_range``.  One ``dist.scatter`` of a packed [world * per, width] buffer."""
    rank, ws = world()
    if total is None:
        total = ws * int(per_rank)
    per = shard_rows(total, ws)
    lo, hi = shard_range(total, rank, ws)
    names = list(shapes)
    widths = [int(math.prod(shapes[k])) if len(tuple(shapes[k])) else 1 for k in names]
    width = sum(widths)
    loc = torch.empty((per, width), dtype=dtype, device=device)
    if _no_group():
        loc[:total] = torch.cat([full[k].to(device=device, dtype=dtype).reshape(total, -1) for k in names], dim=1)
    else:
        chunks = None
        if rank == src:
            packed = torch.zeros((ws * per, width), dtype=dtype, device=device)
            packed[:total] = torch.cat([full[k].to(device=device, dtype=dtype).reshape(total, -1) for k in names], dim=1)
            chunks = list(packed.split(per))
        dist.scatter(loc, chunks, src=src)
    out, off = {}, 0
    for k, w in zip(names, widths):
        out[k] = loc[:hi - lo, off:off + w].reshape((hi - lo,) + tuple(shapes[k])).contiguous()
        off += w
    return out


def broadcast_model(model, shapes, device=None, dtype=torch.float64, src=0):
    """``model``: dict name -> tensor of shape ``shapes[name]`` on rank ``src`` (None elsewhere).  Returns the same dict on every rank: ONE
    ``dist.broadcast`` of a packed buffer whatever the number of arrays."""
    rank, _ = world()
    names = list(shapes)
    widths = [int(math.prod(shapes[k])) if len(tuple(shapes[k])) else 1 for k in names]
    buf = torch.empty((sum(widths),), dtype=dtype, device=device)
    if _no_group() or rank == src:
        buf.copy_(torch.cat([torch.as_tensor(model[k]).to(device=device, dtype=dtype).reshape(-1) for k in names]))
    if not _no_group():
        dist.broadcast(buf, src=src)
    out, off = {}, 0
    for k, w in zip(names, widths):
        out[k] = buf[off:off + w].reshape(tuple(shapes[k])).clone()
        off += w
    return out


def _padded(t, rows, dim):
    """``t`` with dimension ``dim`` zero-padded to ``rows``."""
    if t.shape[dim] == rows:
        return t.contiguous()
    shp = list(t.shape); shp[dim] = rows - t.shape[dim]
    return torch.cat([t, torch.zeros(shp, dtype=t.dtype, device=t.device)], dim=dim).contiguous()


def gather_inputs(u_local, out=None, total=None):
    """All-gather the [count, nu] first inputs of every rank into [total, nu] (instance order).  ``total``: instances in all (default:
    world * count, the even case); ``out``, if given, is the [world * per, nu] exchange buffer (the result is a view of its first rows)."""
    rank, ws = world()
    if _no_group():
        return u_local
    if total is None:
        total = ws * u_local.shape[0]
    per = shard_rows(total, ws)
    if out is None:
        out = torch.empty((ws * per,) + tuple(u_local.shape[1:]), dtype=u_local.dtype, device=u_local.device)
    dist.all_gather_into_tensor(out, _padded(u_local, per, 0))
    return out[:total]


def gather_trajectory(u_traj, out=None, total=None):
    """All-gather a [steps, count, nu] input trajectory of every rank (device loop: one exchange per launch instead of one per step).
    Even shards (``total`` None): returns [world, steps, count, nu], a view of ``out`` ([world*steps, count, nu]) if that is given.
    With ``total``: shards may be short; returns [steps, total, nu] in instance order."""
    rank, ws = world()
    if _no_group():
        return u_traj.unsqueeze(0) if total is None else u_traj
    steps = u_traj.shape[0]
    per = u_traj.shape[1] if total is None else shard_rows(total, ws)
    if out is None:
        out = torch.empty((ws * steps, per) + tuple(u_traj.shape[2:]), dtype=u_traj.dtype, device=u_traj.device)
    dist.all_gather_into_tensor(out, _padded(u_traj, per, 1))
    v = out.view((ws, steps, per) + tuple(u_traj.shape[2:]))
    if total is None:
        return v
    return v.transpose(0, 1).reshape((steps, ws * per) + tuple(u_traj.shape[2:]))[:, :total]
