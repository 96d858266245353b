"""The two exchange steps of sharded self-play (one process per GPU, games sharded
by rank, no collective during self-play):

  * all-gather of finished-game samples into every rank's replay buffer
    (the arrays the reference appends to training_data.npz,
    src/parallel_trainer.py:367-396);
  * gradient all-reduce for the training step (the reference's learn() inner
    step, src/parallel_trainer.py:245-270, made data parallel).

Both work on any torch.distributed backend: "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" on CPU tensors in the tests.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .config import Config


def shard_games(total_games: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the game indices rank owns; sizes differ by at most one."""
    base, rem = divmod(total_games, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_samples(states: torch.Tensor, policies: torch.Tensor, wls: torch.Tensor, group=None):
    """Every rank contributes n_r samples (n_r may differ, may be 0) and receives all
    of them, concatenated in rank order.  One count exchange, then one padded
    all_gather per array."""
    world = dist.get_world_size(group)
    n = torch.tensor([states.shape[0]], dtype=torch.int64, device=states.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    out = []
    for t in (states, policies, wls):
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        out.append(torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0))
    return tuple(out)


def all_gather_packed(packed: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of ragged row blocks: every rank contributes packed[n_r, width] (n_r may
    differ, may be 0) and receives all rows, in rank order.  One count exchange, then ONE
    all_gather_into_tensor of the padded blocks.  Used with the engine's packed sample rows
    (Engine.drain_packed: 792 bytes per sample instead of 1544); unpack the result on the
    receiving GPU with engine.unpack_samples."""
    world = dist.get_world_size(group)
    width = packed.shape[1]
    n = torch.tensor([packed.shape[0]], dtype=torch.int64, device=packed.device)
    counts = torch.empty(world, dtype=torch.int64, device=packed.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = [int(c) for c in counts.tolist()]
    cap = max(counts)
    if cap == 0:
        return packed[:0].clone()
    pad = packed
    if packed.shape[0] != cap:
        pad = torch.zeros((cap, width), dtype=packed.dtype, device=packed.device)
        pad[: packed.shape[0]] = packed
    out = torch.empty((world * cap, width), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(c == cap for c in counts):
        return out
    return torch.cat([out[r * cap: r * cap + c] for r, c in enumerate(counts)], dim=0)


def _flat_allreduce(tensors: List[torch.Tensor], group=None, bucket_bytes: int = 32 << 20):
    """In-place SUM all-reduce of a list of tensors through flat buckets (the 20x128
    network's 32 MB of float32 gradients go out as one bucket by default)."""
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for t in bucket:
            k = t.numel()
            t.copy_(flat[off: off + k].view_as(t))
            off += k
        bucket, size = [], 0

    for t in tensors:
        bucket.append(t)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def allreduce_gradients(model: torch.nn.Module, weight: float, group=None, bucket_bytes: int = 32 << 20):
    """Replace every .grad by the sum over ranks of weight_r * grad_r.  With
    weight_r = (examples in rank r's slice) / (examples in the whole batch) that sum is the
    gradient of the mean loss over the whole batch, however ragged the slices are (an equal
    1/world would give a slice of one example the weight of a full one)."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if weight != 1.0:
        for g in grads:
            g.mul_(weight)
    _flat_allreduce(grads, group, bucket_bytes)


def sync_batchnorm_buffers(model: torch.nn.Module, weight: float, group=None):
    """Keep the replicas' BatchNorm running statistics identical: each rank updated them from
    its own slice, so they are replaced by the slice-size-weighted mean over the ranks (a
    rank with an empty slice ran no forward and counts for nothing); num_batches_tracked
    becomes the maximum.  Self-play folds these statistics into the convolution weights, so
    without this every rank would play with a different network after one learn()."""
    floats = [b for b in model.buffers() if b.is_floating_point()]
    if floats:
        for b in floats:
            b.mul_(weight)
        _flat_allreduce(floats, group)
    counters = [b for b in model.buffers() if not b.is_floating_point()]
    if counters:                                   # one collective for all of them
        flat = torch.stack([b.reshape(()).to(torch.int64) for b in counters])
        dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group)
        for b, v in zip(counters, flat.unbind(0)):
            b.copy_(v)


def losses(model, states, target_pis, target_wls):
    """Soft-target policy cross-entropy + WL cross-entropy (src/parallel_trainer.py:251-260)."""
    pred_pis, pred_wls = model(states)
    log_probs = F.log_softmax(pred_pis, dim=1)
    loss_pi = -torch.sum(target_pis * log_probs) / target_pis.size(0)
    loss_wl = F.cross_entropy(pred_wls, target_wls)
    return loss_pi + loss_wl, loss_pi, loss_wl


def train_step(model, optimizer, states, target_pis, target_wls, group=None, autocast_dtype=None,
               global_batch=None):
    """One optimisation step on this rank's slice of the batch.  With torch.distributed
    initialised the ranks' gradients are combined with weights slice/global_batch (the
    gradient of the whole batch's mean loss; global_batch = None: the slice sizes are
    summed over the ranks first) and the BatchNorm running statistics are averaged the same
    way, so the replicas - parameters and buffers - stay identical.  A rank whose slice is
    empty (ragged last batch) contributes zero gradients and still joins the collectives.
    autocast_dtype (e.g. torch.bfloat16) runs the forward under torch.autocast; None is the
    reference's float32 step."""
    model.train()
    optimizer.zero_grad()
    if states.shape[0] > 0:
        if autocast_dtype is None:
            loss, loss_pi, loss_wl = losses(model, states, target_pis, target_wls)
        else:
            # MIOpen's default pick for the reduced-precision 8x8 convolutions is a slow one;
            # letting it time its solvers once per layer shape gives 20 % more steps per second
            searched = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            try:
                with torch.autocast(device_type=states.device.type, dtype=autocast_dtype):
                    pred_pis, pred_wls = model(states)
                log_probs = F.log_softmax(pred_pis.float(), dim=1)
                loss_pi = -torch.sum(target_pis * log_probs) / target_pis.size(0)
                loss_wl = F.cross_entropy(pred_wls.float(), target_wls)
                loss = loss_pi + loss_wl
                loss.backward()
            finally:
                torch.backends.cudnn.benchmark = searched
        if autocast_dtype is None:
            loss.backward()
        out = (float(loss.detach()), float(loss_pi.detach()), float(loss_wl.detach()))
    else:
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        out = (0.0, 0.0, 0.0)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        mine = states.shape[0]
        if global_batch is None:
            total = torch.tensor([mine], dtype=torch.int64, device=states.device)
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
            global_batch = int(total.item())
        weight = mine / max(1, global_batch)
        allreduce_gradients(model, weight, group)
        with torch.no_grad():
            sync_batchnorm_buffers(model, weight, group)
    if Config.GRAD_CLIP_NORM > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), Config.GRAD_CLIP_NORM)
    optimizer.step()
    return out
