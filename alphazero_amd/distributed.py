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


def allreduce_mean_gradients(model: torch.nn.Module, group=None, bucket_bytes: int = 32 << 20):
    """Average .grad over ranks with flat buckets (the 20x128 network's 32 MB of
    float32 gradients go out as one bucket by default)."""
    world = dist.get_world_size(group)
    params = [p for p in model.parameters() if p.grad is not None]
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in bucket:
            k = p.grad.numel()
            p.grad.copy_(flat[off: off + k].view_as(p.grad))
            off += k
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.grad.numel() * p.grad.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def losses(model, states, target_pis, target_wls):
    """Soft-target policy cross-entropy + WL cross-entropy (src/parallel_trainer.py:251-260)."""
    pred_pis, pred_wls = model(states)
    log_probs = F.log_softmax(pred_pis, dim=1)
    loss_pi = -torch.sum(target_pis * log_probs) / target_pis.size(0)
    loss_wl = F.cross_entropy(pred_wls, target_wls)
    return loss_pi + loss_wl, loss_pi, loss_wl


def train_step(model, optimizer, states, target_pis, target_wls, group=None, autocast_dtype=None):
    """One optimisation step on this rank's slice of the batch; gradients are
    averaged over the ranks of `group` when torch.distributed is initialised.  A
    rank whose slice is empty (ragged last batch) contributes zero gradients and
    still joins the all-reduce.  autocast_dtype (e.g. torch.bfloat16) runs the
    forward under torch.autocast; None is the reference's float32 step."""
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
        allreduce_mean_gradients(model, group)
    if Config.GRAD_CLIP_NORM > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), Config.GRAD_CLIP_NORM)
    optimizer.step()
    return out
