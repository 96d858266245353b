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
               global_batch=None, as_tensor=False):
    """One optimisation step on this rank's slice of the batch.  With torch.distributed
    initialised the ranks' gradients are combined with weights slice/global_batch (the
    gradient of the whole batch's mean loss; global_batch = None: the slice sizes are
    summed over the ranks first) and the BatchNorm running statistics are averaged the same
    way, so the replicas - parameters and buffers - stay identical.  A rank whose slice is
    empty (ragged last batch) contributes zero gradients and still joins the collectives.
    autocast_dtype (e.g. torch.bfloat16) runs the forward under torch.autocast; None is the
    reference's float32 step.  Returns (loss, loss_pi, loss_wl) as floats, or with
    as_tensor=True as one float32 tensor of three on the model's device, which costs no
    host synchronisation (a caller that only wants epoch means adds them up on the device)."""
    model.train()
    # gradients are zeroed in place, not dropped: a GraphedTrainStep of the same trainer has captured
    # these very tensors, and an eager step (a ragged last batch) must not replace them
    optimizer.zero_grad(set_to_none=False)
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
        out = torch.stack((loss.detach().float(), loss_pi.detach().float(), loss_wl.detach().float()))
    else:
        for p in model.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        out = torch.zeros(3, dtype=torch.float32, device=states.device)
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
    return out if as_tensor else tuple(out.tolist())


def make_capturable(optimizer):
    """Adam that can run inside a captured HIP graph: step counters on the parameters' device."""
    for group in optimizer.param_groups:
        group["capturable"] = True
        for p in group["params"]:
            st = optimizer.state.get(p)
            if st and torch.is_tensor(st.get("step")) and st["step"].device != p.device:
                st["step"] = st["step"].to(p.device)


class GraphedTrainStep:
    """train_step for one fixed slice size, replayed from two captured HIP graphs: (A) zero the
    gradients, forward, losses, backward; (B) gradient clipping and the Adam update.  Between
    them, with torch.distributed initialised, the eager gradient all-reduce and BatchNorm
    buffer sync of train_step.  At the reference's batch of 1024 the step is bound by its
    kernels and a graph buys nothing; at a data-parallel rank's slice of it (128 of 1024 on
    8 GPUs) the ~2,500 small launches of a step are what it costs, and the replay takes
    11.1 instead of 14.7 ms in float32 and 9.8 instead of 24.0 ms under bf16 autocast
    (tools/train_graph_probe.py --batch 128).

    The learning rate and weight decay are constants of graph B: it is captured again when
    the optimiser's learning rate has changed (the trainer's schedule steps once per learn()).
    Arithmetic: PyTorch's capturable Adam, which computes the bias corrections on the device;
    weights agree with the eager step's to float32 rounding, not bit for bit.

    The losses of every replay are added into `self.sums` (float32[3] on the device: loss,
    policy loss, WL loss) without a host synchronisation; zero it per epoch."""

    WARMUP = 3

    def __init__(self, model, optimizer, slice_size, autocast_dtype=None, group=None):
        p0 = next(model.parameters())
        if p0.device.type != "cuda":
            raise ValueError("GraphedTrainStep needs the model on a GPU")
        self.model, self.optimizer, self.group = model, optimizer, group
        self.n = int(slice_size)
        self.autocast_dtype = autocast_dtype
        dev = p0.device
        self.states = torch.zeros((self.n, 3, 8, 8), dtype=torch.float32, device=dev)
        self.pis = torch.zeros((self.n, 192), dtype=torch.float32, device=dev)
        self.wls = torch.zeros((self.n, 2), dtype=torch.float32, device=dev)
        self.sums = torch.zeros(3, dtype=torch.float32, device=dev)
        self.graph_a = self.graph_b = None
        self.lr = None
        self.steps_seen = 0

    def _grads(self):
        self.optimizer.zero_grad(set_to_none=False)
        if self.autocast_dtype is None:
            loss, loss_pi, loss_wl = losses(self.model, self.states, self.pis, self.wls)
        else:
            with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                pred_pis, pred_wls = self.model(self.states)
            log_probs = F.log_softmax(pred_pis.float(), dim=1)
            loss_pi = -torch.sum(self.pis * log_probs) / self.n
            loss_wl = F.cross_entropy(pred_wls.float(), self.wls)
            loss = loss_pi + loss_wl
        loss.backward()
        self.sums += torch.stack((loss.detach().float(), loss_pi.detach().float(), loss_wl.detach().float()))

    def _update(self):
        if Config.GRAD_CLIP_NORM > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), Config.GRAD_CLIP_NORM)
        self.optimizer.step()

    def _between(self, global_batch):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            weight = self.n / max(1, global_batch)
            allreduce_gradients(self.model, weight, self.group)
            with torch.no_grad():
                sync_batchnorm_buffers(self.model, weight, self.group)

    def __call__(self, states, target_pis, target_wls, global_batch=None):
        """One optimisation step on a slice of exactly `slice_size` examples."""
        assert states.shape[0] == self.n
        self.model.train()
        self.states.copy_(states); self.pis.copy_(target_pis); self.wls.copy_(target_wls)
        global_batch = self.n if global_batch is None else global_batch
        searched = torch.backends.cudnn.benchmark
        if self.autocast_dtype is not None:
            torch.backends.cudnn.benchmark = True          # see train_step
        try:
            if self.steps_seen < self.WARMUP:
                # the first steps run eagerly on a side stream: optimiser state, MIOpen's solver
                # choices and autograd's buffers exist before anything is captured
                if self.steps_seen == 0:
                    make_capturable(self.optimizer)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._grads()
                    self._between(global_batch)
                    self._update()
                torch.cuda.current_stream().wait_stream(side)
                self.steps_seen += 1
                return
            if self.graph_a is None:
                self.graph_a = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_a):
                    self._grads()
            lr = float(self.optimizer.param_groups[0]["lr"])
            if self.graph_b is None or lr != self.lr:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b):
                    self._update()
                self.lr = lr
            self.graph_a.replay()
            self._between(global_batch)
            self.graph_b.replay()
            self.steps_seen += 1
        finally:
            torch.backends.cudnn.benchmark = searched
