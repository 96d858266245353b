"""SE-ResNet policy/value network with the reference's module and parameter
names (src/model.py:20-199), so that reference checkpoints load unchanged
(src/parallel_trainer.py:301-310), plus the hand-off to the gfx950 inference
kernels.

The torch modules are what the training step differentiates; eval-mode
inference on a GPU goes through libazamd (`HipInference`), which has no
fallback: if the library or the GPU is missing it raises.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .config import Config


class SEBlock(nn.Module):
    """Squeeze-and-excitation emitting a scale and a bias (src/model.py:20-53)."""

    def __init__(self, channels, se_ratio=8):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Linear(channels, channels // se_ratio)
        self.fc2 = nn.Linear(channels // se_ratio, channels * 2)
        self.channels = channels

    def forward(self, x):
        n = x.size(0)
        s = self.pool(x).view(n, -1)
        e = self.fc2(F.relu(self.fc1(s)))
        w, b = e.split(self.channels, dim=1)
        return x * torch.sigmoid(w).view(n, self.channels, 1, 1) + b.view(n, self.channels, 1, 1)


class SEResidualBlock(nn.Module):
    """conv-bn-relu-conv-bn-SE, skip, relu (src/model.py:56-92)."""

    def __init__(self, channels, se_ratio=8):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)
        self.se = SEBlock(channels, se_ratio)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.se(self.bn2(self.conv2(out)))
        return F.relu(out + x)


class AlphaZeroNet(nn.Module):
    """src/model.py:95-195.  forward(x[B,3,8,8]) -> (policy_logits[B,192], wl_logits[B,2])."""

    def __init__(self, num_blocks=Config.RESNET_BLOCKS, num_filters=Config.RESNET_FILTERS,
                 num_input_planes=Config.INPUT_PLANES, num_actions=Config.OUTPUT_ACTIONS,
                 se_ratio=Config.SE_RATIO):
        super().__init__()
        self.num_blocks = num_blocks
        self.num_filters = num_filters
        self.num_input_planes = num_input_planes
        self.num_actions = num_actions
        self.se_ratio = se_ratio
        self.input_conv = nn.Conv2d(num_input_planes, num_filters, kernel_size=3, padding=1, bias=False)
        self.input_bn = nn.BatchNorm2d(num_filters)
        self.residual_tower = nn.ModuleList(
            [SEResidualBlock(num_filters, se_ratio) for _ in range(num_blocks)])
        self.policy_conv = nn.Conv2d(num_filters, num_filters, kernel_size=3, padding=1, bias=False)
        self.policy_bn = nn.BatchNorm2d(num_filters)
        self.policy_fc = nn.Linear(num_filters * 8 * 8, num_actions)
        self.value_conv = nn.Conv2d(num_filters, 32, kernel_size=1, bias=False)
        self.value_bn = nn.BatchNorm2d(32)
        self.value_fc1 = nn.Linear(32 * 8 * 8, 128)
        self.value_fc2 = nn.Linear(128, 2)

    def forward(self, x):
        out = F.relu(self.input_bn(self.input_conv(x)))
        for block in self.residual_tower:
            out = block(out)
        policy = F.relu(self.policy_bn(self.policy_conv(out)))
        policy_logits = self.policy_fc(policy.reshape(policy.size(0), -1))
        value = F.relu(self.value_bn(self.value_conv(out)))
        value = F.relu(self.value_fc1(value.reshape(value.size(0), -1)))
        return policy_logits, self.value_fc2(value)

    def get_value(self, wl_logits):
        wl = F.softmax(wl_logits, dim=1)
        return (wl[:, 0] - wl[:, 1]).unsqueeze(1)

    def hip_supported(self):
        """Whether az_net_create takes this architecture: 3 input planes, 192 actions, a multiple of 32
        filters up to 256 (128 filters with SE ratio 8 run the tuned kernels, every other width
        k_tower_any), at least one hidden SE unit."""
        return (self.num_input_planes == 3 and self.num_actions == 192 and self.num_blocks >= 1
                and self.num_filters % 32 == 0 and 32 <= self.num_filters <= 256
                and 1 <= self.se_ratio <= self.num_filters)


LC0Network = AlphaZeroNet  # src/model.py:199


def pack_state(model):
    """Flatten state_dict() in its own order, skipping num_batches_tracked."""
    parts = [v.detach().reshape(-1).to(torch.float32).cpu() for k, v in model.state_dict().items()
             if not k.endswith("num_batches_tracked")]
    return torch.cat(parts).contiguous()


class HipInference:
    """Eval-mode forward of an AlphaZeroNet on the gfx950 kernels (az_net_* in
    include/azamd.h).  Call refresh() after the weights change."""

    def __init__(self, model, max_batch=4096, device=None):
        if not model.hip_supported():
            raise _lib.AzError("HIP inference supports 3 input planes, 192 actions and a multiple of 32 filters "
                               f"between 32 and 256 (got {model.num_input_planes} planes, {model.num_actions} actions, "
                               f"{model.num_filters} filters, SE ratio {model.se_ratio})")
        if not torch.cuda.is_available():
            raise _lib.AzError("HipInference needs a GPU (no CPU fallback exists)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.model = model
        self.max_batch = int(max_batch)
        cfg = _lib.NetConfig(model.num_blocks, model.num_filters, model.se_ratio, self.max_batch,
                             self.device.index)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().az_net_create(C.byref(h), C.byref(cfg)))
        self.h = h
        self.refresh()

    def refresh(self):
        blob = pack_state(self.model)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().az_net_load_state(self.h, C.c_void_p(blob.data_ptr()), blob.numel()))

    def set_tower_form(self, form=0):
        """0: by batch size; 1, 2, 4, 8, 16: force that form of the tower kernel."""
        _lib.check(_lib.lib().az_net_set_tower_form(self.h, int(form)))

    def set_timing(self, enable=True):
        _lib.check(_lib.lib().az_net_set_timing(self.h, int(bool(enable))))

    def tower_time(self):
        """-> (mean milliseconds per tower-kernel launch, samples) since the last call."""
        ms, n = C.c_double(), C.c_int64()
        _lib.check(_lib.lib().az_net_tower_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().az_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self, B, want_logits):
        d = self.device
        probs = torch.empty((B, 192), dtype=torch.float32, device=d)
        value = torch.empty((B,), dtype=torch.float32, device=d)
        pl = torch.empty((B, 192), dtype=torch.float32, device=d) if want_logits else None
        wl = torch.empty((B, 2), dtype=torch.float32, device=d) if want_logits else None
        return probs, value, pl, wl

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def forward_bits(self, mine, theirs, want_logits=False):
        """mine/theirs: int64 cuda tensors holding the canonical bitboards."""
        B = mine.numel()
        assert mine.dtype == torch.int64 and theirs.dtype == torch.int64 and theirs.numel() == B
        assert mine.device == self.device and theirs.device == self.device, "inputs live on another device"
        assert mine.is_contiguous() and theirs.is_contiguous()
        probs, value, pl, wl = self._out(B, want_logits)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.lib().az_net_forward(self.h, B, self._p(mine), self._p(theirs), self._p(probs),
                                                 self._p(value), self._p(pl), self._p(wl), stream))
        return probs, value, pl, wl

    def forward_bits_into(self, mine, theirs, probs, value, policy_logits=None, wl_logits=None):
        """az_net_forward_noalloc: like forward_bits for at most 256 positions, writing into
        tensors the caller already owns; neither this wrapper nor the library allocates."""
        B = mine.numel()
        assert mine.dtype == torch.int64 and theirs.dtype == torch.int64 and theirs.numel() == B
        assert mine.is_contiguous() and theirs.is_contiguous()
        for name, t, row in (("probs", probs, 192), ("value", value, 1), ("policy_logits", policy_logits, 192),
                             ("wl_logits", wl_logits, 2)):
            if t is None:
                assert name not in ("probs", "value"), f"{name} is required"
                continue
            assert t.dtype == torch.float32 and t.is_contiguous(), f"{name} must be a contiguous float32 tensor"
            assert t.numel() >= B * row, f"{name} holds {t.numel()} elements, {B} positions need {B * row}"
        for t in (mine, theirs, probs, value, policy_logits, wl_logits):
            assert t is None or t.device == self.device, "a tensor lives on another device"
        with torch.cuda.device(self.device):     # the library checks that the network's device is current
            _lib.check(_lib.lib().az_net_forward_noalloc(
                self.h, B, self._p(mine), self._p(theirs), self._p(probs), self._p(value), self._p(policy_logits),
                self._p(wl_logits), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def forward_planes(self, planes, want_logits=False):
        """planes: [B,3,8,8] float32 cuda tensor of encoded states."""
        assert planes.dtype == torch.float32 and planes.is_contiguous() and tuple(planes.shape[1:]) == (3, 8, 8)
        assert planes.device == self.device, "input lives on another device"
        B = planes.shape[0]
        probs, value, pl, wl = self._out(B, want_logits)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.lib().az_net_forward_planes(self.h, B, self._p(planes), self._p(probs),
                                                        self._p(value), self._p(pl), self._p(wl), stream))
        return probs, value, pl, wl
