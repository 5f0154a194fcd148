"""
Host-side mirror of the reference backbone interface (sugar/models/dynamictdnn.py:587-948).

  DynamicTDNN   -- the supernet as a *parameter container* with the reference's state-dict keys, its
                   ``config`` property / setter (:681-711), ``clone(config)`` (:713-730), ``space`` (:792-842),
                   ``count_ops`` / ``count_params`` (:735-778) and a forward that runs the active subnet.
  TDNNSubnet    -- what ``clone`` returns: static weights under the cloned key names (``layer1.conv.weight`` ...),
                   ``load_state_dict(bn, strict=False)`` for the ``*.bn.tar`` files, and a forward that runs the
                   hand-written HIP kernels through the C-ABI (no torch compute on the path).
  TDNNPath      -- the subnet sampler (:845-879), same RNG stream (random.Random(666)).
  factories     -- specfic_tdnn / ecapa_tdnn / tdnn8m2g / tdnn14m4g / tdnn16m4g / tdnn26m7g (:881-948).

Weight materialisation (channel slices, index gathers, k53 / k31 kernel transforms) follows
sugar/modules/dynamicarch.py:123-142, 265-291, 411-432.
"""
import ctypes as C
import math
import random

import torch
import torch.nn as nn

from . import _lib, layout
from .layout import BOTTLENECK, SCALE, SE_REDUCTION

__all__ = ["DynamicTDNN", "TDNNSubnet", "TDNNPath", "specfic_tdnn", "ecapa_tdnn", "tdnn8m2g", "tdnn14m4g",
           "tdnn16m4g", "tdnn26m7g"]

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


def _make_divisible(v, divisor=8, min_value=None):
    """sugar/models/dynamictdnn.py:32-49."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class _Node(nn.Module):
    """Anonymous container: lets a flat {dotted key: shape} layout become an nn.Module tree with those keys."""


def _build_tree(root: nn.Module, shapes, init):
    for key, shape in shapes.items():
        parts = key.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        leaf = parts[-1]
        t = init(key, tuple(shape))
        if leaf in _BUFFER_LEAVES:
            node.register_buffer(leaf, t)
        else:
            node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))


def _flat_tensors(root: nn.Module):
    """{dotted key: parameter / buffer} of a _build_tree module, read straight from the module dicts: what
    state_dict() returns for these hook-free trees at a tenth of its cost (311 tensors: 3 ms per call, paid per sampled
    subnet by materialize() and again by handle())."""
    out, stack = {}, [("", root)]
    while stack:
        prefix, m = stack.pop()
        for k, t in m._parameters.items():
            if t is not None:
                out[prefix + k] = t
        for k, t in m._buffers.items():
            if t is not None:
                out[prefix + k] = t
        for k, c in m._modules.items():
            if c is not None:
                stack.append((prefix + k + ".", c))
    return out


def _default_init(key, shape):
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("k53", "k31", "k75"):
        return torch.eye(shape[0])                                  # dynamicarch.py:385-392
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_var" or (leaf == "weight" and len(shape) == 1):
        return torch.ones(shape)
    if leaf in ("running_mean", "bias"):
        return torch.zeros(shape)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    bound = 1.0 / math.sqrt(fan_in)                                 # nn.Conv1d / nn.Linear default scale
    return torch.empty(shape).uniform_(-bound, bound)


def _sel(t, sel):
    return t[:sel] if isinstance(sel, int) else t.index_select(0, sel)


def _active_conv_weight(w, in_sel, out_sel, k_active, k53, k31):
    """DynamicConv1d.weight (dynamicarch.py:411-432)."""
    w = w[:, :in_sel] if isinstance(in_sel, int) else w.index_select(1, in_sel)
    w = w[:out_sel] if isinstance(out_sel, int) else w.index_select(0, out_sel)
    k_full = w.shape[2]
    if k_full > 5:
        raise ValueError("kernel size %d > 5 is not on the EfficientTDNN path" % k_full)
    if k_full > 3 and k_active < 5:
        w = torch.matmul(w[:, :, 1:-1], k53)
    if k_full > 1 and k_active < 3:
        w = torch.matmul(w[:, :, 1:-1], k31)
    return w.contiguous()


def _check_config(config, num_blocks, channels, kernel_sizes, catconv):
    depth, ch, ks, ccat = config
    if not (1 <= depth <= num_blocks):
        raise ValueError("depth %r not in 1..%d" % (depth, num_blocks))
    if len(ch) < depth + 1 or len(ks) < depth + 1:
        raise ValueError("config needs depth+1 channel and kernel entries: %r" % (config,))
    for i in range(depth + 1):
        if not (0 < ch[i] <= channels[i]):
            raise ValueError("channels[%d]=%r exceeds the supernet's %d" % (i, ch[i], channels[i]))
        if ks[i] not in (1, 3, 5) or ks[i] > kernel_sizes[i]:
            raise ValueError("kernel[%d]=%r not available (supernet kernel %d)" % (i, ks[i], kernel_sizes[i]))
        if i > 0 and ch[i] // SCALE < 1:
            raise ValueError("channels[%d]=%r gives an empty Res2 width" % (i, ch[i]))
    if not (0 < ccat <= catconv):
        raise ValueError("catconv channels %r exceed the supernet's %d" % (ccat, catconv))
    if ch[0] // SE_REDUCTION < 1:
        raise ValueError("channels[0]=%r too small for the SE reduction" % ch[0])
    return (int(depth), [int(c) for c in ch[:depth + 1]], [int(k) for k in ks[:depth + 1]], int(ccat))


class TDNNSubnet(nn.Module):
    """
    Static subnet, the return value of DynamicTDNN.clone (dynamictdnn.py:713-730).  Parameters live on the host
    under the cloned key names; the device copy is the library's packed form (sugar_subnet_create), built lazily
    per (device, precision) and dropped whenever the parameters change (load_state_dict).
    """

    def __init__(self, config, state=None, in_feats=80, out_embeds=192, precision="bf16", adopt=False):
        """state: {key: tensor} in the cloned-subnet layout (copied; `adopt=True`: taken over as they are -- the caller
        made them for this subnet and keeps no other reference, as DynamicTDNN.materialize does)."""
        super().__init__()
        depth, ch, ks, ccat = config
        self.subnet_config = (int(depth), list(ch[:depth + 1]), list(ks[:depth + 1]), int(ccat))
        self.in_feats, self.out_embeds = in_feats, out_embeds
        self.precision = precision
        shapes = layout.subnet_shapes(self.subnet_config, in_feats, out_embeds)
        _build_tree(self, shapes, _default_init if state is None else
                    ((lambda k, s: state[k].detach()) if adopt else (lambda k, s: state[k].detach().clone())))
        self._handles = {}       # (device index, dtype) -> c_void_p
        self._workspace = {}     # (device index, stream) -> torch.uint8 tensor
        self._device = torch.device("cpu")
        self._bn_dirty = None    # (device index, dtype) of the handle holding recalibrated, not yet synced statistics
        self._bn_unread = None   # ... folded into the eval path already, but not yet copied into this module's buffers
        self._bn_batches = 0
        self._generation = 0     # bumped whenever device pointers handed out earlier become invalid (release, precision)
        # parameters changed (also via a parent's load_state_dict) -> drop the packed device copies
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.release())
        self.training = False

    def train(self, mode=True):
        """The children are inert containers of tensors (no forward of their own): the mode lives on this module only
        (nn.Module.train walks the ~100 containers, 1.3 ms per call, several calls per sampled subnet in a sweep)."""
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        self.training = mode
        return self

    # ---- nn.Module plumbing: parameters stay on the host; .to()/.cuda() only select the device ---------------
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0))
        if probe.device != torch.device("cpu"):
            self._device = probe.device
        return self

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if isinstance(dev, (str, torch.device, int)):
            dev = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
            if dev.type == "cuda":
                self._device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
            else:
                self._device = torch.device("cpu")
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def cpu(self):
        self._device = torch.device("cpu")
        return self

    @property
    def device(self):
        return self._device

    def set_precision(self, precision):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision %r not in ['bf16', 'fp32']" % (precision,))
        if precision != self.precision:
            self._generation += 1
        self.precision = precision
        return self

    @property
    def generation(self):
        """Changes whenever packed weights / workspaces handed out earlier may have been freed or the compute dtype
        changed: what a captured HIP graph (graphs.GraphedEmbedder) compares before it replays baked-in pointers."""
        return self._generation

    def release(self):
        self._generation += 1
        for h in self._handles.values():
            _lib.lib().sugar_subnet_destroy(h)
        self._handles = {}
        self._workspace = {}
        self._bn_dirty, self._bn_unread, self._bn_batches = None, None, 0

    def drop_workspaces(self):
        """Give the per-(device, stream) workspaces back to torch's allocator (no device synchronisation; the packed
        weights stay).  The caller knows this subnet's kernels have run."""
        self._generation += 1
        self._workspace = {}

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def dtype_id(self):
        return _lib.SUGAR_BF16 if self.precision == "bf16" else _lib.SUGAR_F32

    # ---- packing -----------------------------------------------------------------------------------------
    def handle(self, device=None):
        device = self._device if device is None else torch.device(device)
        if device.type != "cuda":
            raise _lib.SugarError("sugar_amd runs on an MI355X only: move the model with .cuda() / .to('cuda:0') "
                                  "(there is no CPU fallback)")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, self.dtype_id)
        if key not in self._handles:
            self.sync_bn()                             # (as state_dict() does: recalibrated statistics into the buffers)
            sd = {k: v.detach().to(torch.float32).contiguous() for k, v in _flat_tensors(self).items()
                  if v.dtype.is_floating_point}
            depth, ch, ks, ccat = self.subnet_config

            def bn(prefix):
                return _lib.SugarBN(*[_lib.fptr(sd["%s.%s" % (prefix, leaf)])
                                      for leaf in ("weight", "bias", "running_mean", "running_var")])

            d = _lib.SugarSubnetDesc()
            d.in_feats, d.embed_dim, d.depth = self.in_feats, self.out_embeds, depth
            d.c0, d.k0, d.ccat, d.bottleneck = ch[0], ks[0], ccat, BOTTLENECK
            d.layer1_w, d.layer1_bn = _lib.fptr(sd["layer1.conv.weight"]), bn("layer1.bn")
            for i in range(depth):
                p = "blocks.%d." % i
                b = d.blocks[i]
                b.mid_channels = (ch[i + 1] // SCALE) * SCALE
                b.kernel = ks[i + 1]
                b.first_w, b.first_bn = _lib.fptr(sd[p + "first_conv.conv.weight"]), bn(p + "first_conv.bn")
                for j in range(SCALE - 1):
                    b.res_w[j] = _lib.fptr(sd[p + "resconv.convs.%d.conv.weight" % j])
                    b.res_bn[j] = bn(p + "resconv.convs.%d.bn" % j)
                b.last_w, b.last_bn = _lib.fptr(sd[p + "last_conv.conv.weight"]), bn(p + "last_conv.bn")
                b.se_reduce_w, b.se_reduce_b = _lib.fptr(sd[p + "se.reduce.weight"]), _lib.fptr(sd[p + "se.reduce.bias"])
                b.se_expand_w, b.se_expand_b = _lib.fptr(sd[p + "se.expand.weight"]), _lib.fptr(sd[p + "se.expand.bias"])
            d.catconv_w = _lib.fptr(sd["catconv.weight"])
            d.pool1_w, d.pool2_w = _lib.fptr(sd["pooling.linear1.weight"]), _lib.fptr(sd["pooling.linear2.weight"])
            d.bnpool, d.bnembed = bn("bnpool"), bn("bnembed")
            d.linear_w, d.linear_b = _lib.fptr(sd["linear.weight"]), _lib.fptr(sd["linear.bias"])
            h = C.c_void_p()
            _lib.check(_lib.lib().sugar_subnet_create(C.byref(d), self.dtype_id, idx, C.byref(h)), "sugar_subnet_create")
            self._handles[key] = h
        return self._handles[key]

    def workspace(self, device, B, R):
        need = _lib.check(_lib.lib().sugar_workspace_bytes(self.handle(device), B, R), "sugar_workspace_bytes")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        # one workspace per (device, stream): forwards issued on different streams may overlap on the GPU.  Callers
        # that rotate over streams reuse a fixed set (vectors.stream_lanes), so the table stays small; should a caller
        # keep creating streams anyway, the oldest entries go first.
        key = (idx, torch.cuda.current_stream(device).cuda_stream)
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < need:
            while len(self._workspace) >= 8:
                self._workspace.pop(next(iter(self._workspace)))
            self._workspace[key] = ws = torch.empty(int(need * 1.05) + 256, dtype=torch.uint8, device=device)
        return ws

    # ---- internal channel layout (for parity taps) ----------------------------------------------------------
    def packed_dims(self):
        depth, ch, ks, ccat = self.subnet_config
        up8 = lambda v: (v + 7) // 8 * 8
        return {"c0p": up8(ch[0]), "ccatp": up8(ccat),
                "w": [ch[i + 1] // SCALE for i in range(depth)], "wp": [up8(ch[i + 1] // SCALE) for i in range(depth)]}

    # ---- forward --------------------------------------------------------------------------------------------
    def forward(self, x):
        """x: [B, 80, T] features (the reference's backbone input, dynamictdnn.py:645-649) on the GPU -> [B, 192]."""
        if x.dim() != 3 or x.shape[1] != self.in_feats:
            raise ValueError("expected [B, %d, T] features, got %s" % (self.in_feats, tuple(x.shape)))
        if not x.is_cuda:
            raise _lib.SugarError("sugar_amd runs on an MI355X only: features must be CUDA tensors (no CPU fallback)")
        B, _, T = x.shape
        dt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        feats = x.transpose(1, 2).to(dt).contiguous().view(B * T, self.in_feats)   # channels-last packed rows
        row_off = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=x.device)
        return self.forward_rows(feats, row_off, B, B * T, T)

    def forward_rows(self, feats, row_off, B, R, t_max, normalize=False):
        """feats: [R, 80] channels-last rows in the subnet's dtype; row_off: int32 [B+1] on the device.
        In train() mode this is one BN-recalibration step (batch statistics, running statistics updated)."""
        dev = feats.device
        ws = self.workspace(dev, B, R)
        emb = torch.empty(B, self.out_embeds, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.training:
            rc = _lib.lib().sugar_tdnn_recalibrate(self.handle(dev), feats.data_ptr(), row_off.data_ptr(), B, R, t_max,
                                                   ws.data_ptr(), ws.numel(), float(self.bn_momentum), emb.data_ptr(),
                                                   stream)
            _lib.check(rc, "sugar_tdnn_recalibrate")
            self.mark_recalibrated(dev)
            if normalize:
                _lib.check(_lib.lib().sugar_l2norm(emb.data_ptr(), B, self.out_embeds, stream), "sugar_l2norm")
            return emb
        rc = _lib.lib().sugar_tdnn_fwd(self.eval_handle(dev), feats.data_ptr(), row_off.data_ptr(), B, R, t_max,
                                       ws.data_ptr(), ws.numel(), emb.data_ptr(), int(normalize), stream)
        _lib.check(rc, "sugar_tdnn_fwd")
        return emb

    # ---- BN recalibration (batch_forward, sugar/models/__init__.py:201-242) ------------------------------------
    bn_momentum = 0.1        # nn.BatchNorm1d default, as DynamicBatchNorm1d builds it (dynamicarch.py:248)

    def bn_names(self):
        """BatchNorm prefixes in the order of the C-ABI's bn_index (include/sugar_amd.h)."""
        names = ["layer1.bn"]
        for i in range(self.subnet_config[0]):
            p = "blocks.%d." % i
            names += [p + "first_conv.bn"] + [p + "resconv.convs.%d.bn" % j for j in range(SCALE - 1)] + [p + "last_conv.bn"]
        return names + ["bnpool", "bnembed"]

    def mark_recalibrated(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, self.dtype_id)
        if self._bn_dirty not in (None, key) or self._bn_unread not in (None, key):
            raise _lib.SugarError("BN recalibration is in progress on another device / precision: call sync_bn() first")
        self._bn_dirty = key
        self._bn_batches += 1

    def sync_bn(self, read_back=True):
        """Fold the recalibrated running statistics into the eval path and copy them into this module's buffers
        (what `model.eval()` / `bn_state_dict` see in the reference).  No-op when nothing was recalibrated.
        read_back=False: fold only -- the eval path is current, the module's buffers are NOT (30 small blocking copies
        and their bookkeeping, ~2 ms during which the GPU idles); for callers that evaluate the subnet and throw it away
        (the NAS sweep).  A later sync_bn() / state_dict() still brings the buffers up to date."""
        if self._bn_dirty is None and self._bn_unread is None:
            return self
        if self._bn_dirty is not None:
            self._bn_unread, self._bn_dirty = self._bn_dirty, None
            _lib.check(_lib.lib().sugar_subnet_bn_commit(self._handles[self._bn_unread]), "sugar_subnet_bn_commit")
        if not read_back:
            return self
        key, self._bn_unread = self._bn_unread, None
        h = self._handles[key]
        buffers = dict(self.named_buffers())
        params = dict(self.named_parameters())
        for i, name in enumerate(self.bn_names()):
            rm = buffers.get(name + ".running_mean", params.get(name + ".running_mean"))
            rv = buffers.get(name + ".running_var", params.get(name + ".running_var"))
            m = torch.empty(rm.numel(), dtype=torch.float32)
            v = torch.empty(rv.numel(), dtype=torch.float32)
            _lib.check(_lib.lib().sugar_subnet_bn_read(h, i, _lib.fptr(m), _lib.fptr(v), m.numel()), "sugar_subnet_bn_read")
            with torch.no_grad():
                rm.copy_(m)
                rv.copy_(v)
                nbt = buffers.get(name + ".num_batches_tracked", params.get(name + ".num_batches_tracked"))
                if nbt is not None:
                    nbt.add_(self._bn_batches)
        self._bn_batches = 0
        for k in [k for k in self._handles if k != key]:        # other devices / precisions hold stale statistics
            _lib.lib().sugar_subnet_destroy(self._handles.pop(k))
        return self

    def reset_bn(self):
        """nn.BatchNorm1d.reset_running_stats on every BN (running_mean 0, running_var 1, num_batches_tracked 0)."""
        self._bn_dirty, self._bn_unread, self._bn_batches = None, None, 0
        with torch.no_grad():
            for k, t in list(self.named_buffers()) + list(self.named_parameters()):
                if k.endswith("running_mean"):
                    t.zero_()
                elif k.endswith("running_var"):
                    t.fill_(1)
                elif k.endswith("num_batches_tracked"):
                    t.zero_()
        self.release()
        return self

    bn_lazy = False          # True: recalibrated statistics are folded into the eval path but copied into this module's
                             # buffers only when someone asks for them (state_dict, sync_bn(), another precision / device)

    def eval_handle(self, device=None):
        """Handle for the eval-mode entry points: recalibrated statistics are folded in first."""
        if self._bn_dirty is not None:
            self.sync_bn(read_back=not self.bn_lazy)
        if self._bn_unread is not None:
            dev = self._device if device is None else torch.device(device)
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if (idx, self.dtype_id) != self._bn_unread:
                self.sync_bn()                         # another handle is asked for: it is packed from the buffers
        return self.handle(device)

    def state_dict(self, *args, **kwargs):
        self.sync_bn()
        return super().state_dict(*args, **kwargs)

    def tap(self, feats, row_off, B, R, t_max, what):
        """Parity tap (tests): returns (activation [rows, C] float32 in the internal layout, embeddings)."""
        dev = feats.device
        ws = self.workspace(dev, B, R)
        emb = torch.empty(B, self.out_embeds, dtype=torch.float32, device=dev)
        out = torch.empty(max(R, B) * 4096, dtype=torch.float32, device=dev)
        c = _lib.check(_lib.lib().sugar_tdnn_tap(self.eval_handle(dev), feats.data_ptr(), row_off.data_ptr(), B, R, t_max,
                                                 ws.data_ptr(), ws.numel(), what, out.data_ptr(), out.numel(),
                                                 emb.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "sugar_tdnn_tap")
        rows = B if what == 101 else R
        return out[:rows * c].view(rows, c), emb


class DynamicTDNN(nn.Module):
    """
    Supernet parameter container with the reference's keys, config protocol and clone (dynamictdnn.py:587-842).
    Forward runs the *active* subnet (``supernet.config = cfg; supernet(x)`` equals ``supernet.clone(cfg)(x)`` in the
    reference); the materialised subnet is cached per config.
    """

    def __init__(self, in_feats=80, out_embeds=192, num_blocks=3, channels=512, kernel_sizes=5,
                 catconv_channels=1536, search_space=None, kernels=(1, 3, 5), depths=(-2, -1, 0),
                 widths=(0.25, 0.35, 0.5, 0.75, 1.0)):
        super().__init__()
        assert isinstance(channels, int) or len(channels) == num_blocks + 1
        assert isinstance(kernel_sizes, int) or len(kernel_sizes) == num_blocks + 1
        self.in_feats, self.out_embeds, self.num_blocks = in_feats, out_embeds, num_blocks
        self.channels = [channels] * (num_blocks + 1) if isinstance(channels, int) else list(channels)
        self.kernel_sizes = [kernel_sizes] * (num_blocks + 1) if isinstance(kernel_sizes, int) else list(kernel_sizes)
        self.catconv_channels = catconv_channels
        for k in self.kernel_sizes:
            assert k in (1, 3, 5), "kernel sizes beyond 5 are not on the EfficientTDNN path"
        _build_tree(self, layout.supernet_shapes(in_feats, out_embeds, num_blocks, self.channels, self.kernel_sizes,
                                                 catconv_channels), _default_init)
        self.search_space = None
        if search_space is not None:
            self.search_space = TDNNPath(self.space(mode=search_space, channel_ratios=widths, kernels=kernels,
                                                    depths=depths, verbose=False))
        self.precision = "bf16"
        self._device = torch.device("cpu")
        self._subnets = {}
        self.config = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._subnets.clear())
        self.eval()

    # parameters stay on the host (see TDNNSubnet)
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0))
        if probe.device != torch.device("cpu"):
            self._device = probe.device
        return self

    # ---- config protocol -------------------------------------------------------------------------------------
    @property
    def config(self):
        return (self.active_depth, self.active_channels, self.active_kernels, self.active_catconv_channels)

    @config.setter
    def config(self, config):
        if config is None:
            config = (None, None, None, None)
        depth, channels, kernels, ccat = config
        self.active_depth = self.num_blocks if depth is None else depth
        self.active_channels = self.channels.copy() if channels is None else list(channels)
        self.active_kernels = self.kernel_sizes.copy() if kernels is None else list(kernels)
        self.active_catconv_channels = self.catconv_channels if ccat is None else ccat
        _check_config(self.config, self.num_blocks, self.channels, self.kernel_sizes, self.catconv_channels)

    def materialize(self, config=None):
        """Cloned-subnet state dict for ``config`` (None = active config): the arithmetic of clone()."""
        cfg = _check_config(self.config if config is None else config, self.num_blocks, self.channels,
                            self.kernel_sizes, self.catconv_channels)
        depth, ch, ks, ccat = cfg
        sup = {k: v.detach() for k, v in _flat_tensors(self).items()}
        c0 = ch[0]
        out = {}                                       # every entry a NEW tensor (TDNNSubnet adopts them uncopied)

        def bn(dst, src, sel):
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                t = sup["%s.bn.%s" % (src, leaf)]
                out["%s.%s" % (dst, leaf)] = t[:sel].clone() if isinstance(sel, int) else t.index_select(0, sel)
            out[dst + ".num_batches_tracked"] = sup[src + ".bn.num_batches_tracked"].clone()

        def conv(dst, src, in_sel, out_sel, k_active):
            full = sup[src + ".conv.weight"]
            w = _active_conv_weight(full, in_sel, out_sel, k_active, sup.get(src + ".k53"), sup.get(src + ".k31"))
            if w.untyped_storage().data_ptr() == full.untyped_storage().data_ptr():
                w = w.clone()                          # a contiguous slice of the supernet's own weight
            out[dst + ".weight"] = w

        conv("layer1.conv", "layer1.conv", self.in_feats, c0, ks[0])              # dynamictdnn.py:670
        bn("layer1.bn", "layer1.bn", c0)
        for i in range(depth):
            p = "blocks.%d." % i
            w = ch[i + 1] // SCALE                                                # :410
            mid_idx = torch.arange(self.channels[i + 1]).view(SCALE, -1)[:, :w].flatten().long()   # :411
            conv(p + "first_conv.conv", p + "first_conv.conv", c0, mid_idx, 1)    # :413
            bn(p + "first_conv.bn", p + "first_conv.bn", mid_idx)
            for j in range(SCALE - 1):
                q = p + "resconv.convs.%d." % j
                conv(q + "conv", q + "conv", w, w, ks[i + 1])                     # :307-308
                bn(q + "bn", q + "bn", w)
            conv(p + "last_conv.conv", p + "last_conv.conv", mid_idx, c0, 1)      # :415
            bn(p + "last_conv.bn", p + "last_conv.bn", c0)
            r = c0 // SE_REDUCTION                                                # :105-106
            out[p + "se.reduce.weight"] = sup[p + "se.reduce.linear.weight"][:r, :c0].clone()
            out[p + "se.reduce.bias"] = sup[p + "se.reduce.linear.bias"][:r].clone()
            out[p + "se.expand.weight"] = sup[p + "se.expand.linear.weight"][:c0, :r].clone()
            out[p + "se.expand.bias"] = sup[p + "se.expand.linear.bias"][:c0].clone()
        cat_idx = torch.arange(self.channels[0] * self.num_blocks).view(self.num_blocks, -1)[:depth, :c0].flatten().long()
        conv("catconv", "catconv", cat_idx, ccat, 1)                              # :672-673
        conv("pooling.linear1", "pooling.linear1", ccat, BOTTLENECK, 1)           # :519-520
        conv("pooling.linear2", "pooling.linear2", BOTTLENECK, ccat, 1)
        pool_idx = torch.arange(self.catconv_channels * 2).view(2, -1)[:, :ccat].flatten().long()   # :675
        bn("bnpool", "bnpool", pool_idx)
        out["linear.weight"] = sup["linear.linear.weight"].index_select(1, pool_idx)                # :677
        out["linear.bias"] = sup["linear.linear.bias"].clone()
        for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out["bnembed." + leaf] = sup["bnembed." + leaf].clone()
        return cfg, out

    def clone(self, config=None):
        """DynamicTDNN.clone (dynamictdnn.py:713-730): sets the config, returns a static subnet."""
        self.config = config
        cfg, state = self.materialize()
        sub = TDNNSubnet(cfg, state, self.in_feats, self.out_embeds, self.precision, adopt=True)
        sub._device = self._device
        return sub

    def active_subnet(self):
        cfg = _check_config(self.config, self.num_blocks, self.channels, self.kernel_sizes, self.catconv_channels)
        key = (layout.config_name(cfg), self.precision)
        if key not in self._subnets:
            _, state = self.materialize(cfg)
            self._subnets[key] = TDNNSubnet(cfg, state, self.in_feats, self.out_embeds, self.precision, adopt=True)
        sub = self._subnets[key]
        sub._device = self._device
        return sub

    def forward(self, x):
        if self.training and self.search_space is not None:
            self.config = self.search_space.sample()                             # dynamictdnn.py:650-652
        return self.active_subnet()(x)

    @property
    def device(self):
        return self._device

    # ---- accounting (known answers in BASELINE.md section 2) -------------------------------------------------
    def count_ops(self, in_size):
        """Multiply-adds of the active subnet incl. the reference's 2-ops-per-element BN convention (:735-765)."""
        depth, ch, ks, ccat = self.config
        n, _, T = in_size
        c0 = ch[0]
        ops = n * c0 * T * (self.in_feats * ks[0]) + 2.0 * n * c0 * T
        for i in range(depth):
            w = ch[i + 1] // SCALE
            mid = w * SCALE
            ops += n * mid * T * c0 + 2.0 * n * mid * T
            ops += (SCALE - 1) * (n * w * T * (w * ks[i + 1]) + 2.0 * n * w * T)
            ops += n * c0 * T * mid + 2.0 * n * c0 * T
            ops += n * (c0 // SE_REDUCTION) * c0 + n * c0 * (c0 // SE_REDUCTION)
        ops += n * ccat * T * (depth * c0)
        ops += n * BOTTLENECK * T * ccat + n * ccat * T * BOTTLENECK
        ops += 2.0 * n * 2 * ccat + n * self.out_embeds * 2 * ccat + 2 * n * self.out_embeds
        return float(ops)

    def count_params(self):
        depth, ch, ks, ccat = self.config
        c0 = ch[0]
        p = c0 * self.in_feats * ks[0] + 2 * c0
        for i in range(depth):
            w = ch[i + 1] // SCALE
            mid = w * SCALE
            r = c0 // SE_REDUCTION
            p += mid * c0 + 2 * mid + (SCALE - 1) * (w * w * ks[i + 1] + 2 * w) + c0 * mid + 2 * c0
            p += (c0 * r + r) + (r * c0 + c0)
        p += ccat * depth * c0 + 2 * ccat * BOTTLENECK + 2 * (2 * ccat) + (self.out_embeds * 2 * ccat + self.out_embeds)
        p += 2 * self.out_embeds
        return float(p)

    def space(self, mode=("kernel", "depth", "width1", "width2"), channel_ratios=(0.25, 0.35, 0.5, 0.75, 1.0),
              kernels=(1, 3, 5), depths=(-2, -1, 0), verbose=True):
        """Search space (dynamictdnn.py:792-842): [depths] + per-layer channel choices + per-layer kernel choices + [catconv]."""
        if "kernel" in mode:
            space_kernels = [[ks for ks in kernels if ks <= kernel] for kernel in self.kernel_sizes]
        else:
            space_kernels = [[kernel] for kernel in self.kernel_sizes]
        space_depth = [self.num_blocks + d for d in depths] if "depth" in mode else [self.num_blocks]
        if "width" in mode or "width1" in mode or "width2" in mode:
            space_channels = [[_make_divisible(rate * c) for rate in channel_ratios] for c in self.channels]
            space_catconv = [_make_divisible(rate * self.catconv_channels) for rate in channel_ratios]
        else:
            space_channels = [[c] for c in self.channels]
            space_catconv = [self.catconv_channels]
        if verbose:
            print("searchable depth: %s" % space_depth)
            print("searchable channels: %s" % space_channels)
            print("searchable kernels: %s" % space_kernels)
            print("searchable catconv: %s" % space_catconv)
        return [space_depth] + space_channels + space_kernels + [space_catconv]


class TDNNPath:
    """Subnet sampler (dynamictdnn.py:845-879); ``random.Random(666)`` and per-dimension uniform choice."""

    def __init__(self, space):
        self.nDim = len(space)
        self.space = space
        self.lower = self._config_map([min(c) for c in space])
        self.upper = self._config_map([max(c) for c in space])
        self.types = ["discrete"] * self.nDim
        self.g = random.Random(666)

    def _config_map(self, config):
        depth = config[0]
        half = (len(config) - 2) // 2
        channels = config[1:1 + half]
        kernels = config[1 + half:1 + 2 * half]
        return (depth, channels[:depth + 1], kernels[:depth + 1], config[-1])

    def sample(self, dist="random"):
        assert dist == "random"
        return self._config_map([self.g.choice(choices) for choices in self.space])


def specfic_tdnn(config):
    depth, channels, kernel_sizes, catconv_channels = config
    return DynamicTDNN(80, 192, depth, channels, kernel_sizes, catconv_channels)


def ecapa_tdnn():
    return DynamicTDNN(80, 192, 3, 512, [5, 3, 3, 3], 1536)


def tdnn8m2g(in_feats=80, out_embeds=192, task="supernet", ks=None, ds=None, ws=None):
    """dynamictdnn.py:894-936 without the NAS tooling (ArchCoder / predictor / latency table are out of scope)."""
    search_space = None if "supernet" in task else task
    kw = {}
    if ks is not None:
        kw["kernels"] = ks
    if ds is not None:
        kw["depths"] = ds
    if ws is not None:
        kw["widths"] = ws
    return DynamicTDNN(in_feats, out_embeds, 4, 512, 5, 1536, search_space=search_space, **kw)


def tdnn14m4g(in_feats=80, out_embeds=192, task="supernet"):
    return DynamicTDNN(in_feats, out_embeds, 4, 800, 5, 1536, search_space=None if "supernet" in task else task)


def tdnn16m4g(in_feats=80, out_embeds=192, task="supernet"):
    return DynamicTDNN(in_feats, out_embeds, 3, 1024, 5, 1536)


def tdnn26m7g(in_feats=80, out_embeds=192):
    return DynamicTDNN(in_feats, out_embeds, 5, 1024, 5, 1536)
