"""
Model API with the reference's names and call signatures (sugar/models/__init__.py:44-165):

  SpeakerModel(modelarch, lossfunc=None, transform=None)   -- attributes ``__S__`` / ``__L__`` / ``__T__`` (:61-63)
  WrappedModel(model).module                                -- DDP-shaped wrapper (:117-127)
  WrappedModel.from_pretrained(...) -> (subnet, info)       -- same argument names and load order (:129-165)

Differences, all forced by the environment or by the scope of this build:
  * there is no network: ``supernet_filename`` / ``subnet_filename`` are LOCAL paths (``repo_id`` is a local
    directory they are resolved against, or "" / None); file formats are the reference's
    (``{'state_dict': {'module.__S__.*': ...}}`` and ``{'subnet': config, 'bn': {...}}``);
  * ``precision`` ("bf16" default, "fp32") selects the MFMA path;
  * ``forward`` also accepts a list of 1-D waveforms of different lengths (the reference runs those at batch 1);
  * the loss head (``__L__``) is not on the inference path: ``label`` must be None.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from .dynamictdnn import DynamicTDNN, TDNNSubnet, tdnn8m2g
from .transforms import LogMelFbanks


class SpeakerModel(nn.Module):
    def __init__(self, modelarch, lossfunc=None, transform=None, keep_channel=False, features_last=False):
        super().__init__()
        if keep_channel or features_last:
            raise ValueError("keep_channel / features_last are not used by the EfficientTDNN path")
        self.__S__ = modelarch
        self.__L__ = lossfunc
        self.__T__ = transform
        self.keep_channel = keep_channel
        self.features_last = features_last

    def __repr__(self):
        n = sum(p.numel() for p in self.__S__.parameters())
        return "Arch: %d" % n

    def _subnet(self):
        s = self.__S__
        if isinstance(s, DynamicTDNN):
            if s.training:
                raise _lib.SugarError("BN recalibration runs on a cloned subnet: use supernet.clone(config), as the "
                                      "callers of batch_forward do (SubnetEvaluation.ipynb, train_vox2_veri.py:395)")
            return s.active_subnet()
        return s

    def embed_ragged(self, wavs, normalize=False):
        """List of 1-D float waveforms (any lengths > 256 samples) -> [len(wavs), 192]; one fused C-ABI call."""
        if not isinstance(self.__T__, LogMelFbanks) or not isinstance(self.__S__, (TDNNSubnet, DynamicTDNN)):
            raise _lib.SugarError("embed_ragged needs the fused LogMelFbanks + TDNN path")
        sub = self._subnet()
        dev = wavs[0].device
        if dev.type != "cuda":
            raise _lib.SugarError("sugar_amd runs on an MI355X only: waveforms must be CUDA tensors (no CPU fallback)")
        plan = self.__T__.plan([int(w.numel()) for w in wavs], dev)
        flat = torch.cat([w.reshape(-1).float() for w in wavs]) if len(wavs) > 1 else wavs[0].reshape(-1).float().contiguous()
        return self._embed_flat(sub, flat, plan, normalize)

    def _embed_flat(self, sub, flat, plan, normalize):
        dev = flat.device
        ws = sub.workspace(dev, plan.B, plan.R)
        emb = torch.empty(plan.B, sub.out_embeds, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if sub.training:          # one BN-recalibration step (batch_forward): batch statistics, running stats updated
            rc = _lib.lib().sugar_embed_recalibrate(sub.handle(dev), flat.data_ptr(), plan.wav_off.data_ptr(),
                                                    plan.wav_len.data_ptr(), plan.row_off.data_ptr(), plan.B, plan.R,
                                                    plan.t_max, ws.data_ptr(), ws.numel(), float(sub.bn_momentum),
                                                    emb.data_ptr(), stream)
            _lib.check(rc, "sugar_embed_recalibrate")
            sub.mark_recalibrated(dev)
            if normalize:
                _lib.check(_lib.lib().sugar_l2norm(emb.data_ptr(), plan.B, sub.out_embeds, stream), "sugar_l2norm")
            return emb
        rc = _lib.lib().sugar_embed_fwd(sub.eval_handle(dev), flat.data_ptr(), plan.wav_off.data_ptr(),
                                        plan.wav_len.data_ptr(), plan.row_off.data_ptr(), plan.B, plan.R, plan.t_max,
                                        ws.data_ptr(), ws.numel(), emb.data_ptr(), int(normalize),
                                        torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "sugar_embed_fwd")
        return emb

    def forward(self, batch, label=None, normalize=False):
        """
        batch: [B, L], [B, 1, L] or [B, U, L] waveforms (sugar/models/__init__.py:79-94), or pre-computed features
        when transform is None, or a list of 1-D waveforms.  Returns embeddings [B*U, 192].
        """
        if label is not None:
            raise ValueError("the loss head is outside the embedding-extraction path: label must be None")
        if isinstance(batch, (list, tuple)):
            return self.embed_ragged(list(batch), normalize)
        sizes = list(batch.size())
        if len(sizes) == 2 or (len(sizes) == 3 and sizes[1] != 1):
            batch = batch.unsqueeze(1)
            sizes = list(batch.size())
        batch = batch.reshape([-1] + sizes[2:])
        if self.__T__ is None:
            emb = self.__S__(batch.squeeze(1) if batch.dim() == 4 else batch)
            if normalize:
                _lib.check(_lib.lib().sugar_l2norm(emb.data_ptr(), emb.shape[0], emb.shape[1],
                                                   torch.cuda.current_stream(emb.device).cuda_stream), "sugar_l2norm")
            return emb
        if isinstance(self.__T__, LogMelFbanks) and isinstance(self.__S__, (TDNNSubnet, DynamicTDNN)) and batch.dim() == 2:
            if not batch.is_cuda:
                raise _lib.SugarError("sugar_amd runs on an MI355X only: waveforms must be CUDA tensors (no CPU fallback)")
            B, L = batch.shape
            plan = self.__T__.plan([L] * B, batch.device)
            return self._embed_flat(self._subnet(), batch.float().contiguous().view(-1), plan, normalize)
        feats = self.__T__(batch).squeeze(1)
        return self.__S__(feats)


class WrappedModel(nn.Module):
    """sugar/models/__init__.py:117-165."""

    def __init__(self, model):
        super().__init__()
        self.module = model

    def forward(self, x, label=None, normalize=False):
        return self.module(x, label, normalize=normalize)

    @classmethod
    def from_pretrained(cls, repo_id="", supernet_filename="depth/depth.torchparams",
                        subnet_filename="depth/depth.ecapa-tdnn.3.512.512.512.512.5.3.3.3.1536.bn.tar",
                        feat_type="logmelfbank", feat_dim=80, model_type="tdnn8m2g", embed_dim=192,
                        precision="bf16"):
        """(subnet, info) from a supernet checkpoint and a subnet BN file; argument names, file formats and the
        returned `info` keys are the reference's (sugar/models/__init__.py:129-165), the files are local."""
        front_end = _choice("feat_type", feat_type, _FRONT_ENDS)
        backbone = _choice("model_type", model_type, _BACKBONES)
        _choice("precision", precision, {"bf16": None, "fp32": None})
        info = {"supernet_file": _local_file(repo_id, supernet_filename), "subnet_file": None, "subnet": None}
        info["supernet"] = torch.load(info["supernet_file"], map_location="cpu", weights_only=True)["state_dict"]
        transform = front_end(n_mels=feat_dim)
        supernet = backbone(in_feats=feat_dim, out_embeds=embed_dim)
        supernet.precision = precision
        info["supernet_model"] = cls(SpeakerModel(supernet, transform=transform))
        # keys of the training-time wrapper that are not ours (module.__L__.W, module.__T__.*) are ignored, as upstream
        info["supernet_model"].load_state_dict(info["supernet"], strict=False)
        if subnet_filename in ("", None):
            return None, info
        info["subnet_file"] = _local_file(repo_id, subnet_filename)
        info["subnet"] = torch.load(info["subnet_file"], map_location="cpu", weights_only=True)
        subnet = supernet.clone(info["subnet"]["subnet"])                # config tuple -> static subnet (:158)
        subnet.load_state_dict(info["subnet"]["bn"], strict=False)      # recalibrated BN statistics (:159)
        return cls(SpeakerModel(subnet, transform=transform)), info


_FRONT_ENDS = {"logmelfbank": LogMelFbanks}
_BACKBONES = {"tdnn8m2g": tdnn8m2g}


def _choice(what, value, table):
    if value not in table:
        raise ValueError("%s=%r: this build provides %s" % (what, value, sorted(table)))
    return table[value]


def _local_file(repo_id, name):
    """There is no network here: `repo_id` is a local directory (or empty) the file name is resolved against."""
    path = name if (os.path.isabs(name) or not repo_id) else os.path.join(repo_id, name)
    if not os.path.isfile(path):
        raise FileNotFoundError("%s: no network in this build -- pass local files (repo_id is treated as a local "
                                "directory)" % path)
    return path


def checkpoint2backbone(state_dict):
    """Strip ``module.__S__.`` from a checkpoint's state dict (sugar/models/__init__.py:168-198, without the file I/O)."""
    return {k.replace("module.__S__.", ""): v for k, v in state_dict.items() if "__S__" in k}


def bn_state_dict(model, removed_prefix=""):
    """BN-only state dict of a subnet: the ``'bn'`` entry of a ``*.bn.tar`` file (sugar/utils/utility.py:9-23)."""
    keys = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")
    out = {}
    for k, v in model.state_dict().items():
        prefix, _, leaf = k.rpartition(".")
        if leaf in keys and (prefix.endswith(".bn") or prefix.endswith("bnpool") or prefix.endswith("bnembed")):
            out[k.replace(removed_prefix, "")] = v
    return out


def load_bn_state_dict(model, state_dict):
    """Load a BN-only state dict (the 'bn' entry of a *.bn.tar file) into a cloned subnet: every entry must match a
    tensor of the model by name and size, else AssertionError naming the rest (sugar/utils/utility.py:26-42)."""
    own = model.state_dict()
    unloaded = []
    for name, param in state_dict.items():
        if name in own and own[name].size() == param.size():
            own[name].copy_(param)
        else:
            unloaded.append(name)
    assert len(unloaded) == 0, f"{unloaded} NOT Loaded!"
    model.load_state_dict(own)


def keep_params(state_dict_path, params_path):
    """Strip optimizer / scheduler / lr_step from a training checkpoint: what a `*.torchparams` file, the supernet file of
    from_pretrained, holds (sugar/utils/utility.py:67-75).  Refuses to overwrite.  The checkpoint is read with
    weights_only=True (tensors and plain containers only)."""
    assert os.path.exists(state_dict_path) and not os.path.exists(params_path)
    state_dict = torch.load(state_dict_path, map_location="cpu", weights_only=True)
    assert "state_dict" in state_dict
    for k in ("optimizer", "scheduler", "lr_step"):
        del state_dict[k]
    torch.save(state_dict, params_path)


def veri_validate(loader, model, args=None, tboard=None, norm=True, p_target=0.05, device="cuda:0", mode="cosine",
                  ret_info=False, vectors=None):
    """
    Speaker-verification evaluation (sugar/models/__init__.py:535-595): extract the vectors of
    `loader.dataset.utterances` (skipped when `vectors` is given), score `loader.dataset`'s trials, return
    ((eer, thresh), (mindcf, thresh)) -- plus (vectors, scores) with ret_info=True.  `loader` is anything with the
    reference DataLoader's attributes (`dataset` iterable of {"label", "enroll", "test"} with `.utterances`,
    `batch_size`, `num_workers`); `args.gpu` selects the device as in the reference; `tboard` receives the same
    two scalars when given.
    """
    from .metrics import calculate_eer, calculate_mindcf
    from .scores import score_trials
    from .vectors import extract_vectors

    val_task = "evaluate"
    if args is not None:
        val_task = "test" if getattr(args, "evaluate", False) is True else "validate"
        if getattr(args, "gpu", None) is None:
            raise _lib.SugarError("sugar_amd runs on an MI355X only: args.gpu must name a device (no CPU fallback)")
        device = "cuda:{}".format(args.gpu)
    name = "Cuda:{}".format(device.split(":")[1]) if ":" in device else "Cuda:0"
    if vectors is None:
        with torch.no_grad():
            vectors = extract_vectors(model, loader.dataset.utterances, norm=norm, device=device,
                                      n_jobs=getattr(loader, "num_workers", 0), batch_size=loader.batch_size)
    scs, labs, _ = score_trials(loader.dataset, vectors, mode=mode, n_jobs=1)
    eer = calculate_eer(labs, scs)
    dcf = calculate_mindcf(labs, scs, p_target=p_target)
    if tboard is not None:
        tboard.add_scalars("Metric/%s/EER" % val_task, {name: eer[0]}, args.epoch)
        tboard.add_scalars("Metric/%s/DCF" % val_task, {name: dcf[0]}, args.epoch)
    if not ret_info:
        return eer, dcf
    return eer, dcf, vectors, scs


class _BatchFeed:
    """
    Host batches of a DataLoader -> the device without stalling the forward that runs meanwhile: a batch in pageable host
    memory is packed into one of three PINNED staging buffers (native multi-threaded copy, sugar_pack_segments) and
    sent on a copy stream; the compute stream only waits for that copy's event.  `tensor.to(device)` from pageable memory
    is a synchronous ~10 GB/s copy on the compute stream: 5 ms for a 256 x 3 s batch whose train-mode forward takes 1.8.
    The recalibration steps themselves stay strictly in order on one stream (running statistics are updated step by step).
    """
    SLOTS = 3

    def __init__(self, device):
        self.device = device
        self.pinned, self.dev = [None] * self.SLOTS, [None] * self.SLOTS
        self.h2d = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.free = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.copy_stream = None
        self.k, self.slot = 0, None

    def to_device(self, inp):
        self.slot = None
        if inp.is_cuda or inp.numel() == 0:
            return inp.to(self.device)
        import ctypes
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream(self.device)
        s, n = self.k % self.SLOTS, inp.numel()
        if self.k >= self.SLOTS:
            self.free[s].synchronize()                 # the forward that read this slot three batches ago has run
        if self.pinned[s] is None or self.pinned[s].numel() < n:
            self.pinned[s] = torch.empty(n, dtype=torch.float32, pin_memory=True)
            self.dev[s] = torch.empty(n, dtype=torch.float32, device=self.device)
            # the caching allocator may hand out memory that kernels already queued on the CURRENT stream still use
            # (stream-ordered reuse); the copy stream writes into it, so it must run behind them
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        if inp.dtype == torch.float32 and inp.is_contiguous():
            src = (ctypes.c_void_p * 1)(inp.data_ptr())
            lens = (ctypes.c_int64 * 1)(n)
            _lib.check(_lib.lib().sugar_pack_segments(src, lens, 1, self.pinned[s].data_ptr(), 8), "sugar_pack_segments")
        else:
            self.pinned[s][:n].copy_(inp.reshape(-1))
        with torch.cuda.stream(self.copy_stream):
            self.dev[s][:n].copy_(self.pinned[s][:n], non_blocking=True)
            self.h2d[s].record(self.copy_stream)
        torch.cuda.current_stream(self.device).wait_event(self.h2d[s])
        self.slot = s
        self.k += 1
        return self.dev[s][:n].view(inp.shape)

    def consumed(self):
        """The forward reading the last batch has been enqueued on the current stream."""
        if self.slot is not None:
            self.free[self.slot].record(torch.cuda.current_stream(self.device))

    def close(self):
        torch.cuda.current_stream(self.device).synchronize()     # the staging buffers outlive the forwards that read them


def batch_forward(loader, model, args=None, device="cuda:0"):
    """
    BN recalibration (sugar/models/__init__.py:201-242): forward the model in train() mode over `loader` so every
    BatchNorm of the cloned subnet re-estimates its running statistics, then restore the model's mode.  `loader`
    yields batches whose first element is the waveform tensor [B, L] / [B, 1, L]; `args.gpu` selects the device as in
    the reference.  The statistics are folded into the eval path (and copied into the module's buffers) on return.
    """
    if args is not None and getattr(args, "gpu", None) is not None:
        device = "cuda:{}".format(args.gpu)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
    origin_state = model.training
    with torch.no_grad():
        model.train()
        model.to(dev)
        feed = _BatchFeed(dev)
        for batch in loader:
            inp = batch[0] if isinstance(batch, (list, tuple)) else batch
            model.forward(feed.to_device(inp), None)
            feed.consumed()
        feed.close()
        model.train(origin_state)
    spk = model.module if hasattr(model, "module") else model
    sub = getattr(spk, "__S__", None)
    if isinstance(sub, TDNNSubnet):
        sub.sync_bn(read_back=not sub.bn_lazy)
    return model
