"""
HIP-graph replay of the embedding forward for launch-bound batches (small B, serving loops).

One forward is ~35 kernel launches on one stream.  At 256 x 3 s they take 1.2 ms and the launches are free; at one
3 s utterance the kernels themselves need ~0.15 ms and the host-side launch path (Python -> ctypes -> 35
hipLaunchKernel calls) is what the caller waits for.  `sugar_embed_fwd` only enqueues -- no allocation, no
synchronisation, every buffer passed in -- so the whole call can be captured into a hipGraph once per input shape and
replayed with one launch.  (The reference has no counterpart: its latency table, sugar/metrics.py:362-410, times eager
PyTorch modules.)

    g = GraphedEmbedder(model, example_wav)       # model: SpeakerModel / WrappedModel, example_wav: [B, L] on the GPU
    emb = g(wav)                                  # same shape as example_wav -> [B, 192]; bit-identical to model(wav)

The returned tensor is the graph's static output buffer: it is overwritten by the next call (clone it to keep it).
Replays of one GraphedEmbedder must not overlap each other (they share the static buffers and the workspace).
"""
import torch

from . import _lib


class GraphedEmbedder:
    def __init__(self, model, example_wav, normalize=False, warmup=2, on_stale="recapture"):
        """on_stale: what a call does when the model's packed weights were released or re-packed since the capture
        (load_state_dict, BN recalibration + sync, set_precision, a sweep's sub.release(), another active config of a
        supernet): "recapture" (default) captures again on the new weights, "raise" raises SugarError.  Either way a
        graph whose baked-in device pointers are dead is never replayed."""
        if not example_wav.is_cuda or example_wav.dim() != 2:
            raise _lib.SugarError("GraphedEmbedder: example_wav must be a [B, L] CUDA tensor")
        if on_stale not in ("recapture", "raise"):
            raise ValueError("on_stale %r not in ['recapture', 'raise']" % (on_stale,))
        self.model = model
        self.normalize = bool(normalize)
        self.warmup = max(1, int(warmup))
        self.on_stale = on_stale
        self.shape = tuple(example_wav.shape)
        self.device = example_wav.device
        self.static_in = example_wav.detach().float().contiguous().clone()
        self.captures = 0
        self._capture()

    def _subnet(self):
        spk = self.model.module if hasattr(self.model, "module") else self.model
        return spk._subnet()

    def _token(self):
        """Identity of everything the graph bakes in: the subnet object, its generation (release / precision change)
        and the packed-weight handle of this device."""
        sub = self._subnet()
        return (id(sub), sub.generation, sub.precision, sub.eval_handle(self.device).value)

    def _capture(self):
        model = self.model
        # Eager passes first, on a side stream as torch's capture rules ask: they build the packed-weight handle and the
        # batch plan (host -> device copies), and opt the kernels into their LDS sizes -- none of which may happen
        # while a stream is capturing.
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):
                model(self.static_in, normalize=self.normalize)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        sub = self._subnet()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = model(self.static_in, normalize=self.normalize)
        # the workspace the captured kernels write is the subnet's cache entry of the capture stream: keep it alive
        # here too (the cache drops entries on release() and when a caller cycles through many streams)
        self._keep = list(sub._workspace.values())
        self._captured = self._token()
        self.captures += 1

    def __call__(self, wav):
        if tuple(wav.shape) != self.shape or wav.device != self.device:
            raise _lib.SugarError("GraphedEmbedder: captured for %s on %s, got %s on %s"
                                  % (self.shape, self.device, tuple(wav.shape), wav.device))
        if self._token() != self._captured:
            if self.on_stale == "raise":
                raise _lib.SugarError("GraphedEmbedder: the model's packed weights changed since the capture "
                                      "(load_state_dict / recalibration / set_precision / release); capture again")
            self.graph = None                     # drop the dead graph before its pool is reused
            self._capture()
        self.static_in.copy_(wav, non_blocking=True)
        self.graph.replay()
        return self.static_out
