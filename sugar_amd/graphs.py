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
    def __init__(self, model, example_wav, normalize=False, warmup=2):
        if not example_wav.is_cuda or example_wav.dim() != 2:
            raise _lib.SugarError("GraphedEmbedder: example_wav must be a [B, L] CUDA tensor")
        self.model = model
        self.normalize = bool(normalize)
        self.shape = tuple(example_wav.shape)
        self.device = example_wav.device
        self.static_in = example_wav.detach().float().contiguous().clone()
        # Eager passes first, on a side stream as torch's capture rules ask: they build the packed-weight handle and the
        # batch plan (host -> device copies), and opt the kernels into their LDS sizes -- none of which may happen
        # while a stream is capturing.
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, int(warmup))):
                model(self.static_in, normalize=self.normalize)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = model(self.static_in, normalize=self.normalize)

    def __call__(self, wav):
        if tuple(wav.shape) != self.shape or wav.device != self.device:
            raise _lib.SugarError("GraphedEmbedder: captured for %s on %s, got %s on %s"
                                  % (self.shape, self.device, tuple(wav.shape), wav.device))
        self.static_in.copy_(wav, non_blocking=True)
        self.graph.replay()
        return self.static_out
