"""
Front-end transforms with the reference's interface (sugar/transforms.py:5-77), computed by the fused HIP
front-end (sugar_fbank_fwd).  There is no torch / torchaudio arithmetic on this path.
"""
import torch
import torch.nn as nn

from . import _lib

HOP = 160


def num_frames(num_samples: int) -> int:
    """center=True STFT with hop 160: T = 1 + floor(L / 160)."""
    return 1 + int(num_samples) // HOP


class BatchPlan:
    """
    Device-side index arrays the C-ABI needs for one batch shape: sample offsets / lengths and frame offsets.
    Built on the host from the (host-known) utterance lengths and cached per shape by the callers.
    """

    def __init__(self, lengths, device):
        lengths = [int(n) for n in lengths]
        if min(lengths) <= 256:
            raise ValueError("utterances must be longer than 256 samples (reflect padding of the centred STFT); "
                             "the reference pads short audio to 48 001 samples (sugar/waveforms.py:94-98)")
        frames = [num_frames(n) for n in lengths]
        wav_off, row_off = [0], [0]
        for n, t in zip(lengths, frames):
            wav_off.append(wav_off[-1] + n)
            row_off.append(row_off[-1] + t)
        self.B = len(lengths)
        self.R = row_off[-1]
        self.t_max = max(frames)
        self.total_samples = wav_off[-1]
        self.lengths = lengths
        self.frames = frames
        self.wav_off = torch.tensor(wav_off[:-1], dtype=torch.int64).to(device)
        self.wav_len = torch.tensor(lengths, dtype=torch.int32).to(device)
        self.row_off = torch.tensor(row_off, dtype=torch.int32).to(device)


class PreEmphasis(nn.Module):
    """Kept for interface parity (sugar/transforms.py:5-27); pre-emphasis itself is fused into the DFT kernel's loader."""

    def __init__(self, coef: float = 0.97):
        super().__init__()
        if abs(coef - 0.97) > 1e-12:
            raise ValueError("the fused front-end implements the reference's fixed coefficient 0.97")
        self.coef = coef


class LogMelFbanks(nn.Module):
    """
    LogMelFbanks(n_mels=80) (sugar/transforms.py:30-77): [B, L] waveforms -> [B, 1, 80, T] float32
    log-mel features with per-utterance mean/variance normalisation.
    The SpecAugment ``masker`` branch (:55-57, 74-76) is training-only and not provided.
    """

    def __init__(self, n_mels: int = 80, log_input: bool = True, autocast: bool = False, masker: bool = False):
        super().__init__()
        if n_mels != 80 or not log_input or masker:
            raise ValueError("the fused front-end implements n_mels=80, log_input=True, masker=False")
        self.n_mels = n_mels
        self._plans = {}

    def plan(self, lengths, device):
        key = (tuple(lengths), str(device))
        if key not in self._plans:
            if len(self._plans) > 64:
                self._plans.clear()
            self._plans[key] = BatchPlan(lengths, device)
        return self._plans[key]

    def rows(self, wav_flat, plan, dtype=torch.float32):
        """Flat float32 samples on the GPU -> channels-last packed features [R, 80] in ``dtype`` (float32 or bfloat16)."""
        if not wav_flat.is_cuda:
            raise _lib.SugarError("sugar_amd runs on an MI355X only: waveforms must be CUDA tensors (no CPU fallback)")
        dev = wav_flat.device
        nbytes = _lib.lib().sugar_fbank_scratch_bytes(plan.B, plan.R)
        if nbytes < 0:
            _lib.check(int(nbytes), "sugar_fbank_scratch_bytes")
        logmel = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        feats = torch.empty(plan.R, self.n_mels, dtype=dtype, device=dev)
        rc = _lib.lib().sugar_fbank_fwd(wav_flat.data_ptr(), plan.wav_off.data_ptr(), plan.wav_len.data_ptr(),
                                        plan.row_off.data_ptr(), plan.B, plan.R, plan.t_max, logmel.data_ptr(),
                                        feats.data_ptr(), _lib.SUGAR_F32 if dtype == torch.float32 else _lib.SUGAR_BF16,
                                        torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "sugar_fbank_fwd")
        return feats

    def forward(self, x):
        """x: [B, L] -> [B, 1, 80, T] float32, as the reference."""
        if x.dim() != 2:
            raise ValueError("expected [batch, wav_length], got %s" % (tuple(x.shape),))
        B, L = x.shape
        plan = self.plan([L] * B, x.device)
        feats = self.rows(x.float().contiguous().view(-1), plan)          # [B*T, 80]
        return feats.view(B, plan.t_max, self.n_mels).permute(0, 2, 1).unsqueeze(1)
