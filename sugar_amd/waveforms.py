"""
Waveform loading of the evaluation front door (sugar/waveforms.py:34-115): same functions, arguments and results.

Host-side numpy; the samples go to the GPU as float32 in the int16 value range, exactly as the reference feeds them
(wavfile.read -> astype(float32), no rescaling: the per-utterance instance norm of the fbank front-end removes the
gain).
"""
import random

import numpy as np
from scipy.io import wavfile

__all__ = ["round_down", "load_wav", "loadWAV", "wav_segment"]


def round_down(num, divisor):
    """sugar/waveforms.py:28-29."""
    return num - (num % divisor)


def load_wav(path, sample_rate=16000):
    """Whole waveform as float32 (sugar/waveforms.py:34-44)."""
    sr, audio = wavfile.read(path)
    assert sample_rate == sr, "%d != %d" % (sr, sample_rate)
    return audio.astype(np.float32)


def loadWAV(path, num_samples=48000, mode_eval=False, num_eval=6, sample_rate=16000, min_samples=48000):
    """
    sugar/waveforms.py:47-115.  Reads a wav; utterances not longer than max(min_samples, num_samples) are wrap-padded
    (np.pad mode="wrap") to one sample more than that; then
      mode_eval False : one random crop of num_samples ([1, num_samples]; random.random() picks the start)
      mode_eval True  : num_eval crops of num_samples starting at np.linspace(0, len - num_samples, num_eval)
                        ([num_eval, num_samples]), or the whole (padded) utterance [1, len] when num_samples == 0.
    Returns float32.
    """
    assert not (min_samples is None and num_samples == 0), "Given min is None with 0 samples"
    sr, audio = wavfile.read(path)
    assert sr == sample_rate, "%d != %d" % (sr, sample_rate)
    audiosize = audio.shape[0]
    floor = min_samples if min_samples is not None and min_samples >= num_samples else num_samples
    if audiosize <= floor:
        audio = np.pad(audio, (0, floor - audiosize + 1), mode="wrap")
        audiosize = audio.shape[0]
    if mode_eval is False:
        start = max(0, np.int64(random.random() * (audiosize - num_samples)))
        wave = audio[start: start + num_samples][None]
    elif num_samples != 0:
        starts = np.linspace(0, audiosize - num_samples, num=num_eval)
        wave = np.stack([audio[int(s): int(s) + num_samples] for s in starts], axis=0)
    else:
        wave = audio[None]
    return wave.astype(np.float32)


def eval_segments(audiosize, num_samples=48000, num_eval=6, min_samples=48000):
    """
    The segments loadWAV(mode_eval=True) cuts from files of `audiosize` samples (int array), WITHOUT reading them:
    (starts [n_files, n_seg], length) in samples of the wrap-padded waveform -- sample i of the padded waveform is
    sample i mod audiosize of the file (np.pad mode="wrap").  num_samples == 0: one segment per file, the whole padded
    utterance (lengths differ per file: returned as an array).  Same arithmetic as loadWAV above (np.linspace on the
    padded size, int() truncation), so a native reader that follows this plan returns loadWAV's samples exactly.
    """
    assert not (min_samples is None and num_samples == 0), "Given min is None with 0 samples"
    size = np.asarray(audiosize, dtype=np.int64)
    floor = min_samples if min_samples is not None and min_samples >= num_samples else num_samples
    padded = np.where(size <= floor, floor + 1, size)
    if num_samples == 0:
        return np.zeros((size.shape[0], 1), dtype=np.int64), padded
    starts = np.linspace(0, padded - num_samples, num=num_eval, axis=-1)      # per file: linspace(0, len - num_samples, num_eval)
    return starts.astype(np.int64), np.full(size.shape[0], num_samples, dtype=np.int64)


def wav_segment(audio, num_samples=48000, noverlap=None):
    """
    sugar/waveforms.py:118-146: wrap-pad the waveform by num_samples - len % num_samples (a whole extra segment when the
    length already divides, as written there) and cut it into rows of num_samples, hop `noverlap` (default: num_samples).
    """
    audio = audio[0] if audio.ndim == 2 else audio
    audio = np.pad(audio, (0, num_samples - (audio.shape[0] % num_samples)), mode="wrap")
    hop = num_samples if noverlap is None else noverlap
    return np.stack([audio[i:i + num_samples] for i in range(0, audio.shape[0], hop)], axis=0)
