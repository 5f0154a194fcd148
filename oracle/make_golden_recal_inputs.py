"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Cases and seeded waveform batches shared by the BN-recalibration fixture
generator (oracle/make_golden_recal.py) and the tests; inputs regenerate from seeds, never stored."""
from sugar_amd import layout, synth

# name -> (config, utterances per batch, samples per utterance, batches)
CASES = {
    "smallest": (layout.SMALLEST, 6, 16000, 2),
    "base": (layout.BASE, 4, 10000, 2),
    "odd": ((3, [176, 384, 128, 176], [3, 5, 1, 3], 536), 5, 12000, 1),
}
EVAL_SEED = 77


def batches(name, n_batches, B, L):
    return [synth.waveforms(B, L, seed=1000 + 17 * i + len(name)) for i in range(n_batches)]
