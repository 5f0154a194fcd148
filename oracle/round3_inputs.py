"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  What the round-3 stage-tap fixture (oracle/make_golden_round3.py, which needs
/root/reference) and its GPU test (which does not) share: the networks and the frames every tap is stored at.
"""
from sugar_amd import layout

# name: (reference factory, layout.supernet_shapes arguments, subnet config)
NETS = {
    "base": ("tdnn8m2g", (80, 192, 4, 512, 5, 1536), layout.BASE),
    "wide_full": ("tdnn16m4g", (80, 192, 3, 1024, 5, 1536), (3, [1024] * 4, [5] * 4, 1536)),
}
# 24 of the 301 frames: both ends (zero "same" padding, reflect-padded STFT) and a spread over the interior
FRAMES = [0, 1, 2, 3, 7, 15, 16, 31, 32, 63, 64, 100, 127, 128, 150, 191, 192, 222, 255, 256, 290, 297, 299, 300]
