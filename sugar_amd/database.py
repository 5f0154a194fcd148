"""
Evaluation lists of the data front door: what `extract_vectors`, `score_trials` and `veri_validate` consume.

The reference keeps these as torch Datasets walked by a DataLoader (sugar/database.py:71-158, 289-364).  This build
stages an evaluation list ONCE on the GPU as packed ragged batches (`vectors.PackedUtterances`), so the classes here
are thin host-side tables: a resolved path list (`Utterance`) and a parsed trial table (`VerificationTrials`).  They
keep the reference's constructor arguments, public attributes and item dictionaries -- the contract the callers and
the notebooks rely on:

    Utterance[i]            -> {"file": path, "input": FloatTensor [n_seg, num_samples]}   (database.py:136-142)
    Utterance.collate_fn(b) -> {"file": [...], "input": cat of the segment stacks, "num": [n_seg per item]}  (:144-158)
    VerificationTrials[i]   -> {"label": int, "enroll": path, "test": path}                (:314-318)
    VerificationTrials.labels / .enrolls / .tests / .files (sorted unique) / .utterances    (:308-309, 343-364)
"""
import os
import random

import torch

from .waveforms import loadWAV

__all__ = ["txt2list", "Utterance", "VerificationTrials"]


def txt2dict(txtfile, txtroot=""):
    """Lines "speaker path" -> {speaker: [txtroot/path, ...]} in file order (sugar/database.py:39-69)."""
    utts = {}
    with open(txtfile, "r") as f:
        for line in f:
            spk, path = line.split(" ")
            utts.setdefault(spk, []).append(os.path.join(txtroot, path.replace("\n", "")))
    return utts


def txt2list(txtfile, txtroot=""):
    """List file, one "[speaker ]path" per line -> [txtroot/path, ...] (sugar/database.py:71-103)."""
    with open(txtfile, "r") as f:
        fields = [line.rstrip("\n").split(" ") for line in f]
    return [os.path.join(txtroot, (fl[0] if len(fl) == 1 else fl[1]).replace("\n", "")) for fl in fields]


class Utterance(torch.utils.data.Dataset):
    """A list of wav paths with loadWAV's segmentation settings; see the module docstring for the item contract."""

    def __init__(self, datalst, num_samples=48000, mode_eval=False, num_eval=2, root_dir=""):
        self.datalst = datalst
        self.root_dir = root_dir
        self.num_samples, self.mode_eval, self.num_eval = num_samples, mode_eval, num_eval
        # the reference distinguishes waveform lists (items carry a segment axis in front of the samples) from
        # feature lists by the first entry's extension; kept as an attribute because callers read it
        self.ndim = 2 - int(".wav" in datalst[0])

    def path(self, idx):
        p = self.datalst[idx]
        return os.path.join(self.root_dir, p) if self.root_dir else p

    def __len__(self):
        return len(self.datalst)

    def __getitem__(self, idx):
        p = self.path(idx)
        segs = loadWAV(p, self.num_samples, self.mode_eval, self.num_eval)
        return {"file": p, "input": torch.from_numpy(segs).float()}

    def segments_of(self, tensor):
        """How many rows of a collated batch an item contributes: its leading axis when it has a segment axis."""
        return tensor.shape[0] if tensor.dim() >= self.ndim else 1

    def collate_fn(self, batch):
        inputs = [it["input"] for it in batch]
        return {"file": [it["file"] for it in batch], "input": torch.cat(inputs, dim=0),
                "num": [self.segments_of(t) for t in inputs]}

    def stage(self, device="cuda:0", batch_size=256, max_frames=262144):
        """This list read once and packed on the GPU (what extract_vectors embeds; reusable across subnets)."""
        from .vectors import PackedUtterances
        return PackedUtterances(self, device=device, batch_size=batch_size, max_frames=max_frames)


def _parse_trial_line(line, trials_dir):
    """"label enroll test" or "enroll test" -> (label, enroll path, test path); a missing label is drawn with
    random.randint(0, 1), one draw per such line in file order, as the reference does (database.py:357)."""
    cols = line.split(" ")
    cols[-1] = cols[-1].replace("\n", "")
    label = random.randint(0, 1) if len(cols) == 2 else cols[0]
    enroll, test = cols[-2], cols[-1]
    return int(label), os.path.join(trials_dir, enroll), os.path.join(trials_dir, test)


class VerificationTrials(torch.utils.data.Dataset):
    """A parsed trial list; see the module docstring for the attributes and the item contract."""

    def __init__(self, trials_lst=None, num_samples=0, mode_eval=True, num_eval=1, trials_dir=None):
        super().__init__()
        self.num_samples, self.mode_eval, self.num_eval = num_samples, mode_eval, num_eval
        self.load_trial(trials_lst, trials_dir)
        self.utterances = Utterance(self.files, num_samples, mode_eval, num_eval, trials_dir)

    def load_trial(self, trials_lst, trials_dir):
        with open(trials_lst, "r") as f:
            rows = [_parse_trial_line(line, trials_dir) for line in f if line]
        self.labels = [r[0] for r in rows]
        self.enrolls = [r[1] for r in rows]
        self.tests = [r[2] for r in rows]
        self.files = sorted(set(self.enrolls).union(self.tests))

    def trials2utts(self):
        return self.files

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        return {"label": self.labels[idx], "enroll": self.enrolls[idx], "test": self.tests[idx]}
