"""
Evaluation datasets of the data front door (sugar/database.py:71-158, 289-364): the utterance list, the ragged
collate bookkeeping and the trial-list parser that `extract_vectors`, `score_trials` and `veri_validate` consume.
Host-side python; same class names, attributes and item dictionaries as the reference.
"""
import os
import random

import torch
from torch.utils.data import Dataset

from .waveforms import loadWAV

__all__ = ["txt2list", "Utterance", "VerificationTrials"]


def txt2list(txtfile, txtroot=""):
    """List file "[speaker ]path" per line -> [txtroot/path, ...] (sugar/database.py:71-103)."""
    utts = []
    with open(txtfile, "r") as f:
        for line in f:
            parts = line.split(" ")
            path = parts[0] if len(parts) == 1 else parts[1]
            utts.append(os.path.join(txtroot, path.replace("\n", "")))
    return utts


class Utterance(Dataset):
    """
    Utterances from a list of wav paths (sugar/database.py:105-158).  Item: {"file": path, "input": FloatTensor
    [n_seg, num_samples]} with loadWAV's segmentation; collate_fn concatenates the segments of a batch and records
    how many each utterance contributed ("num"), which extract_vectors uses to split the embeddings again.
    """

    def __init__(self, datalst, num_samples=48000, mode_eval=False, num_eval=2, root_dir=""):
        self.datalst = datalst
        self.num_samples = num_samples
        self.mode_eval = mode_eval
        self.num_eval = num_eval
        self.root_dir = root_dir
        self.ndim = 1 if ".wav" in datalst[0] else 2          # ".wav" inputs a waveform

    def __len__(self):
        return len(self.datalst)

    def __getitem__(self, idx):
        path = self.datalst[idx]
        if self.root_dir:
            path = os.path.join(self.root_dir, path)
        return {"file": path, "input": torch.from_numpy(loadWAV(path, self.num_samples, self.mode_eval, self.num_eval))}

    def collate_fn(self, batch):
        path = [item["file"] for item in batch]
        wave = torch.cat([item["input"] for item in batch], dim=0)
        num = [item["input"].shape[0] if item["input"].ndim >= self.ndim else 1 for item in batch]
        return {"file": path, "input": wave, "num": num}


class VerificationTrials(Dataset):
    """
    Trial list "label enroll test" per line (sugar/database.py:289-364).  `labels`, `enrolls`, `tests`, the sorted
    unique `files` and `utterances` (an Utterance over them) as in the reference; a missing label becomes a random
    0/1 (random.randint, as there).
    """

    def __init__(self, trials_lst=None, num_samples=0, mode_eval=True, num_eval=1, trials_dir=None):
        super().__init__()
        self.num_samples = num_samples
        self.mode_eval = mode_eval
        self.num_eval = num_eval
        self.load_trial(trials_lst, trials_dir)
        self.utterances = Utterance(self.files, num_samples, mode_eval, num_eval, trials_dir)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        return {"label": self.labels[idx], "enroll": self.enrolls[idx], "test": self.tests[idx]}

    def trials2utts(self):
        return self.files

    def load_trial(self, trials_lst, trials_dir):
        self.labels, self.enrolls, self.tests = [], [], []
        with open(trials_lst, "r") as f:
            for line in f:
                data = line.split(" ")
                data[-1] = data[-1].replace("\n", "")
                if len(data) == 2:                             # append a random label if missing
                    data = [random.randint(0, 1)] + data
                self.labels.append(int(data[0]))
                self.enrolls.append(os.path.join(trials_dir, data[1]))
                self.tests.append(os.path.join(trials_dir, data[2]))
        self.files = sorted(set(self.enrolls + self.tests))
