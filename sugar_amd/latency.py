"""
Latency table of the dynamic TDNN for MI355X (TDNNLatency, sugar/models/dynamictdnn.py:1281-1441; latency(),
sugar/metrics.py:362-410).

Same table format (yaml: {unit: {"(k1=v1, k2=v2)": milliseconds}}), same `query` arithmetic and the same units as the
reference (layer1 / blocks / catconv / pool / embed; cat, bnpool, bnembed and the dilation are not modelled).  What
changes is how a unit is measured: the reference instantiates each nn.Module and wall-clocks it; here a unit is a
group of stages of the fused C-ABI forward, bracketed by hipEvents on the launch stream (sugar_profile_*), on a
minimal subnet that contains the unit with the requested sizes.
"""
import itertools
import os
import time

import torch
import yaml

from . import _lib
from .dynamictdnn import TDNNSubnet

__all__ = ["latency", "TDNNLatency", "COARSE_GRID"]

_WIDTHS = [128, 176, 256, 384, 512]
_CATS = [384, 536, 768, 1152, 1536]
# the coarse search space of tdnn8m2g (dynamictdnn.py:792-842): what TDNNLatency.measure() covers by default
COARSE_GRID = {
    "layer1": {"kwargs": {"in_channels": [80], "out_channels": _WIDTHS, "kernel_size": [1, 3, 5]}},
    "blocks": {"kwargs": {"channels": _WIDTHS, "mid_channels": _WIDTHS, "kernel_size": [1, 3, 5]}},
    "catconv": {"kwargs": {"in_channels": sorted({c * d for c in _WIDTHS for d in (2, 3, 4)}), "out_channels": _CATS}},
    "pool": {"kwargs": {"in_dim": _CATS}},
    "embed": {"kwargs": {"in_features": [2 * c for c in _CATS], "out_features": [192]}},
}
_UNIT_STAGES = {
    "layer1": ("layer1",),
    "blocks": ("first_conv", "res2_steps", "last_conv", "se_scale", "se_apply"),
    "catconv": ("catconv",),
    "pool": ("pool_linear1", "pool_stats"),
    "embed": ("tail+l2norm",),
}


def latency(model, input_size, device="cuda:0", warmup_steps=10, measure_steps=50):
    """sugar/metrics.py:362-410: mean wall-clock milliseconds of `model(inp)` on a random input of `input_size`
    (warm-up, device sync, timed loop, sync); the model's train/eval state is restored."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
    origin_state = model.training
    model.eval()
    model.to(dev)
    with torch.no_grad():
        inp = torch.rand(*input_size, device=dev)
        torch.cuda.synchronize(dev)
        for _ in range(warmup_steps):
            model(inp)
        torch.cuda.synchronize(dev)
        st = time.time()
        for _ in range(measure_steps):
            model(inp)
        torch.cuda.synchronize(dev)
        total = time.time() - st
    model.train(origin_state)
    return total / measure_steps * 1000


class TDNNLatency(object):
    def __init__(self, table=None):
        if table is not None:
            if not os.path.exists(table):
                raise FileExistsError(table)
            with open(table, "r") as f:
                self.latency_table = yaml.safe_load(f)
        else:
            self.latency_table = {}

    def map_param(self, param):
        keys = sorted(param.keys())
        return "({})".format(", ".join("{}={}".format(k, str(param[k])) for k in keys))

    def query(self, config):
        """Latency (ms) of a sub-architecture = sum of its units' table entries (dynamictdnn.py:1336-1370)."""
        depth, channel, kernel, catconv = config
        param = {
            "layer1": {"in_channels": 80, "out_channels": channel[0], "kernel_size": kernel[0]},
            "blocks": [{"channels": channel[0], "mid_channels": channel[1 + i], "kernel_size": kernel[1 + i]}
                       for i in range(depth)],
            "catconv": {"in_channels": channel[0] * depth, "out_channels": catconv},
            "pool": {"in_dim": catconv},
            "embed": {"in_features": catconv * 2, "out_features": 192},
        }
        total_time = 0
        for key, value in param.items():
            if not isinstance(value, list):
                value = [value]
            total_time += sum(self.latency_table[key][self.map_param(v)] for v in value)
        return total_time

    @staticmethod
    def _unit_subnet(unit, p):
        """Smallest subnet config that contains `unit` with the parameter set p."""
        if unit == "layer1":
            return (1, [p["out_channels"], 128], [p["kernel_size"], 1], 384)
        if unit == "blocks":
            return (1, [p["channels"], p["mid_channels"]], [1, p["kernel_size"]], 384)
        if unit == "catconv":        # K of the catconv = depth * channels: one block of that many channels
            return (1, [p["in_channels"], 128], [1, 1], p["out_channels"])
        if unit == "pool":
            return (1, [128, 128], [1, 1], p["in_dim"])
        if unit == "embed":
            if p["in_features"] % 2:
                raise ValueError("embed.in_features is 2 x the catconv width")
            return (1, [128, 128], [1, 1], p["in_features"] // 2)
        raise KeyError(unit)

    def measure(self, model_class=None, device="cuda:0", exclude=("dilation",), batch=64, frames=300,
                precision="bf16", warmup_steps=3, measure_steps=10, verbose=False):
        """
        Fill the table on this GPU (dynamictdnn.py:1372-1430).  `model_class`: {unit: {"kwargs": {name: [values]}}} for
        the five units (the reference's "class" / "input" entries are accepted and ignored: batch and frames are
        arguments here, default 64 x 300 as in the reference's CUDA measurement); default = the coarse search space.
        Every entry is the mean over `measure_steps` forwards of the hipEvent-bracketed stages of that unit.
        """
        model_class = COARSE_GRID if model_class is None else model_class
        assert sorted(model_class.keys()) == sorted(["layer1", "blocks", "catconv", "pool", "embed"])
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.SugarError("sugar_amd runs on an MI355X only: pass device='cuda:N' (there is no CPU fallback)")
        x = torch.rand(batch, 80, frames, device=dev)
        names = {_lib.lib().sugar_profile_stage_name(i).decode(): i for i in range(_lib.STAGE_COUNT)}
        self.latency_table = {}
        for unit, value in model_class.items():
            self.latency_table[unit] = {}
            grid = value["kwargs"]
            keys = list(grid.keys())
            mask = 0
            for st in _UNIT_STAGES[unit]:
                mask |= 1 << names[st]
            for combo in itertools.product(*[grid[k] for k in keys]):
                p = dict(zip(keys, combo))
                sub = TDNNSubnet(self._unit_subnet(unit, p), precision=precision).to(dev)
                with torch.no_grad():
                    for _ in range(warmup_steps):
                        sub(x)
                    _lib.profile_enable(mask)
                    for _ in range(measure_steps):
                        sub(x)
                    torch.cuda.synchronize(dev)
                    ms = sum(t for _, t in _lib.profile_read().values()) / measure_steps
                    _lib.profile_enable(0)
                sub.release()
                key = self.map_param({k: v for k, v in p.items() if k not in exclude})
                self.latency_table[unit][key] = float(ms)
                if verbose:
                    print("%s %s: %.4f ms" % (unit, key, ms), flush=True)
        return self.latency_table

    def to_yaml(self, yaml_file):
        if os.path.exists(yaml_file):
            raise FileExistsError(yaml_file)
        with open(yaml_file, "w") as f:
            yaml.dump(self.latency_table, f)
