"""
TDNNLatency (sugar/models/dynamictdnn.py:1281-1441).  `query` against the reference's own class
on the reference's own V100 table (tests/golden/latency.json holds the entries the queries touch and the results);
`measure` on the GPU: a small grid, then the table's additive estimate against a directly timed forward.
"""
import json
import os

import pytest
import torch
import yaml

import sugar_amd
from sugar_amd import _lib, layout
from sugar_amd.latency import COARSE_GRID, TDNNLatency, latency

with open(os.path.join(os.path.dirname(__file__), "golden", "latency.json")) as f:
    GOLD = json.load(f)


def test_query_matches_reference(tmp_path):
    path = tmp_path / "table.yml"
    with open(path, "w") as f:
        yaml.dump(GOLD["table_subset"], f)
    lat = TDNNLatency(str(path))
    for cfg, want in zip(GOLD["configs"], GOLD["query_ms"]):
        assert abs(lat.query((cfg[0], cfg[1], cfg[2], cfg[3])) - want) < 1e-9
    assert lat.map_param({"out_channels": 512, "in_channels": 80, "kernel_size": 5}) == \
        "(in_channels=80, kernel_size=5, out_channels=512)"
    with pytest.raises(FileExistsError):
        TDNNLatency(str(tmp_path / "missing.yml"))
    with pytest.raises(FileExistsError):
        lat.to_yaml(str(path))
    with pytest.raises(KeyError):
        lat.query((3, [520, 512, 512, 512], [5, 3, 3, 3], 1536))
    # the default grid covers every coarse-space sub-architecture
    for c in COARSE_GRID["catconv"]["kwargs"]["in_channels"]:
        assert c % 8 == 0


@pytest.mark.gpu
def test_measure_and_query_on_gpu(tmp_path):
    base = layout.BASE
    grid = {
        "layer1": {"kwargs": {"in_channels": [80], "out_channels": [512], "kernel_size": [5]}},
        "blocks": {"kwargs": {"channels": [512], "mid_channels": [512], "kernel_size": [3], "dilation": [2]}},
        "catconv": {"kwargs": {"in_channels": [1536], "out_channels": [1536]}},
        "pool": {"kwargs": {"in_dim": [1536]}},
        "embed": {"kwargs": {"in_features": [3072], "out_features": [192]}},
    }
    lat = TDNNLatency()
    table = lat.measure(grid, device="cuda:0", batch=64, frames=300)
    assert set(table) == set(grid)
    assert list(table["blocks"]) == ["(channels=512, kernel_size=3, mid_channels=512)"]      # dilation excluded
    est = lat.query(base)
    assert est > 0
    sub = sugar_amd.TDNNSubnet(base, precision="bf16")
    direct = latency(sub, [64, 80, 300], device="cuda:0", warmup_steps=5, measure_steps=20)
    # additive table (no cat / bnpool / bnembed, as in the reference) vs the wall clock of the whole forward,
    # which also pays the host-side transpose of the [B, 80, T] input: same order of magnitude, table below wall
    assert 0.3 * direct < est < 1.2 * direct, (est, direct)
    out = tmp_path / "mi355x.yml"
    lat.to_yaml(str(out))
    assert TDNNLatency(str(out)).query(base) == est
