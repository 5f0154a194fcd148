#!/usr/bin/env python3
"""
bench.py -- speaker embeddings/sec (3 s @ 16 kHz) of the MI355X-native EfficientTDNN extraction path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload base256|largest256|wide256] [--precision bf16|fp32]
    python bench.py --workload largest --global-batch 512 [--gpus N]      # BASELINE configs[2] as written: strong scaling

One step = one pass of the hot path (waveforms already resident in HBM -> fused fbank -> TDNN -> L2-normalised
embeddings) over one batch of 256 synthetic 3 s utterances per GPU.  N > 1: launched by torch.distributed.run, one
rank per GPU, utterance batches sharded across ranks (weak scaling, no data-path collective inside the network);
the only exchange step is the all-gather of the [256, 192] embeddings of each step (RCCL over xGMI), issued
asynchronously and completed inside the timed region.  --global-batch G splits ONE batch of G utterances evenly over the
ranks instead (G / N per GPU, "scaling": "strong"): BASELINE configs[2] is the Largest subnet at a global batch of 512
on 1 / 2 / 4 / 8 GPUs; on one GPU `--workload largest --global-batch 512|256|128|64` measures the per-GPU share of each.

Rank 0 prints ONE JSON line (see the driver contract); it carries
  roofline     : the dominant kernel = the catconv GEMM launch (49 % of the MACs): algorithmic FLOPs per
                 launch / its average duration, measured live with hipEvents on the launch stream over the timed steps
  cpu_baseline : the oracle (oracle/ref_cpu.py, a CPU restatement of the reference path) timed on the host cores
                 on a bounded sample (8 x 3 s batches for ~10 s), rank 0 at N = 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (config, per-GPU batch, samples, supernet)   BASELINE.json configs[1] / configs[2]; wide256 = the full
    # 1024-channel tdnn16m4g (sugar/models/dynamictdnn.py:942-944), the "1024-channel convs" of north_star
    "base256": ("BASE", 256, 48000, "tdnn8m2g"),
    "largest256": ("LARGEST", 256, 48000, "tdnn8m2g"),
    "base64": ("BASE", 64, 48000, "tdnn8m2g"),       # diagnostic: every activation tensor fits the 256 MB Infinity Cache
    "base128": ("BASE", 128, 48000, "tdnn8m2g"),     # diagnostic: 3.5 rounds of catconv tiles
    "base1": ("BASE", 1, 48000, "tdnn8m2g"),         # diagnostic: one utterance, the latency floor of ~35 dependent kernels
    "wide256": ((3, [1024] * 4, [5] * 4, 1536), 256, 48000, "tdnn16m4g"),
    "largest": ("LARGEST", 512, 48000, "tdnn8m2g"),   # configs[2]: the batch is GLOBAL (--global-batch, default 512)
    "base": ("BASE", 256, 48000, "tdnn8m2g"),         # with --global-batch: Base at a global batch split over the ranks
}
SUPERNETS = {"tdnn8m2g": (80, 192, 4, 512, 5, 1536), "tdnn16m4g": (80, 192, 3, 1024, 5, 1536)}   # supernet_shapes arguments
PREHEAT_STEPS = 30   # untimed passes at set-up, before the W warm-up steps: the clocks of an idle MI355X take ~25 ms of
                     # work to ramp, longer than W = 5 steps (measured: 1.351 ms/step timed after 5 steps from idle,
                     # 1.324 after 60, 1.328 over 100 timed steps); reported in config.preheat_steps
PEAK = {"bf16": (2500.0, "dense bf16 MFMA peak, MI355X_MICROARCH.md chip table"),
        "fp32": (157.3, "f32 matrix peak, MI355X_MICROARCH.md chip table")}


def cpu_baseline(config, shapes, seconds=10.0):
    """Oracle on the host cores: embeddings/s of 8 x 3 s batches (bounded sample of the same workload)."""
    from oracle import ref_cpu
    from sugar_amd import layout, synth
    cores = min(os.cpu_count() or 1, 16)           # the GPU box gives one job a 16-CPU share
    torch.set_num_threads(cores)
    sd = synth.fill_state_dict(layout.supernet_shapes(*shapes), seed=0)
    osd = ref_cpu.clone_state(sd, config, num_blocks=shapes[2], channels=shapes[3])
    wav = synth.waveforms(8, 48000, seed=100)
    for _ in range(2):
        ref_cpu.embed(osd, config[0], wav, norm=True)
    n, t0 = 0, time.perf_counter()
    while n < 3 or time.perf_counter() - t0 < seconds:
        ref_cpu.embed(osd, config[0], wav, norm=True)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(8 * n / dt, 2), "unit": "embeddings/s", "cores": cores, "kind": "port",
            "sample": "%d forwards of 8 x 3 s (48000-sample) utterances, f32 oracle (torch CPU), %.1f s" % (n, dt)}


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def stage_roofline(stages_ms, config, B, L, R, precision):
    """Per stage: algorithmic GFLOP and MB per step, the roofline that bounds it and the fraction
    of that roofline the measured time reaches.  Bytes count each activation tensor once in and once out."""
    depth, ch, ks, ccat = config
    esz = 2 if precision == "bf16" else 4
    c0 = ch[0]
    w = [ch[i + 1] // 8 for i in range(depth)]
    mid = [8 * x for x in w]
    work = {   # stage: (flops, bytes)
        "fbank_dft+instnorm": (0.0, 4.0 * B * L + esz * 80.0 * R),
        "layer1": (2.0 * R * c0 * 80 * ks[0], (80.0 + c0) * R * esz),
        "first_conv": (sum(2.0 * R * c0 * m for m in mid), sum((c0 + m) * R * esz for m in mid)),
        "res2_steps": (sum(7 * 2.0 * R * x * x * ks[i + 1] for i, x in enumerate(w)), sum(2.0 * 7 * x * R * esz for x in w)),
        "last_conv": (sum(2.0 * R * m * c0 for m in mid), sum((m + c0) * R * esz for m in mid)),
        "se_scale": (0.0, depth * c0 * R * esz * 1.0),
        "se_apply": (0.0, (4.0 * (depth - 1) + 3.0) * c0 * R * esz),
        "catconv": (2.0 * R * depth * c0 * ccat, (depth * c0 + ccat) * R * esz * 1.0),
        "pool_linear1": (2.0 * R * ccat * 128, (ccat + 128.0) * R * esz),
        "pool_stats": (2.0 * R * 128 * ccat, (ccat + 128.0) * R * esz),
    }
    peak_tf = PEAK[precision][0]
    out = {}
    for name, (flops, nbytes) in work.items():
        ms = stages_ms.get(name)
        if not ms:
            continue
        t_mfma, t_hbm = flops / (peak_tf * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
        bound = "mfma" if t_mfma > t_hbm else "hbm"
        out[name] = {"ms": ms, "gflop": round(flops / 1e9, 2), "mbytes": round(nbytes / 1e6, 1), "bound": bound,
                     "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "gbytes_per_s": round(nbytes / (ms * 1e-3) / 1e9, 1),
                     "frac": round(max(t_mfma, t_hbm) / (ms * 1e-3), 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="base256", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: ONE batch of this many utterances split evenly over the ranks (per-GPU batch = "
                         "G / N); default for --workload largest / base: 512 / 256.  Without it the per-GPU batch is "
                         "the workload's and the global batch grows with N (weak scaling)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the timed steps alternate over (each with its own workspace).  1 (default) keeps "
                         "the per-kernel roofline clean; the 2-stream rate (consecutive batches overlapping on the GPU) "
                         "is measured after the timed region and reported as 'pipelined'")
    ap.add_argument("--preheat", type=int, default=PREHEAT_STEPS,
                    help="untimed set-up passes before the warm-up steps (clock ramp); 0 for counter-collection runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the 2-stream leg after the timed region (rocprofv3 runs: its overlapped kernels would "
                         "be averaged into the per-kernel durations)")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage hipEvent breakdown to stderr")
    args = ap.parse_args()

    import torch.distributed as dist

    import sugar_amd
    from sugar_amd import _lib, layout, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    # SUGAR_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share the cards,
    # the embeddings are exchanged through gloo); the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SUGAR_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local >= n_dev:
        sys.exit("bench.py: rank %d has no GPU of its own (%d visible)" % (rank, n_dev))
    local = local % max(1, n_dev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg, B, L, family = WORKLOADS[args.workload]
    strong = args.global_batch is not None or args.workload in ("largest", "base")
    if strong:
        G = args.global_batch if args.global_batch is not None else B
        if G % world or G < world:
            sys.exit("bench.py: --global-batch %d does not split evenly over %d ranks" % (G, world))
        B = G // world
    config = getattr(layout, cfg) if isinstance(cfg, str) else cfg
    sup = getattr(sugar_amd, family)()
    sup.load_state_dict(synth.fill_state_dict(layout.supernet_shapes(*SUPERNETS[family]), seed=0))
    sub = sup.clone(config).set_precision(args.precision)
    model = sugar_amd.WrappedModel(sugar_amd.SpeakerModel(sub, transform=sugar_amd.LogMelFbanks())).to(dev).eval()

    # synthetic N(0,1) waveforms, resident in HBM before the timed region; each rank gets its own utterances
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    wav = torch.randn(B, L, device=dev, generator=g, dtype=torch.float32)
    T = 1 + L // 160
    R = B * T

    gathered = [torch.empty(world * B, 192, device=dev) for _ in range(2)] if world > 1 else None

    # rehearsal backend (gloo): the exchange is staged through pinned host buffers by hand -- torch's gloo path for CUDA
    # tensors synchronises and stages on streams of its own, and with steps alternating over two side streams that
    # cost 4.6 s per step in round 2's rehearsal (profiles/bench_r02_2rank_gloo_rehearsal.json, "pipelined" leg)
    host_in = [torch.empty(B, 192).pin_memory() for _ in range(2)] if world > 1 and backend != "nccl" else None
    host_out = [torch.empty(world * B, 192).pin_memory() for _ in range(2)] if world > 1 and backend != "nccl" else None

    class _Done:
        def wait(self):
            return True

    def all_gather(i, emb):
        if backend == "nccl":
            return dist.all_gather_into_tensor(gathered[i & 1], emb, async_op=True)
        host_in[i & 1].copy_(emb, non_blocking=True)
        torch.cuda.current_stream().synchronize()          # gloo moves host memory
        dist.all_gather_into_tensor(host_out[i & 1], host_in[i & 1])
        gathered[i & 1].copy_(host_out[i & 1], non_blocking=True)
        return _Done()

    lanes = [torch.cuda.Stream(dev) for _ in range(max(1, args.streams))]

    def step(i):
        # one step = one full pass over one batch of B utterances; step i runs on stream i % streams
        with torch.cuda.stream(lanes[i % len(lanes)]):
            emb = model(wav, normalize=True)
            if world > 1:
                return all_gather(i, emb)
        return None

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.preheat):                      # set-up: clock ramp (see PREHEAT_STEPS); no collective
        model(wav, normalize=True)
    sync_all()
    for i in range(args.warmup):
        h = step(i)
        if h is not None:
            h.wait()
    _lib.profile_enable(1 << _lib.STAGE_CATCONV)       # 2 hipEvents per step around the dominant kernel
    sync_all()
    t0 = time.perf_counter()
    pending = None
    for i in range(args.steps):
        h = step(i)
        if pending is not None:
            pending.wait()
        pending = h
    if pending is not None:
        pending.wait()
    sync_all()
    dt = time.perf_counter() - t0
    prof = _lib.profile_read()
    _lib.profile_enable(0)
    ranks_seen = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # evidence for the scaling record: which ranks took part, on which cards, what the last all-gather delivered
        # (row 0 of every rank's block of the last step's embeddings must be a unit vector from that rank)
        last = gathered[(args.steps - 1) & 1].view(world, B, 192)[:, 0].norm(dim=1)
        mine = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(dev),
                "pci_bus": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None),
                "gathered_row_norms": [round(float(x), 4) for x in last.cpu()]}
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, mine)

    # after the timed region: the same K steps alternating over two streams (HBM-bound kernels of one batch under
    # the MFMA-bound kernels of the next); reported beside `value`, never as `value`
    pipelined = None
    if len(lanes) == 1 and not args.no_pipelined:
        lanes2 = [torch.cuda.Stream(dev) for _ in range(2)]

        def step2(i):
            with torch.cuda.stream(lanes2[i & 1]):
                emb = model(wav, normalize=True)
                if world > 1:
                    return all_gather(i, emb)
            return None

        for i in range(2):
            h = step2(i)
            if h is not None:
                h.wait()
        sync_all()
        t1 = time.perf_counter()
        pending = None
        for i in range(args.steps):
            h = step2(i)
            if pending is not None:
                pending.wait()
            pending = h
        if pending is not None:
            pending.wait()
        sync_all()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        pipelined = {"streams": 2, "value": round(world * B * args.steps / dt2, 1), "unit": "embeddings/s",
                     "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                     "note": "same steps, consecutive batches alternating over 2 HIP streams; measured after the "
                             "timed region"}

    # after the timed region, rank 0: hipEvent brackets around every stage of 5 more steps, and each stage's algorithmic
    # FLOPs / bytes against the MFMA and HBM peaks (north_star: "MFMA utilisation for the conv GEMMs and achieved HBM
    # GB/s for the feature and pooling kernels, each against MI355X peak")
    stages = stage_roof = None
    if rank == 0:
        _lib.profile_enable((1 << _lib.STAGE_COUNT) - 1)
        for i in range(5):
            model(wav, normalize=True)
        torch.cuda.synchronize(dev)
        stages = {k: round(ms / 5, 4) for k, (n, ms) in _lib.profile_read().items()}
        _lib.profile_enable(0)
        stage_roof = stage_roofline(stages, config, B, L, R, args.precision)
        if args.stages:
            print("per-stage ms/step (hipEvents, 5 steps): " + json.dumps(stages), file=sys.stderr)

    if rank == 0:
        n_cat, ms_cat = prof.get("catconv", (0, 0.0))
        depth, ch, _, ccat = config
        flops_launch = 2.0 * (depth * ch[0]) * ccat * R               # algorithmic: 2 * K * N * frames
        peak, peak_src = PEAK[args.precision]
        achieved = flops_launch / (ms_cat / n_cat * 1e-3) / 1e12 if n_cat else None
        # HBM bytes per launch come from a separate rocprofv3 --pmc job (counters cannot be read from inside this
        # process; tools/pmc_table.py pivots their CSVs): profiles/traffic_catconv.json; traffic_source says which run that was
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic_catconv.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("%s_%s" % (args.workload, args.precision))
            traffic_src = "offline: rocprofv3 --pmc passes of this command, %s (profiles/traffic_catconv.json)" \
                          % tj.get("_source", "round 1")
        out = {
            "metric": "speaker embeddings/sec (3 s @16 kHz)",
            "value": round(world * B * args.steps / dt, 1),
            "unit": "embeddings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s: EfficientTDNN subnet %s, %d x 3 s (48000-sample) utterances per GPU, "
                                   "waveform -> fbank -> TDNN -> L2-normalised 192-d embedding"
                                   % (args.workload, layout.config_name(config), B),
                       "per_gpu_batch": B, "global_batch": world * B, "frames_per_utt": T,
                       "parallelism": "utterance-sharded x%d, all-gather of embeddings (%s)" % (world, backend),
                       "world_size": world, "collective_backend": backend if world > 1 else None,
                       "streams": len(lanes), "preheat_steps": args.preheat,
                       "weights": "deterministic synthetic (sugar_amd/synth.py)"},
            "roofline": {"kernel": "%s catconv launch [%d frames x K %d -> %d]"
                                   % ("gemm1x1_persist_kernel (256x256x64 LDS-DMA MFMA GEMM, quadrant ping-pong schedule, persistent split-tail grid)" if args.precision == "bf16"
                                      else "conv_gemm_kernel<float,128,128>", R, depth * ch[0], ccat),
                         "bound": "mfma", "achieved": round(achieved, 2) if achieved else None, "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                         "launches_timed": n_cat,
                         "avg_launch_ms": round(ms_cat / n_cat, 4) if n_cat else None,
                         "flops_per_launch": flops_launch, "peak_source": peak_src},
        }
        if stages:
            out["stages_ms"] = stages
            out["stage_roofline"] = stage_roof
        if pipelined:
            out["pipelined"] = pipelined
        if ranks_seen:
            out["ranks"] = ranks_seen
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(config, SUPERNETS[family])
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
