"""Headline benchmark: MIPHEI-ViT training tiles/s (256x256 H&E -> 16-channel mIF) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one ModelModule.training_step on one synthetic minibatch per rank: HIP generator forward (H-Optimus-0 ViT-g/14
encoder with LoRA + ViTMatte decoder, bf16 MFMA / f32 accumulate), fused WeightedMSE, HIP backward, gradient
all-reduce (RCCL) when N>1, global-norm clip + Adam.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the instantiation with the
largest share of the step in profiles/r06_kernel_stats_train.txt: the 256x128 MFMA GEMM with the plain store epilogue
-- qkv forward and the dgrad GEMMs, since round 4 the wave-specialised gemm_ws_kernel<STORE>; algorithmic flops /
HIP-event durations recorded live on every fourth launch of ONE of the timed steps), `roofline_step` (whole step, algorithmic FLOPs of SURVEY.md 8d / wall time) and `cpu_baseline` (the CPU oracle =
port of the reference arithmetic, timed on this host's cores on a bounded sample: 2 warm-ups + median of 5).

`--gpus N` from a plain shell (no WORLD_SIZE in the environment) starts the N ranks itself: the parent never touches
the GPU, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, relays rank 0's
JSON line and exits with the child's code.  `--dry --backend gloo` runs launcher + rendezvous + exchange + timing
protocol on CPU tensors (used by tests/test_host_cpu.py; it measures nothing).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOPS_TRAIN = {256: 1647.2e9, 512: 7604.0e9}   # SURVEY.md section 8(d), per tile
FLOPS_FWD = {256: 793.4e9, 512: 3449.1e9}
FLOPS_ENC = {256: 773.6e9}                       # encoder only (patch embed + 40 blocks)
ORION_MARKERS = ["Hoechst", "CD31", "CD45", "CD68", "CD4", "FOXP3", "CD8a", "CD45RO", "CD20", "PD-L1", "CD3e", "CD163",
                 "E-cadherin", "Ki67", "Pan-CK", "SMA"]
PEAK_BF16 = 2.5e15                                # dense MFMA bf16, MI355X_MICROARCH.md
# second, explicitly labelled denominator: what a loop of nothing but v_mfma_f32_16x16x32_bf16 on random register operands sustains at
# the socket power cap (tools/probes/mfma_power.hip, profiles/README.md: 2033.7 TF/s) -- `peak` stays the nominal 2.5 PF
SUSTAINED_BF16 = 2.03e15


from miphei_vit_amd.synthetic import synthetic_batch, synthetic_init_  # noqa: E402,F401  (kept importable as bench.*)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU (cfg.train.batch_size of the reference)")
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--mode", choices=["train", "infer", "embed"], default="train",
                    help="embed = encoder-only class-token embeddings (SURVEY.md 8f row 4, extract_embeddings.py)")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default=None, help="infer / embed modes: fp16 = the reference's evaluation "
                    "convention generator.eval().cuda().half() on the fp16-operand library (default: embed fp16 -- the reference's "
                    "extraction script calls .half() --, infer bf16; training is always bf16: BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2, help="batch of the timed CPU-oracle training step")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU oracle (0: sweep 32 / 64 / 128 / all, report the best)")
    ap.add_argument("--cpu-repeats", type=int, default=5, help="timed repetitions of the CPU sample (median), after 2 warm-ups")
    ap.add_argument("--parity-batch", type=int, default=-1, help="batch of the parity leg (-1: the timed per-GPU batch)")
    ap.add_argument("--encoder", default="hoptimus0")
    ap.add_argument("--generator", choices=["myvitmatte", "unet_lora"], default="myvitmatte",
                    help="unet_lora = the UNETR baseline of the reference (SURVEY.md 8f row 4)")
    ap.add_argument("--metrics", type=int, default=0, help="train mode: also run the per-step PSNR/SSIM state update of the "
                    "reference (models.py:140-143); the headline number is taken with it off (SURVEY.md section 6)")
    ap.add_argument("--graph", type=int, default=1, help="infer mode: replay a hipGraph-captured forward")
    ap.add_argument("--probe", type=int, default=1, help="0: no HIP-event probe of the dominant kernel (no `roofline` object)")
    ap.add_argument("--chunked-conv", type=int, default=1, help="0: wide fusion blocks on the implicit GEMM (A/B of csrc/conv_chunked.hip)")
    ap.add_argument("--bn-fold", type=int, default=1, help="0: eval-mode BatchNorm applied by its own kernels instead of folded into the ConvStream (A/B)")
    ap.add_argument("--attn-residual", type=int, default=1, help="0: attention backward forms D from the bf16 output alone (A/B)")
    ap.add_argument("--lora-group", type=int, default=0, help="ViT blocks per batched LoRA weight-gradient launch (0 = engine default)")
    ap.add_argument("--lora-buckets", type=int, default=4, help="sub-buckets of the LoRA gradient all-reduce (N > 1)")
    ap.add_argument("--decoder-bucket-split", type=int, default=1, help="N > 1: the decoder gradient bucket (18.9 MB) as this many all-reduces")
    ap.add_argument("--nccl-max-nchannels", type=int, default=0, help="N > 1: NCCL_MAX_NCHANNELS for RCCL (0: leave the environment alone); "
                    "fewer channels = fewer CUs held by the ring kernels beside the backward GEMMs")
    ap.add_argument("--nccl-proto", default="", help="N > 1: NCCL_PROTO for RCCL (e.g. Simple, LL, LL128; empty: leave the environment alone)")
    ap.add_argument("--comm-standin", default="0", help="opt-in (e.g. 16,1000) train mode pre-flight, outside the timed region: BLOCKS,USEC "
                    "of the ring-kernel stand-in launched on a side stream at the five bucket-issue points (csrc/standin.hip); 0 = off (default "
                    "since round 6: 39 untimed steps, and the delta it reports includes the tail the last stand-in exposes in finish())")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL; gloo only with --dry")
    ap.add_argument("--dry", action="store_true", help="no GPU work: launcher / rendezvous / exchange / timing protocol on CPU")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch rendezvous port (0: pick a free one)")
    return ap.parse_args(argv)


def comm_overlap_probe(a, mod, eng, sync, step, dev):
    """What do a collective's kernels cost the compute stream in CU residency?    The wave-specialised
    GEMM blocks take 504 of a SIMD's 512 registers and 144 KB of LDS: nothing co-resides on a CU, so a ring kernel can only get a CU
    between GEMM blocks, and while it holds one a one-round GEMM launch (252 tiles on 256 CUs) runs a CU short.  A one-GPU box
    cannot run RCCL with peers, so the probe launches a stand-in (csrc/standin.hip: BLOCKS workgroups of 256 threads, 64 VGPRs,
    held for USEC microseconds, no memory traffic) on a side stream at the five points of the backward pass where
    trainer.DataParallelSync issues its buckets, `finish()` waiting for them as for the all-reduces.  Interleaved A/B outside the
    timed region: 3 x (6 steps without, 6 with), the same hook points (and grouped LoRA launches) in both arms."""
    from miphei_vit_amd.trainer import DataParallelSync
    blocks, usec = (int(v) for v in a.comm_standin.split(","))
    own = sync is None
    s2 = DataParallelSync(eng, hooks_only=True, lora_buckets=a.lora_buckets) if own else sync
    prev = mod.grad_sync
    mod.grad_sync = s2

    def run(n, base):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            step(base + i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    off, on = [], []
    base = a.warmup + a.steps + 8
    try:
        s2.standin = None
        run(3, base)
        for r in range(3):
            s2.standin = None
            off.append(run(6, base + 12 * r))
            s2.standin = (blocks, usec)
            on.append(run(6, base + 12 * r + 6))
    finally:
        s2.standin = None
        mod.grad_sync = prev
    m_off, m_on = sum(off) / len(off), sum(on) / len(on)
    return {"standin_blocks": blocks, "standin_us": usec, "launches_per_step": 1 + a.lora_buckets,
            "ms_per_step_without": round(m_off, 3), "ms_per_step_with": round(m_on, 3),
            "delta_pct": round(100.0 * (m_on - m_off) / m_off, 2), "rounds_ms": [[round(x, 3), round(y, 3)] for x, y in zip(off, on)],
            "note": "stand-in for RCCL ring kernels on a side stream at the bucket-issue points of the backward pass; "
                    "interleaved A/B outside the timed region; no real peer exchange is part of this number"}


def self_launch(a, argv):
    """Plain-shell `bench.py --gpus N`: start the ranks as a child `torch.distributed.run` (never exec: the parent stays a
    GPU-free supervisor), pass rank 0's stdout through, return the child's exit code."""
    port = a.master_port
    if not port:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required for RCCL across processes on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    out, _ = proc.communicate()
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in out.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode else (0 if lines else 1)


def dry_run(a, world, rank):
    """CPU rehearsal of the multi-rank protocol: rendezvous, broadcast, bucketed exchange with the backward hooks, barriers,
    MAX-over-ranks timing, one JSON line from rank 0."""
    import types
    from miphei_vit_amd.trainer import DataParallelSync
    if world > 1:
        dist.init_process_group(a.backend)
    L, per, ndec = 8, 64, 500

    class Eng:
        def __init__(self):
            g = torch.Generator().manual_seed(100 + rank)
            self._flat = types.SimpleNamespace(flat=torch.full((L * per + ndec,), float(rank)),
                                               gflat=torch.randn(L * per + ndec, generator=g), n_lora=L * per)
            self._pack_key = "x"

        def params_changed(self):
            self._pack_key = None

        def _ensure_flat(self):
            return self._flat

        def grad_buckets(self):
            return self._flat.gflat[L * per:], self._flat.gflat[:L * per]

        def lora_blocks(self):
            return L

    eng = Eng()
    sync = DataParallelSync(eng, lora_buckets=a.lora_buckets)
    sync.broadcast_parameters(0)
    ref = torch.stack([torch.randn(L * per + ndec, generator=torch.Generator().manual_seed(100 + r))
                       for r in range(world)]).mean(0)

    def step():
        sync.decoder_ready()
        for l in range(L - 1, -1, -1):
            sync.lora_block_done(l)
        sync.finish()

    g0 = eng._flat.gflat.clone()
    step()
    ok = bool(torch.allclose(eng._flat.gflat, ref, atol=1e-6)) and bool((eng._flat.flat == 0).all())
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng._flat.gflat.copy_(g0)
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        ranks = dist.get_world_size()
        dist.destroy_process_group()
    else:
        ranks = 1
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": round(world * a.batch * a.steps / dt, 3), "unit": "tiles/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "dry": True, "exchange_ok": ok,
                          "rccl_ranks": ranks, "backend": a.backend, "scaling": "weak"}), flush=True)
    return 0 if ok else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.dtype is None:
        a.dtype = "fp16" if a.mode == "embed" else "bf16"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a, argv)             # before anything touches the GPU
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry:
        return dry_run(a, world, rank)
    if a.backend != "nccl":
        raise SystemExit("--backend gloo is only meaningful with --dry: the product path is RCCL")

    # Everything written to stdout while the job runs (RCCL's version banner comes through C stdio, library chatter) is sent
    # to stderr; the descriptor is restored for the single JSON line at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_ddp = os.environ.get("MIPHEI_FORCE_DDP", "0") == "1"  # run the RCCL exchange even on one rank (testing)
    if world > 1 or force_ddp:
        # RCCL reads these at communicator creation: set them before init (every rank parses the same command line)
        if a.nccl_max_nchannels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = str(a.nccl_max_nchannels)
        if a.nccl_proto:
            os.environ["NCCL_PROTO"] = a.nccl_proto
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from miphei_vit_amd import ops
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss, marker_weights_from_file
    from miphei_vit_amd.models import ModelModule
    from miphei_vit_amd.trainer import DataParallelSync

    nc = 16
    weights = marker_weights_from_file(os.path.join(ROOT, "configs", "channel_stats_orion.json"), ORION_MARKERS)
    if a.mode == "embed":
        from miphei_vit_amd.generators.foundation_models import FOUNDATION_MODEL_REGISTRY
        with torch.device(dev):
            emb_model = FOUNDATION_MODEL_REGISTRY[a.encoder](a.img, pretrained=False, global_pool="token")
        synthetic_init_(emb_model, seed=0)
        emb_model = emb_model.eval()
        if a.dtype == "fp16":
            emb_model = emb_model.half()
    unet = a.generator == "unet_lora"
    if unet and a.mode == "embed":
        raise SystemExit("--generator unet_lora: train / infer modes only")
    with torch.device(dev):
        if unet:
            from miphei_vit_amd.generators.unet import Unet
            model = Unet(a.img, a.encoder, use_lora=True, classes=nc, pretrained=False)
        else:
            model = get_vitmatte("tiny" if a.mode == "embed" else a.encoder, a.img, nc, use_lora=True, pretrained=False)
    synthetic_init_(model, seed=0)
    eng = model._engine
    if hasattr(eng, "use_chunked_conv"):
        eng.use_chunked_conv, eng.attn_residual = bool(a.chunked_conv), bool(a.attn_residual)
        eng.bn_fold = bool(a.bn_fold)
    if a.lora_group > 0:
        (eng._encoder_engine() if hasattr(eng, "_encoder_engine") else eng).lora_group = a.lora_group
    if unet and not hasattr(eng, "capture_inference"):
        a.graph = 0
    mod = ModelModule(model, None, 2e-4 * a.batch ** 0.5, 0., WeightedMSELoss(50.0, weights)).to(dev)
    mod.total_iters = 100000
    mod.update_pix_metrics = bool(a.metrics)
    sync = None
    if world > 1 or force_ddp:
        sync = DataParallelSync(eng, force=force_ddp, lora_buckets=a.lora_buckets, timing=True, decoder_split=a.decoder_bucket_split)
        sync.broadcast_parameters(0)
        mod.grad_sync = sync
    batches = [synthetic_batch(1234 + rank * 1000 + i, a.batch, a.img, nc, dev) for i in range(4)]

    if a.dtype == "fp16" and a.mode == "train":
        raise SystemExit("--dtype fp16: infer / embed modes only (training needs fp32 master parameters and bf16 operands)")
    x16 = [b[0].half() for b in batches] if a.mode == "embed" else None
    if a.mode == "infer":
        model.eval()
        if a.dtype == "fp16":
            model.half()
        run_graph, x_static, _ = eng.capture_inference(a.batch) if a.graph else (None, None, None)

    def step(i):
        x, y = batches[i % len(batches)]
        if a.mode == "train":
            mod.training_step({"image": x, "target": y}, i)
        elif a.mode == "embed":
            with torch.no_grad():
                emb_model(x16[i % len(batches)])
        elif a.graph:
            x_static.copy_(x)      # device-to-device; the tile batch is already resident
            run_graph()
        elif unet:
            with torch.no_grad():
                model(x)
        else:
            eng.forward(x, train=False, bn_train=False)

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if sync is not None:
        sync.exposed_events = []
        sync.bucket_events = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # roofline leg: HIP events around the launches of the dominant kernel, on a SAMPLE of the timed steps (every probe_every-th,
    # at least one): an event pair costs the stream ~6 us of idle time per launch, which on all 159 launches of every step was
    # 2.9 % of the headline number (395 vs 406.5 tiles/s same box)
    # Round 4: the probe samples ONE of the timed steps (the middle one) and every fourth of its 159 launches of the dominant kernel
    # (qkv forward x 40, then the dfc1 / dproj / dqkv input gradients in turn: a stride of 4 keeps the four shapes' proportions), and one event per
    # step boundary (a marker every ~35 ms) gives the duration of every step, so the line also carries the rate of the unprobed
    # steps (`unprobed`).  Measured per step (`unprobed.step_ms`): a probed step in the middle of the run costs +1.0 ms, the FIRST
    # timed step +2.6 ms when probed (the host is not yet ahead of the device there and pays 318 event creations on the critical
    # path): five sampled steps of 20 cost the headline ~0.7 %, steps {0, K/2} 0.5 %, the middle step alone 0.15 %.
    ops.PROBE.start()
    ops.PROBE.stride = 4      # every fourth launch of the sampled step (40 of 159: the four shapes in their proportions): the probed step
    ops.PROBE.on = False      # then costs ~0.25 ms instead of ~1 ms
    probed = [a.steps // 2] if a.probe else []
    probe_every = max(1, a.steps // 2)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host_ms = []
    for i in range(a.steps):
        ops.PROBE.on = i in probed
        th = time.perf_counter()
        step(a.warmup + i)
        marks[i + 1].record()
        host_ms.append((time.perf_counter() - th) * 1e3)      # host time to enqueue the step (the device runs behind)
    ops.PROBE.on = True
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    probe = ops.PROBE.stop()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    free_ms = [t for i, t in enumerate(step_ms) if i not in probed] or step_ms
    comm = (sync.exposed_ms(), sync.bucket_report()) if (sync is not None and a.mode == "train") else None   # timed steps only
    kernels = None
    if a.probe and a.mode == "train":       # (every rank: the extra steps run the gradient exchange too)
        # per-kernel roofline leg, OUTSIDE the timed region: two extra steps with an event pair around every dense GEMM call
        # (per tile variant + epilogue) and every attention call, so the weakest kernels are on the line, not only the dominant one
        ops.KPROBE.start()
        for i in range(2):
            step(a.warmup + a.steps + i)
        kernels = ops.KPROBE.stop()
    if a.probe and a.mode == "infer" and not unet:
        # configs[4]: the same per-kernel leg on two extra EAGER forwards after the timed region (the captured graph cannot carry
        # events): every dense GEMM and attention call of the inference forward with its algorithmic FLOPs
        ops.KPROBE.start()
        for i in range(2):
            eng.forward(batches[i][0], train=False, bn_train=False)
        kernels = ops.KPROBE.stop()
    overlap_probe = None
    if a.mode == "train" and a.comm_standin not in ("0", "") and hasattr(eng, "grad_buckets"):
        overlap_probe = comm_overlap_probe(a, mod, eng, sync, step, dev)
    if world > 1:
        dist.barrier()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    tiles = world * a.batch * a.steps
    value = tiles / dt
    flops_tile = None if unet else (FLOPS_TRAIN if a.mode == "train" else FLOPS_ENC if a.mode == "embed" else FLOPS_FWD).get(a.img)

    cfg_idx = (1 if a.img == 256 else 3) if a.mode == "train" else 4
    if a.mode == "train" and a.img == 256 and world > 1:
        cfg_idx = 2
    res = {
        "metric": (f"training tiles/sec ({a.img}x{a.img} H&E->16ch mIF)" if a.mode == "train" else
                   "embedding tiles/sec (encoder only, class token, fp16 in/out)" if a.mode == "embed" else "inference tiles/sec"),
        "value": round(value, 3), "unit": "tiles/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype if a.mode != "train" else "bf16", "data": "synthetic",
        "config": {"workload": (f"UNETR baseline ({a.encoder} + LoRA r8, ViT pyramid + up-conv decoder, 16 heads) " if unet else
                                f"MIPHEI-ViT ({a.encoder} ViT-g/14 + LoRA r8, ViTMatte decoder, 16 heads) ") + f"{a.mode} step, "
                               f"{a.img}x{a.img} tiles, per-GPU batch {a.batch} " +
                               ("(SURVEY.md 8f row 4)" if a.mode == "embed" or unet else f"(BASELINE.json configs[{cfg_idx}])"),
                   "global_batch": world * a.batch, "img": a.img, "parallelism": f"dp{world}", "pix_metrics": bool(a.metrics)},
    }
    res["unprobed"] = {"ms_per_step": round(sum(free_ms) / len(free_ms), 3), "tiles_per_s": round(a.batch * world * len(free_ms) / (sum(free_ms) * 1e-3), 2),
                       "steps": len(free_ms), "probed_steps": len(probed), "step_ms": [round(t, 2) for t in step_ms],
                       "host_enqueue_ms": [round(t, 2) for t in host_ms],
                       "note": "this rank's timed steps that carried no HIP-event probe (step-boundary events on the compute stream); "
                               "`value` is the contract number over ALL timed steps, probe overhead included"}
    if a.mode == "train":
        res["comm_overlap_probe"] = overlap_probe
        res["multi_gpu_measured"] = bool(world > 1)     # (N = 1 lines: no N > 1 throughput of this build exists in this run)
    if dist.is_initialized():
        res["rccl_ranks"] = dist.get_world_size()
        if comm is not None:
            res["exposed_comm_ms_per_step"] = None if comm[0] is None else round(comm[0], 4)
            res["comm_buckets"] = comm[1]      # per bucket: bytes, slack behind the backward pass, stall in finish()
            res["config"]["lora_buckets"] = a.lora_buckets
            res["config"]["decoder_bucket_split"] = a.decoder_bucket_split
            res["rccl_env"] = {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_PROTO", "NCCL_ALGO", "NCCL_NCHANNELS_PER_PEER")
                               if k in os.environ}
    if a.mode == "infer":
        # p50 latency of one batch, measured after the throughput window with a sync per batch
        ts = []
        for i in range(min(a.steps, 20)):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(i)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        ts.sort()
        res["p50_batch_latency_ms"] = round(ts[len(ts) // 2] * 1e3, 3)
        res["config"]["hipgraph"] = bool(a.graph)
    if flops_tile:
        res["model_flops_frac"] = round(value / world * flops_tile / PEAK_BF16, 4)
        ach = value / world * flops_tile
        res["roofline_step"] = {"bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                                "frac": round(ach / PEAK_BF16, 4),
                                "flops_per_tile": flops_tile, "note": "algorithmic FLOPs of SURVEY.md 8(d) / wall time, per GPU"}
    if rank == 0:
        if probe["n"]:
            ach = probe["flops"] / (probe["ms"] * 1e-3)
            dom = ("mvit_gemm::ws::gemm_ws_kernel<STORE> (256x128 tile, 8 MFMA + 4 DMA waves)" if ops.PROBE.ws else
                   "mvit_gemm::gemm_kernel<256,128,4,2,DENSE,STORE> (8 waves)")
            traffic, src = pmc_traffic("gemm_ws_kernel<0>" if ops.PROBE.ws else "gemm_kernel<256, 128, 4, 2, 0, 0>")
            res["roofline"] = {"bound": "mfma", "kernel": dom,
                               "achieved": round(ach / 1e12, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16, 4), "traffic": traffic, "traffic_source": src,
                               "launches": probe["seen"], "sampled_launches": probe["n"],
                               "avg_launch_us": round(probe["ms"] * 1e3 / probe["n"], 2),
                               "sampled_steps": len(probed),
                               "peak_sustained": SUSTAINED_BF16 / 1e12, "frac_of_sustained": round(ach / SUSTAINED_BF16, 4),
                               "note": "HIP events on the stream around every fourth launch of this kernel (`sampled_launches` of `launches`: its four "
                                       "shapes in their proportions) in the sampled timed step"}
        if kernels and not probe["n"]:
            # no in-region probe (inference: the timed region replays a graph): `roofline` = the kernel family with the largest
            # share of the two extra eager forwards
            key, v = max(kernels.items(), key=lambda kv: kv[1]["ms"])
            ach = v["flops"] / (v["ms"] * 1e-3)
            res["roofline"] = {"bound": "mfma", "kernel": key, "achieved": round(ach / 1e12, 2), "peak": PEAK_BF16 / 1e12,
                               "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16, 4), "traffic": None,
                               "launches": v["n"] // 2, "avg_launch_us": round(v["ms"] * 1e3 / v["n"], 2),
                               "note": "dominant dense-GEMM family of the forward: HIP events around each of its calls on two eager "
                                       "forwards after the timed (graph-replay) region; rocprofv3 table of this command: "
                                       "profiles/r06_kernel_stats_infer_b64.txt"}
        if kernels:
            tot_ms = sum(v["ms"] for v in kernels.values())
            rk = []
            traf = pmc_traffic_all()
            for key, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"]):
                ach = v["flops"] / (v["ms"] * 1e-3)
                rk.append({"kernel": key, "traffic": traf.get(_pmc_key(key)), "launches_per_step": v["n"] // 2, "gflop_per_launch": round(v["flops"] / v["n"] / 1e9, 2),
                           "avg_us": round(v["ms"] * 1e3 / v["n"], 1), "achieved": round(ach / 1e12, 1),
                           "frac": round(ach / PEAK_BF16, 4), "ms_per_step": round(v["ms"] / 2, 3)})
            res["roofline_kernels"] = {"bound": "mfma", "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "kernels": rk,
                                       "ms_per_step_total": round(tot_ms / 2, 2),
                                       "note": "HIP events around every dense-GEMM and attention call of two extra steps after "
                                               "the timed region (event pairs add ~6 us of stream idle per call: durations are "
                                               "upper bounds); achieved = algorithmic FLOPs / duration; a ragged-M product "
                                               "(two launches) is timed as one call under its main tile"}
        if not a.no_cpu_baseline and world == 1 and a.mode == "train" and not unet:
            res["cpu_baseline"], res["parity"] = cpu_baseline(model, a, nc, weights, dev, mod)
    # RCCL's banner sits in the C-level stdout buffer until that is flushed (normally at exit, i.e. AFTER anything Python
    # printed): flush it (to stderr, see the top of main) on every rank, tear the group down, then hand stdout back and print
    # the result as the job's only stdout line.
    import ctypes
    libc = ctypes.CDLL(None)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    libc.fflush(None)
    if rank == 0:
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)   # whatever is flushed at interpreter exit stays off stdout
    return 0


def _pmc_key(kprobe_name):
    """name of a KPROBE entry as rocprofv3 spells the kernel: 'mvit_gemm::gemm_kernel<256,128,4,2,DENSE,RESID>' ->
    'gemm_kernel<256, 128, 4, 2, 0, 3>'; the attention entries are spelled as KPROBE spells them (backward = sum of its launches)"""
    epi = {"STORE": 0, "GELU": 1, "SWIGLU": 2, "RESID": 3, "PATCH": 4, "STATS": 5, "DSWIGLU": 6, "DGELU": 7}
    if "gemm_ws_kernel<" in kprobe_name:
        return "ws::gemm_ws_kernel<%d>" % epi.get(kprobe_name.split("<", 1)[1].rstrip(">"), -1)
    if "gemm_kernel<" in kprobe_name:
        f = kprobe_name.split("<", 1)[1].rstrip(">").split(",")
        return "gemm_kernel<%s, %s, %s, %s, 0, %d>" % (f[0], f[1], f[2], f[3], epi.get(f[5], -1))
    return kprobe_name


def pmc_traffic_all():
    """{kernel key: HBM-side bytes per launch} of the newest committed PMC summary (see pmc_traffic)."""
    out = {}
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                k = json.load(f)["kernels"]
        except Exception:  # noqa: BLE001
            continue
        best = {}
        for kn, v in k.items():
            key = kn.replace("void ", "").replace("mvit_gemm::", "").strip()
            out[key] = round(v["hbm_bytes_per_launch_corrected"])
            if "gemm_ws_kernel<" in key:      # 'ws::gemm_ws_kernel<2, true>' (epilogue, band mode) -> alias 'ws::gemm_ws_kernel<2>'
                alias = key.split(",")[0] + ">"
                if v.get("launches", 0) > best.get(alias, -1):
                    best[alias] = v.get("launches", 0)
                    out[alias] = out[key]
        if "attn_bwd_dq_kernel" in out and "attn_bwd_dkv_kernel" in out:
            out["attn_bwd (all launches of mvit_attention_bwd)"] = sum(v for kk, v in out.items() if kk.startswith("attn_bwd_"))
        return out
    return out


PMC_FILES = ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")


def pmc_traffic(kernel="gemm_kernel<256, 128, 4, 2, 0, 0>"):
    """(HBM-side bytes per launch of the dominant kernel, where the figure comes from).  The PMC counters cannot be read from
    inside the timed run: the value is the committed rocprofv3 summary of this same command (separate FETCH_SIZE / WRITE_SIZE
    passes, (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction of MI355X_MICROARCH.md; L2-side fabric requests, so
    Infinity-Cache hits are included), newest round first.  (None, None) when no summary is present."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                doc = json.load(f)
            k = doc["kernels"]
            for kn, v in k.items():
                if kernel in kn or kernel.rstrip(">") + "," in kn:
                    return round(v["hbm_bytes_per_launch_corrected"]), (f"profiles/{name} (static: committed rocprofv3 --pmc passes "
                                                                        f"of the build at commit {doc.get('commit', 'unknown')})")
        except Exception:  # noqa: BLE001
            continue
    return None, None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, warmups, repeats):
    for _ in range(warmups):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baseline(model, a, nc, weights, dev, mod=None):
    """Time the CPU oracle (port of the reference arithmetic, fp32) on a bounded sample of the same workload -- one training step
    (forward + backward) at batch `--cpu-batch`, same weights -- for a sweep of thread counts (one warm-up + one timed step each),
    then 2 warm-ups + median of `--cpu-repeats` at the best count ("all cores" is NOT the fastest setting for the
    torch-CPU GEMMs on a 256-thread host, so the sweep is on the line); the BASELINE configs[0] tiny case the same way; then
    check the HIP path against the oracle at the TIMED batch (train-mode BatchNorm, i.e. the tile path the benchmark ran):
    outputs per channel, loss and the global gradient norm of one training step."""
    from oracle import VIT_CONFIGS
    from oracle.model import OracleTrainer
    ncpu = os.cpu_count() or 1
    cfg = VIT_CONFIGS[a.encoder]
    p = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    B = a.cpu_batch
    x, y = synthetic_batch(999, B, a.img, nc, dev)
    xc, yc = x.cpu(), y.cpu()
    tr = OracleTrainer(p, cfg, nc, batch_size=B, total_iters=100000, weights=weights)
    tr.p = p  # no second copy of the 4.6 GB state
    # Thread sweep on a BOUNDED proxy of the same arithmetic: encoder block 0, forward + backward to its LoRA gradients, at the
    # sample's token count (1/40 of the step's encoder work per run).  The full step is then timed at the best count only:
    # torch-CPU GEMMs collapse far below 256 threads on this host class (measured on the EPYC 9575F box: 0.54 / 0.31 / 0.14 /
    # 0.003 tiles/s for the full step at 32 / 64 / 128 / 256 threads -- the last one ran for 11 minutes), so the sweep stops at
    # the first count that is more than 3x slower than the best one; larger counts are reported as null (= not run).
    from oracle.vit import vit_block
    xb = torch.randn(B, cfg.tokens(a.img), cfg.dim)

    def proxy():
        leaves = {k: p[k].detach().clone().requires_grad_(True) for k in p if ".lora_" in k and k.startswith("encoder.vit.blocks.0.")}
        q = dict(p)
        q.update(leaves)
        h = vit_block(q, "encoder.vit.blocks.0.", xb, cfg, True)
        torch.autograd.grad(h.square().mean(), list(leaves.values()))

    counts = sorted({min(ncpu, c) for c in ((a.cpu_threads,) if a.cpu_threads > 0 else (32, 64, 128, ncpu))})
    sweep, best_t = {str(c): None for c in counts}, None
    for c in counts:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        proxy()                                           # warm-up (also the only run of a collapsed count)
        tw = time.perf_counter() - t0
        if best_t is not None and tw > 3.0 * best_t:
            sweep[str(c)] = round(B / tw, 3)
            break
        _, ts1 = _median_time(proxy, 0, 3)
        sweep[str(c)] = round(B / min(ts1), 3)            # tile-blocks per second (proxy units)
        best_t = min(ts1) if best_t is None else min(best_t, min(ts1))
    cores = int(max((k for k in sweep if sweep[k] is not None), key=lambda k: sweep[k]))
    torch.set_num_threads(cores)
    med, ts = _median_time(lambda: tr.loss_and_grads(xc, yc), 2, a.cpu_repeats)
    base = {"value": round(B / med, 4), "unit": "tiles/s", "cores": cores, "host_cores": ncpu, "cpu_model": _cpu_model(),
            "kind": "port", "threads_sweep": sweep,
            "threads_sweep_unit": "tile-blocks/s of the proxy (encoder block 0 fwd+bwd at the sample's batch; 1 warm-up, best of 3; the sweep "
                                  "stops at the first count > 3x slower than the best: null = not run)",
            "sample": f"training step (fwd+bwd, fp32) of the CPU oracle at batch {B}, same weights, at the thread count the sweep "
                      f"ranks best: 2 warm-ups + median of {len(ts)} = {med:.2f} s (min {min(ts):.2f}, max {max(ts):.2f})"}
    base["tiny"] = cpu_baseline_tiny(cores)
    # parity at the timed batch: one oracle training step (forward, loss, every gradient) against the HIP step on the same inputs
    PB = a.batch if a.parity_batch < 0 else a.parity_batch
    xp, yp = synthetic_batch(998, PB, a.img, nc, dev)
    trp = OracleTrainer(p, cfg, nc, batch_size=PB, total_iters=100000, weights=weights)
    trp.p = p
    out_ref, loss_ref, g_ref = trp.loss_and_grads(xp.cpu(), yp.cpu())
    gn_ref = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g_ref.values())))
    model.train()
    eng = model._engine
    out_d = eng.forward(xp, train=True, bn_train=True)
    out = out_d.float().cpu()
    loss_d, dY = eng.loss_and_grad(out_d, yp, mod.loss_reconstruct.marker_weights.to(dev), mod.loss_reconstruct.lambda_factor)
    gflat = eng.backward(dY)
    gn = float(gflat.double().pow(2).sum().sqrt())
    rel = ((out - out_ref) ** 2).sum(dim=(0, 2, 3)) / (out_ref ** 2).sum(dim=(0, 2, 3))
    xm, ym = out - out.mean(dim=(0, 2, 3), keepdim=True), out_ref - out_ref.mean(dim=(0, 2, 3), keepdim=True)
    pear = (xm * ym).sum(dim=(0, 2, 3)) / ((xm ** 2).sum(dim=(0, 2, 3)).sqrt() * (ym ** 2).sum(dim=(0, 2, 3)).sqrt())
    parity = {"worst_channel_rel_mse": float(rel.max()), "min_pearson_r": float(pear.min()), "tolerance_rel_mse": 1e-3,
              "batch": PB, "loss": float(loss_d), "loss_ref": float(loss_ref),
              "loss_rel_err": abs(float(loss_d) - float(loss_ref)) / abs(float(loss_ref)),
              "grad_norm": gn, "grad_norm_ref": gn_ref, "grad_norm_rel_err": abs(gn - gn_ref) / gn_ref,
              "note": "HIP training step (bf16 MFMA) vs the fp32 CPU oracle on the same inputs and weights: outputs per channel, "
                      "WeightedMSE loss, global norm of all 6.7 M trainable gradients; per-parameter gradients are compared in "
                      "tests/test_full_size_gpu.py"}
    return base, parity


def cpu_baseline_tiny(cores):
    """BASELINE.json configs[0]: tiny ViT (2 layers, 64-d, patch 16) + decoder, 4 x 256x256 tiles -> 3 channels, CPU oracle."""
    from oracle import VIT_CONFIGS, det_state_dict
    from oracle.model import OracleTrainer, generator_state_shapes, orion_marker_weights
    cfg = VIT_CONFIGS["tiny"]
    sd = det_state_dict(generator_state_shapes(cfg, 256, 3), seed=1, layerscale=0.5)
    p = {k: torch.from_numpy(v) if not torch.is_tensor(v) else v for k, v in sd.items()}
    torch.set_num_threads(cores)
    x, y = synthetic_batch(997, 4, 256, 3, "cpu")
    tr = OracleTrainer(p, cfg, 3, batch_size=4, total_iters=100000, weights=orion_marker_weights(3))
    med, ts = _median_time(lambda: tr.loss_and_grads(x, y), 2, 5)
    return {"value": round(4 / med, 3), "unit": "tiles/s", "cores": cores,
            "sample": f"BASELINE configs[0]: tiny ViT + decoder training step, 4 x 256x256 -> 3 ch, median of 5 = {med:.3f} s"}


if __name__ == "__main__":
    sys.exit(main())
