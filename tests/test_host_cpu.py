"""CPU: host logic, C-ABI surface, data-parallel exchange (gloo, world_size 2).  No GPU compute here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every MVIT_API function of include/miphei_hip.h is exported by the built library and bound by ctypes."""
    import __graft_entry__ as g
    from miphei_vit_amd import _lib
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.LIB_PATH_F16)):
        g.build()
    hdr = open(os.path.join(ROOT, "include", "miphei_hip.h")).read()
    declared = set(re.findall(r"MVIT_API\s+(?:int|long long)\s+(mvit_\w+)\s*\(", hdr))
    assert len(declared) >= 30
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for path in (_lib.LIB_PATH, _lib.LIB_PATH_F16):        # the bf16-operand library and its fp16-operand twin export the same ABI
        handle = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(handle, name), (path, name)
    _lib.lib()
    with _lib.operands("f16"):
        assert _lib.lib() is not _lib._libs["bf16"] and _lib.operand_torch_dtype() == torch.float16
    assert _lib.operand_mode() == "bf16" and _lib.operand_torch_dtype() == torch.bfloat16
    # struct layout of mvit_gemm_args matches the C definition (pointers first, then ints, 8-byte aligned)
    body = hdr[hdr.index("typedef struct mvit_gemm_args"):hdr.index("} mvit_gemm_args;")]
    n_ptr = len(re.findall(r"\*\s*\w+\s*[;,]", body))
    n_int = sum(len(l.replace("int", "", 1).split(",")) for l in body.splitlines() if l.strip().startswith("int "))
    assert (n_ptr, n_int) == (11, 25)
    assert ctypes.sizeof(_lib.GemmArgs) == n_ptr * 8 + ((n_int * 4 + 7) // 8) * 8


def test_state_dict_contract_and_flat_freeze():
    from oracle import VIT_CONFIGS
    from oracle.model import generator_state_shapes
    from miphei_vit_amd.generators import get_generator, get_vitmatte
    m = get_vitmatte("tiny_swiglu", 128, 16, use_lora=True, pretrained=False)
    sd = m.state_dict()
    shapes = generator_state_shapes(VIT_CONFIGS["tiny_swiglu"], 128, 16)
    assert sorted(sd) == sorted(shapes)
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    train = {k for k, p in m.named_parameters() if p.requires_grad}
    assert all((".lora_" in k) or k.startswith("decoder.") for k in train)
    assert not any(p.requires_grad for k, p in m.named_parameters() if k.startswith("encoder.") and ".lora_" not in k)
    assert hasattr(m, "encoder") and hasattr(m.encoder, "vit") and m.encoder.num_prefix_tokens == 5
    assert m.encoder.grid_size == (9, 9) and m.encoder.embed_dim == 96
    assert abs(m.encoder.scale_factor[0] - 8 / 9) < 1e-12
    # patch-16 encoders: no re-grid, and the attribute is None as in the reference (mipheivit.py:150-157)
    assert get_vitmatte("tiny", 128, 3, use_lora=True, pretrained=False).encoder.scale_factor is None
    cfg = {"model": {"model_name": "myvitmatte", "encoder": {"encoder_name": "tiny", "encoder_weights": None,
                                                             "pretrained": False}}}
    g = get_generator("myvitmatte", 128, 3, 3, cfg)
    assert type(g).__name__ == "ViTMatte"
    with pytest.raises(NotImplementedError):
        get_generator("smp_unet", 128, 3, 3, cfg)
    with pytest.raises(ValueError):
        m.set_input_size((100, 100))
    with pytest.raises(ValueError):
        m.set_input_size((64, 64))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 128, 128))  # no CPU fallback: the product path fails loudly without a ROCm device


def test_hoptimus0_key_contract_on_meta_device(golden_dir):
    from miphei_vit_amd.generators import get_vitmatte
    g = np.load(os.path.join(golden_dir, "keys_f256.npz"))
    with torch.device("meta"):
        m = get_vitmatte("hoptimus0", 256, 16, use_lora=True, pretrained=False)
    assert sorted(m.state_dict().keys()) == list(g["keys"]) and len(m.state_dict()) == 945
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_train == 6697712
    assert sum(p.numel() for p in m.parameters()) == 1141576432


def test_lr_schedule_loss_and_resample_tables():
    import torch.nn.functional as F
    from miphei_vit_amd.loss import WeightedMSELoss, marker_weights_from_stats
    from miphei_vit_amd.resample import _DENSE
    from miphei_vit_amd.utils import pix2pix_lr_scheduler
    from oracle import weighted_mse_loss
    f = pix2pix_lr_scheduler(1000, 400, 500)
    assert [f(0), f(200), f(450), f(750), f(1000)] == [0.0, 0.5, 1.0, 0.5, 0.0]
    w = marker_weights_from_stats([2.0, 1.0, 4.0])
    assert torch.allclose(w, torch.tensor([2.0, 4.0, 1.0]))
    y, p = torch.randn(2, 3, 8, 8), torch.randn(2, 3, 8, 8)
    assert torch.allclose(WeightedMSELoss(50.0, w)(y, p), weighted_mse_loss(y, p, w, 50.0))
    for mode, (i, o) in [("bilinear", (16, 32)), ("bicubic", (18, 16)), ("bicubic", (36, 32))]:
        R = torch.from_numpy(_DENSE[mode](i, o))
        x = torch.randn(1, 1, i, i, dtype=torch.float64)
        ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) if mode == "bilinear" else \
            F.interpolate(x, scale_factor=(o / i, o / i), mode="bicubic")
        assert float((R @ x[0, 0] @ R.T - ref[0, 0]).abs().max()) < 1e-12


def test_swiglu_pack_index_is_a_permutation():
    from miphei_vit_amd.engine import swiglu_pack_index
    for hidden in (512, 8192):
        idx = swiglu_pack_index(hidden)
        assert sorted(idx.tolist()) == list(range(hidden))
        H = hidden // 2
        assert idx[0] == 0 and idx[32] == H and idx[64] == 32 and idx[96] == H + 32


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from miphei_vit_amd.trainer import allreduce_mean_
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
g = torch.Generator().manual_seed(7)
full = torch.randn(world, 1000, generator=g)          # per-rank gradients of a sharded minibatch
mine = full[rank].clone()
allreduce_mean_(mine, world)
assert torch.allclose(mine, full.mean(0), atol=1e-6), "all-reduce mean mismatch"
# sharding a minibatch: every rank gets a disjoint, equal slice of the global tile indices
B = 8
idx = torch.arange(B * world)[rank * B:(rank + 1) * B]
gathered = [torch.empty_like(idx) for _ in range(world)]
dist.all_gather(gathered, idx)
assert torch.equal(torch.cat(gathered), torch.arange(B * world))
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_gloo_world2_gradient_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_checkpoint_prune_load_validate(tmp_path):
    from miphei_vit_amd.checkpoint import (get_generator_state_dict, load_generator_checkpoint, save_pruned_safetensors,
                                           validate_load_info)
    from miphei_vit_amd.generators import get_vitmatte
    torch.manual_seed(0)
    src = get_vitmatte("tiny_swiglu", 128, 3, use_lora=True, pretrained=False)
    with torch.no_grad():
        for p in src.parameters():
            if p.requires_grad:
                p.add_(torch.randn_like(p) * 0.1)
    keys = save_pruned_safetensors(src, tmp_path / "model.safetensors")
    assert all((".lora" in k) or k.startswith("decoder.") for k in keys)
    assert not any(k.startswith("encoder.vit.") and ".lora" not in k for k in keys)
    dst = get_vitmatte("tiny_swiglu", 128, 3, use_lora=True, pretrained=False)
    info = load_generator_checkpoint(dst, tmp_path)
    assert info.missing_keys and all(k.startswith("encoder.vit.") for k in info.missing_keys)
    a, b = src.state_dict(), dst.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in keys)
    # error conventions of validate_load_info
    from safetensors.torch import load_file, save_file
    sd = load_file(str(tmp_path / "model.safetensors"))
    bad = dict(sd); bad["decoder.bogus"] = torch.zeros(1)
    with pytest.raises(ValueError, match="Unexpected"):
        validate_load_info(dst.load_state_dict(bad, strict=False))
    no_lora = {k: v for k, v in sd.items() if ".lora_q.A" not in k}
    with pytest.raises(ValueError, match="Missing LoRA"):
        validate_load_info(dst.load_state_dict(no_lora, strict=False))
    no_dec = {k: v for k, v in sd.items() if k != "decoder.fusion_blks.0.conv.conv.weight"}
    with pytest.raises(ValueError, match="Missing key"):
        validate_load_info(dst.load_state_dict(no_dec, strict=False))
    lightning = {"generator." + k: v for k, v in src.state_dict().items()}
    lightning["loss_reconstruct.marker_weights"] = torch.ones(3)
    assert sorted(get_generator_state_dict(lightning)) == sorted(src.state_dict())


def test_pos_embed_resize_on_load_values():
    """Load-time re-grid of a 224-pixel checkpoint (16x16 table) to the 256 / 512-pixel grids, and a down-sampling case, value by
    value against the independent restatement of timm's resample_abs_pos_embed (bicubic, antialias) in oracle/posembed.py."""
    import torch.nn.functional as F
    from oracle.posembed import resample_abs_pos_embed
    from miphei_vit_amd.generators.foundation_models import VisionTransformer, resize_pos_embed_statedict
    for img, old in ((256, 16), (512, 16), (128, 16)):
        m = VisionTransformer(img_size=img, patch_size=14, embed_dim=96, depth=1, num_heads=3, mlp="swiglu", hidden=512)
        g = m.patch_embed.grid_size
        pe = torch.randn(1, old * old, 96, generator=torch.Generator().manual_seed(img))
        out = resize_pos_embed_statedict({"pos_embed": pe.clone()}, m, img)["pos_embed"]
        assert out.shape == (1, g[0] * g[1], 96) and out.dtype == pe.dtype
        want = resample_abs_pos_embed(pe.numpy(), (old, old), tuple(g))
        assert float((out - torch.from_numpy(want)).abs().max()) < 1e-5      # f32 filter weights vs the f64 restatement
        # the antialias flag is part of the contract: plain bicubic (a = -0.75, no low-pass) gives a different table
        plain = F.interpolate(pe.reshape(1, old, old, 96).permute(0, 3, 1, 2), size=tuple(g), mode="bicubic")
        assert float((plain.permute(0, 2, 3, 1).reshape(1, -1, 96) - out).abs().max()) > 1e-3
    # same grid: untouched; a checkpoint carrying a class-token slot loses it
    m = VisionTransformer(img_size=224, patch_size=14, embed_dim=96, depth=1, num_heads=3, mlp="swiglu", hidden=512)
    pe = torch.randn(1, 256, 96)
    assert torch.equal(resize_pos_embed_statedict({"pos_embed": pe}, m, 224)["pos_embed"], pe)
    pe1 = torch.randn(1, 257, 96)
    assert torch.equal(resize_pos_embed_statedict({"pos_embed": pe1}, m, 224)["pos_embed"], pe1[:, 1:])


_DDP_WORKER = r'''
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from miphei_vit_amd.trainer import DataParallelSync
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n, n_lora, L = 1000, 300, 10
class FakeEngine:                      # the exchange only touches the flat buffers and the bucket slices
    def __init__(self):
        g = torch.Generator().manual_seed(100 + rank)
        self._flat = types.SimpleNamespace(flat=torch.full((n,), float(rank)), gflat=torch.randn(n, generator=g), n_lora=n_lora)
        self._pack_key = "x"
    def params_changed(self): self._pack_key = None
    def _ensure_flat(self): return self._flat
    def grad_buckets(self): return self._flat.gflat[n_lora:], self._flat.gflat[:n_lora]
    def lora_blocks(self): return L
eng = FakeEngine()
ref = torch.stack([torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]).mean(0)
for nb, ksplit in ((4, 1), (1, 1), (3, 3), (64, 7)):
    eng = FakeEngine()
    sync = DataParallelSync(eng, lora_buckets=nb, decoder_split=ksplit)
    sync.broadcast_parameters(0)
    assert torch.equal(eng._flat.flat, torch.zeros(n)) and eng._pack_key is None      # rank 0's parameters everywhere
    sync.decoder_ready()        # launched from inside backward once the decoder gradients exist
    assert len(sync._work) == ksplit, (len(sync._work), ksplit)                       # the decoder bucket in `decoder_split` all-reduces
    issued = []
    for l in range(L - 1, -1, -1):  # encoder backward, block 39 -> 0: a sub-bucket goes out when its lowest block is done
        before = len(sync._work)
        sync.lora_block_done(l)
        if len(sync._work) > before: issued.append(l)
    assert len(issued) == min(nb, L) and issued[-1] == 0 and issued == sorted(issued, reverse=True), issued
    sync.finish()               # wait for every bucket + average
    assert torch.allclose(eng._flat.gflat, ref, atol=1e-6), nb
dist.barrier(); dist.destroy_process_group()
'''


def test_gloo_world2_data_parallel_sync(tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_self_launch_dry_gloo_world2():
    """`python bench.py --gpus 2` from a plain shell: the parent spawns torch.distributed.run as a child (it must never
    touch the GPU), the ranks rendezvous on 127.0.0.1, run the bucketed exchange protocol and rank 0 prints ONE JSON line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--backend", "gloo", "--steps",
                        "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["dry"] and r["exchange_ok"] and r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["steps"] == 3
    # a failing child is reported through the exit code, not swallowed
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--backend", "nccl", "--steps",
                        "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip()


def test_bench_parent_does_not_touch_the_gpu_before_launching():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main("):]
    assert main.index("self_launch(a, argv)") < main.index("torch.cuda.")
    assert "os.exec" not in src and "execv" not in src


def test_resume_keeps_the_configured_lr_horizon(tmp_path):
    """Resuming with a larger train.max_steps extends training: the horizon configured on the module before the load wins (as
    Lightning / LambdaLR rebuild the lambda from the new trainer), the checkpoint's one is adopted only when none is set."""
    import warnings
    from miphei_vit_amd.checkpoint import save_checkpoint_atomic
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    mk = lambda: ModelModule(torch.nn.Linear(3, 3), None, 1e-3, 0., WeightedMSELoss(50.0, torch.ones(3)))
    a = mk()
    a.total_iters, a.global_step_ = 1000, 900
    path = save_checkpoint_atomic(a.checkpoint_state(), tmp_path / "last.ckpt")
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    ck = torch.load(path, map_location="cpu", weights_only=True)          # plain tensors / dicts: loads in the safe mode
    b = mk()
    b.total_iters = 4000                                                   # run.py: module.total_iters = train.max_steps
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        b.load_checkpoint_state(ck)
    assert b.global_step_ == 900 and b.total_iters == 4000 and any("horizon" in str(x.message) for x in w)
    assert b.current_lr() == pytest.approx(1e-3)                           # step 900 of 4000: plateau (old horizon: 0.2e-3)
    assert b.current_lr(3000) == pytest.approx(0.5e-3)
    c = mk()
    c.load_checkpoint_state(ck)                                            # nothing configured: the saved horizon is adopted
    assert c.total_iters == 1000 and c.current_lr() == pytest.approx(0.2e-3)
    # an interrupted write leaves the previous checkpoint in place
    class Boom:
        def __reduce__(self):
            raise RuntimeError("disk full")
    with pytest.raises(RuntimeError):
        save_checkpoint_atomic({"x": Boom()}, path)
    assert torch.load(path, map_location="cpu", weights_only=True)["global_step"] == 900
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]


@pytest.mark.parametrize("override,needle", [("++train.losses.use_weighted_mae=true", "use_weighted_mae"),
                                             ("++train.losses.cell_loss.use_loss=true", "cell_loss")])
def test_run_py_refuses_objectives_outside_the_path(override, needle):
    """reference src/train.py:118-150 switches the loss on these keys; run.py must not silently train with WeightedMSELoss"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "run.py"), "+default_configs=tiny", override], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "NotImplementedError" in p.stderr and needle in p.stderr, p.stderr[-2000:]


def test_entry_points_do_not_depend_on_the_benchmark_script():
    for f in ("run.py", "run_inference.py"):
        src = open(os.path.join(ROOT, f)).read()
        assert "from bench" not in src and "import bench" not in src, f


def test_shuffled_indices_are_epoch_permutations_shared_by_all_ranks():
    """run.py's resident-tile sampler (reference: DataLoader(shuffle=True), dataset.py:117-121): every epoch is a permutation of the
    tile set, different from the previous one, and a pure function of the global sample number -- two rank layouts read the same
    stream."""
    from miphei_vit_amd.io_stage import shuffled_indices
    N, B = 37, 8
    e0 = shuffled_indices(0, N, N, seed=3)
    e1 = shuffled_indices(N, N, N, seed=3)
    assert sorted(e0.tolist()) == list(range(N)) and sorted(e1.tolist()) == list(range(N))
    assert e0.tolist() != e1.tolist() and e0.tolist() != list(range(N))
    assert shuffled_indices(0, N, N, seed=4).tolist() != e0.tolist()
    # world 1 (one rank, batches of 2B) vs world 2 (two ranks, batches of B): the same global samples, straddling an epoch edge
    one = torch.cat([shuffled_indices(i * 2 * B, 2 * B, N, 3) for i in range(6)])
    two = torch.cat([torch.cat([shuffled_indices((i * 2 + r) * B, B, N, 3) for r in range(2)]) for i in range(6)])
    assert torch.equal(one, two)
    assert torch.equal(one[:N], e0) and torch.equal(one[N:2 * N], e1)


def test_pretrained_encoder_comes_from_the_hub_or_fails_loudly(tmp_path, monkeypatch):
    """get_vitmatte(..., pretrained=True) without a ckpt_path (reference foundation_models.py:59-64: timm's load_state_dict_from_hf):
    the weights are fetched through huggingface_hub (model.safetensors, then pytorch_model.bin) and loaded; without network and
    without a cached copy the factory raises, naming the hub id and the ways out -- never silent random weights."""
    import huggingface_hub
    from safetensors.torch import save_file
    from miphei_vit_amd.generators import foundation_models as fm

    def no_network(repo_id, filename, **kw):
        raise OSError(f"offline: cannot reach the hub for {repo_id}/{filename}")
    monkeypatch.setattr(huggingface_hub, "hf_hub_download", no_network)
    monkeypatch.delenv("MIPHEI_RANDOM_INIT", raising=False)
    with pytest.raises(RuntimeError, match="bioptimus/H-optimus-0"):
        fm._hub_checkpoint("hoptimus0")
    with pytest.raises(RuntimeError, match="no hub id"):
        fm._hub_checkpoint("tiny")
    # a hub that answers: the downloaded file is what the model ends up with (tiny encoder standing in for the 1.1 B one)
    src = fm.tiny_gelu(128)
    sd = {k: torch.randn_like(v) for k, v in src.state_dict().items()}
    path = str(tmp_path / "model.safetensors")
    save_file(sd, path)
    seen = []
    monkeypatch.setattr(huggingface_hub, "hf_hub_download", lambda repo_id, filename, **kw: seen.append((repo_id, filename)) or path)
    monkeypatch.setitem(fm.FOUNDATION_HF_CKPT_REGISTRY, "tiny", "someone/tiny-vit")
    got = fm._build(128, True, None, "tiny", patch_size=16, embed_dim=64, depth=2, num_heads=4, mlp="gelu", hidden=256, reg_tokens=4,
                    init_values=1e-5, global_pool="")
    assert seen == [("someone/tiny-vit", "model.safetensors")]
    assert all(torch.equal(v, sd[k]) for k, v in got.state_dict().items())
