"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports every symbol of include/idiff.h,
host logic (options, registries, schedules, caches, sharding) and the loud-failure contract.  No kernel
is launched here."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from instancediff_amd import _lib, options, pipeline  # noqa: E402
from instancediff_amd.models import create_model  # noqa: E402
from instancediff_amd.models.SDEs import create_sde  # noqa: E402
from instancediff_amd.models.modules import create_net  # noqa: E402
from instancediff_amd.utils.sde_utils import IRSDE  # noqa: E402
from instancediff_amd.utils.synthetic import ARTIFACT_TYPES, make_batch  # noqa: E402
from oracle import sde_ref  # noqa: E402


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "instancediff_amd", "csrc"), "-j", "4"], check=True)
    return _lib.load()


def test_cabi_exports_every_header_symbol(built_lib):
    syms = _lib.header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(built_lib, s), f"{s} declared in include/idiff.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms
    assert built_lib.idiff_version() >= 1
    assert built_lib.idiff_conv2d_num_tiles(256, 256) == 256  # 8x32 tiles
    assert built_lib.idiff_conv2d_num_tiles(16, 16) == 1
    assert built_lib.idiff_smm_xattn_ws_floats(16, 5, 4, 256, 65536) > 0


def test_cabi_argument_errors_are_reported_not_crashes(built_lib):
    import ctypes as C
    d = _lib.ConvDesc()
    assert built_lib.idiff_conv2d_fwd(C.byref(d), None) == -1
    assert b"null pointer" in built_lib.idiff_last_error()
    assert built_lib.idiff_irsde_reverse_step(None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None) == -1
    assert built_lib.idiff_linear_fwd(None, 0, None, 0, None, None, 0, None, None, 0, 1, 1, 1, 0, 0, None) == -1
    k6 = (C.c_float * 6)(0, 0, 0, 0, 0, 0)
    assert built_lib.idiff_irsde_map(99, None, None, None, None, 0.0, None, 1, 16, None, k6, 0, 0, None) == -1
    assert b"bad op" in built_lib.idiff_last_error()
    assert built_lib.idiff_irsde_map(0, None, None, None, None, 0.0, None, 1, 16, None, k6, 0, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libidiff_hip.so")
    with pytest.raises(_lib.IdiffError, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors(built_lib):
    from instancediff_amd import ops
    x = torch.zeros(1, 8, 8, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.affine_silu_add(x)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.irsde_reverse_step(x, x, x, None, 0.1, 0.1, 0.1, 0.1, 0.3)
    sde = IRSDE(0.4, T=10, device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="GPU"):  # the reference-surface methods go through the same C ABI: no torch-CPU arithmetic
        sde.reverse_sde_step(x, x, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        sde.mu_bar(x, torch.full((1, 1, 1, 1), 3))


# ---- options / registries ---------------------------------------------------------------------------
def test_options_parse_train_and_test(tmp_path):
    opt = options.parse(pipeline.DEFAULT_YAML, is_train=True)
    assert opt["is_train"] and opt["path"]["models"].endswith(os.path.join("experiments", "UM_IDDM_SM_IB", "models"))
    assert opt["datasets"]["train"]["phase"] == "train" and opt["datasets"]["val"]["scale"] == 1
    assert opt["models"]["DriftNoise"]["nnet_settings"]["ch_mult"] == [1, 2, 4, 4]
    assert opt["models"]["DriftNoise"]["drift_net_lr"] == pytest.approx(2e-5)
    nd = options.dict_to_nonedict(opt)
    assert nd["no_such_key"] is None and nd["train"]["no_such_key"] is None
    t = options.parse(pipeline.DEFAULT_YAML, is_train=False)
    assert "results_root" in t["path"] and "experiments_root" not in t["path"]
    assert "train" in options.dict2str(opt)
    # debug names switch the frequencies (options.py:80-83)
    p = tmp_path / "dbg.yml"
    p.write_text(open(pipeline.DEFAULT_YAML).read().replace("name: UM_IDDM_SM_IB", "name: debug_run"))
    d = options.parse(str(p), is_train=True)
    assert d["train"]["val_freq"] == 8 and d["logger"]["print_freq"] == 1


def test_registry_builds_reference_config_on_cpu():
    model, sde = pipeline.build(phase="test", device=torch.device("cpu"), T=20)
    from instancediff_amd.models.drift_noise_model import CLIPDriftModel
    assert isinstance(model, CLIPDriftModel)
    for name in ("feed_data", "set_sde", "optimize_parameters", "test", "get_visuals", "get_nets", "save", "load",
                 "save_training_state", "resume_training", "set_eval", "set_train", "set_gpu", "reinit_loss_message",
                 "get_loss_message", "get_current_learning_rate", "update_lr", "optimize_parameters_inputRes"):
        assert hasattr(model, name), name
    nets = model.get_nets()
    assert set(nets) == {"noise_net", "drift_net"}
    n_params = sum(p.numel() for p in nets["drift_net"].parameters())
    assert 30e6 < n_params < 40e6
    keys = list(nets["drift_net"].state_dict().keys())
    assert any(k.startswith("CLIP_ScoreMapModule.0.") for k in keys)
    assert sde.T == 20 and sde.max_sigma == 0.4 and len(sde.drift_schedule) == 21
    # the CPU has no kernels: forward must raise, not fall back
    b = make_batch(1, 32)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            nets["drift_net"](b["input"], b["input"], torch.tensor([3.0]), b["names"], model.text_encoder, image_context=b["A_emb"])


def test_create_model_and_create_sde_registry_names():
    opt = pipeline.load_options()
    m = create_model({"dist": False, "nepoch": 1}, opt["models"]["DriftNoise"], phase="test")
    s = create_sde(m.get_nets(), opt["sdes"]["driftSDE"])
    assert type(s).__name__ == "driftSDE" and s.T == 100
    with pytest.raises((ImportError, AttributeError)):
        create_net({"module_name": "MSM_degEmb_Unet", "class_name": "NoSuchNet"})


def test_checkpoint_file_names_roundtrip(tmp_path):
    model, _ = pipeline.build(phase="test", device=torch.device("cpu"), T=4)
    model.save(123, str(tmp_path))
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["123_DP.pth", "123_NP.pth", "123_DN.pth", "123_NN.pth", "lastest_DP_ema.pth", "lastest_NP_ema.pth",
                            "lastest_DN_ema.pth", "lastest_NN_ema.pth"])
    sd = torch.load(tmp_path / "lastest_DN_ema.pth")
    assert any(k.startswith("online_model.") for k in sd) and any(k.startswith("ema_model.") for k in sd)
    w0 = model.drift_net.init_conv.weight.detach().clone()
    with torch.no_grad():
        model.drift_net.init_conv.weight.zero_()
    # a reference-era DDP checkpoint has 'module.' prefixes (drift_noise_model.py:712-730)
    dn = torch.load(tmp_path / "123_DN.pth")
    torch.save({("module." + k): v for k, v in dn.items()}, tmp_path / "123_DN.pth")
    model.load(123, str(tmp_path))
    assert torch.equal(model.drift_net.init_conv.weight, w0)


# ---- schedules ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kw", [("cos100", dict(max_sigma=0.4, T=100, schedule="cosine", eps=0.01)),
                                     ("cos100_s50", dict(max_sigma=0.4, T=100, sample_T=50, schedule="cosine", eps=0.01)),
                                     ("cos1000", dict(max_sigma=0.4, T=1000, schedule="cosine", eps=0.01)),
                                     ("lin100", dict(max_sigma=0.4, T=100, schedule="linear", eps=0.01)),
                                     ("const100", dict(max_sigma=0.4, T=100, schedule="constant", eps=0.01)),
                                     ("cos100_ms50", dict(max_sigma=50, T=100, schedule="cosine", eps=0.01))])
def test_product_irsde_tables_bit_exact_vs_reference_golden(golden_sde, name, kw):
    sde = IRSDE(device=torch.device("cpu"), **kw)
    for k in ["thetas", "sigmas", "thetas_cumsum", "sigma_bars"]:
        assert np.array_equal(getattr(sde, k).numpy(), golden_sde[f"{name}/{k}"]), k
    assert float(sde.dt) == float(golden_sde[f"{name}/dt"])
    assert sde.max_sigma == float(golden_sde[f"{name}/max_sigma"]) and sde.sample_scale == float(golden_sde[f"{name}/sample_scale"])
    with pytest.raises(ValueError):
        IRSDE(0.4, schedule="nope")


def test_product_drift_sde_tables_equal_oracle():
    from instancediff_amd.models.SDEs.driftSDE import driftSDE
    for sched in ("sigmoid", "cosine", "linear"):
        p = driftSDE(nets={}, T=50, max_sigma=0.3, drift_schedule=sched, noise_schedule=sched)
        o = sde_ref.DriftSDERef(50, None, None, max_sigma=0.3, drift_schedule=sched, noise_schedule=sched)
        assert torch.equal(p.drift_schedule, o.drift_schedule) and torch.equal(p.noise_schedule, o.noise_schedule)
        assert torch.equal(p._a, o.a) and torch.equal(p._b, o.b) and torch.equal(p._c, o.c)
    assert float(p._c[1]) == 0.0  # last step is noise free (s_0 = 0)


def test_synthetic_batch_layout():
    b = make_batch(7, 32, seed=3)
    assert set(b) == {"input", "target", "names", "A_emb"}
    assert b["input"].shape == b["target"].shape == (7, 1, 32, 32) and b["A_emb"].shape == (7, 1, 512)
    assert set(b["names"]) <= set(ARTIFACT_TYPES) and len(set(b["names"])) == 5
    assert float(b["target"].min()) >= -1.0 and float(b["target"].max()) <= 1.0
    assert torch.allclose(b["A_emb"].norm(dim=-1), torch.ones(7, 1))
    b2 = make_batch(7, 32, seed=3)
    assert torch.equal(b["input"], b2["input"])


# ---- multi-process (gloo, world_size 2) ----------------------------------------------------------------
WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["IDIFF_ROOT"])
import torch.distributed as dist
from instancediff_amd import parallel
rank, world, local = parallel.init_distributed(backend="gloo")
assert world == 2 and dist.get_world_size() == 2
torch.manual_seed(100 + rank)           # different init per rank: broadcast must equalise it
net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
sync = parallel.FlatGradAllReduce(list(net.parameters()))
sync.broadcast_parameters()
w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
ws = [torch.zeros_like(w) for _ in range(2)]
dist.all_gather(ws, w)
assert torch.equal(ws[0], ws[1])
x = torch.full((3, 8), float(rank + 1))
net(x).sum().backward()
for p in net.parameters():              # grads landed in the flat buffer (views)
    assert p.grad.data_ptr() >= sync.flat.data_ptr()
g_local = sync.flat.clone()
gs = [torch.zeros_like(g_local) for _ in range(2)]
dist.all_gather(gs, g_local)
f = sync.all_reduce(average=False)
assert abs(f - 0.5) < 1e-12
assert torch.allclose(sync.flat, gs[0] + gs[1])
# GradSync: the form the fused-Adam train step uses (one all-reduce per optimizer-owned flat buffer, scale deferred)
gs = parallel.GradSync()
fa, fb = torch.full((7,), float(rank + 1)), torch.arange(5, dtype=torch.float32) * (rank + 1)
scale = gs.all_reduce_flat([fa, fb])
assert scale == 0.5 and torch.equal(fa, torch.full((7,), 3.0)) and torch.equal(fb, torch.arange(5, dtype=torch.float32) * 3)
pw = torch.nn.Parameter(torch.full((4,), float(rank)))
gs.broadcast_parameters([pw])
assert torch.equal(pw.data, torch.zeros(4))
# sampling shards: disjoint cover (data_sampler.py:59 semantics)
idx = parallel.shard_indices(11, rank, world)
alli = [None, None]
dist.all_gather_object(alli, idx)
assert sorted(alli[0] + alli[1]) == list(range(11)) and not set(alli[0]) & set(alli[1])
dist.barrier()
dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"rank{rank}.ok"), "w").write("ok")
"""


def test_gloo_world2_flat_grad_allreduce_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IDIFF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout[-2000:]


# ---- text-encoder selection is explicit: nothing falls back to random embeddings silently --------------------
def test_text_encoder_selection_is_loud(tmp_path):
    from instancediff_amd.models.text_encoder import StubTextEncoder, build_text_encoder
    enc, dim = build_text_encoder(None, "stub")
    assert isinstance(enc, StubTextEncoder) and dim == 512 and not any(p.requires_grad for p in enc.parameters())
    with pytest.raises(FileNotFoundError, match="refusing"):
        build_text_encoder("pretrained/ViT-B-32.pt", "CLIP")  # the reference's configured path (config.yml:137), absent here
    with pytest.raises(ValueError):
        build_text_encoder(None, "CLIP")
    present = tmp_path / "ViT-B-32.pt"
    present.write_bytes(b"not a checkpoint")
    with pytest.raises(Exception, match="(?i)zip|archive|jit|pytorch|constants|file"):  # a real archive is parsed by torch.jit.load, as the reference does
        build_text_encoder(str(present), "CLIP")
    with pytest.raises(Exception, match="(?i)pickle|zip|archive|load|invalid|weights"):  # HFContextTextEncoder.init_weights parses it with torch.load
        build_text_encoder(str(present), "BiomedCLIP")
    opt = pipeline.load_options()
    assert opt["models"]["DriftNoise"]["CLIP_Type"] == "stub"  # the shipped synthetic configuration asks for the stub by name


def test_placeholder_class_tokens_are_refused_by_a_token_reading_encoder():
    from instancediff_amd.models.modules.MSM_degEmb_Unet import ScoreMapModule
    smm = ScoreMapModule(visual_dim=64)
    assert smm.tokens_are_placeholders

    class Reads(torch.nn.Module):
        def forward(self, text, context):
            return torch.zeros(context.shape[0], text.shape[0], 512)

    with pytest.raises(RuntimeError, match="placeholder"):
        smm.text_embeddings(Reads(), 2)
    ids = torch.arange(5 * 12).reshape(5, 12)
    smm.set_class_tokens(ids)
    assert not smm.tokens_are_placeholders and smm.text_embeddings(Reads(), 2).shape == (2, 5, 512)
    # real ids travel with the state dict (any prompt length) and placeholders stay flagged across save/load
    other = ScoreMapModule(visual_dim=64)
    other.load_state_dict(smm.state_dict())
    assert torch.equal(other.tokens, ids) and not other.tokens_are_placeholders
    fresh = ScoreMapModule(visual_dim=64)
    fresh.load_state_dict(ScoreMapModule(visual_dim=64).state_dict())
    assert fresh.tokens_are_placeholders
    tok = ScoreMapModule(visual_dim=64, tokenizer=lambda names: torch.ones(len(names), 9, dtype=torch.long))
    assert tok.tokens.shape == (5, 9) and not tok.tokens_are_placeholders


def test_reference_era_state_file_resumes(tmp_path):
    """`{iter}.state` as the reference writes it -- pickled torch.optim.Adam and CosineAnnealingLR OBJECTS
    (models/drift_noise_model.py:694-704) -- is taken over into the fused optimizers: moments, step count, lr schedule."""
    from instancediff_amd.train_ops import FusedAdam
    torch.manual_seed(0)
    net_ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    net_new = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    net_new.load_state_dict(net_ref.state_dict())
    adam = torch.optim.Adam(net_ref.parameters(), lr=2e-5, betas=(0.9, 0.99), weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(adam, T_max=50, eta_min=1e-6)
    for _ in range(3):
        adam.zero_grad()
        net_ref(torch.randn(4, 6)).square().mean().backward()
        adam.step()
        sched.step()
    path = tmp_path / "3.state"
    torch.save({"epoch": 1, "iter": 3, "schedulers": [sched], "optimizers": [adam]}, path)
    state = torch.load(path, map_location="cpu", weights_only=False)
    fused = FusedAdam(net_new.parameters(), lr=1.0, betas=(0.5, 0.5), weight_decay=0.0)
    fsched = torch.optim.lr_scheduler.CosineAnnealingLR(fused, T_max=50, eta_min=1e-6)
    fused.load_torch_adam(state["optimizers"][0])
    fsched.load_state_dict({k: v for k, v in state["schedulers"][0].state_dict().items() if k != "optimizer"})
    f = fused._flat[0]
    assert f["step"] == 3 and fused.param_groups[0]["betas"] == (0.9, 0.99) and fused.param_groups[0]["weight_decay"] == 1e-4
    want_m = torch.cat([adam.state[p]["exp_avg"].reshape(-1) for p in net_ref.parameters()])
    want_v = torch.cat([adam.state[p]["exp_avg_sq"].reshape(-1) for p in net_ref.parameters()])
    assert torch.equal(f["m"], want_m) and torch.equal(f["v"], want_v)
    assert fsched.last_epoch == 3 and abs(fused.param_groups[0]["lr"] - adam.param_groups[0]["lr"]) < 1e-12
    # this build's own format still round-trips
    sd = fused.state_dict()
    again = FusedAdam(torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)).parameters(), lr=2e-5)
    again.load_state_dict(sd)
    assert again._flat[0]["step"] == 3 and torch.equal(again._flat[0]["m"], want_m)


def test_reference_era_state_file_resumes_through_the_model_entry_point(tmp_path):
    """The path trainUM.py takes: CLIPDriftModel.load_training_state(path) (torch.load under weights_only with an allow-list of the
    classes a reference-era `.state` pickles) -> model.resume_training(state).  A plain torch.load(path) of such a file raises
    UnpicklingError under torch >= 2.6 defaults, which is what trainUM.py used to call."""
    import pickle
    model, _ = pipeline.build(phase="train", device=torch.device("cpu"), T=4)
    adams, scheds = [], []
    for net, lr in ((model.drift_net, 3e-5), (model.noise_net, 2e-5)):
        params = [torch.nn.Parameter(p.detach().clone()) for p in net.parameters()]
        adam = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99), weight_decay=1e-4)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(adam, T_max=50, eta_min=1e-6)
        g = torch.Generator().manual_seed(int(lr * 1e7))
        for _ in range(2):
            for p in params[:6]:  # parameters without state (never stepped) are legal too
                p.grad = torch.randn(p.shape, generator=g) * 1e-3
            adam.step()
            sched.step()
        adams.append(adam), scheds.append(sched)
    path = tmp_path / "2.state"
    torch.save({"epoch": 0, "iter": 2, "schedulers": scheds, "optimizers": adams}, path)  # models/drift_noise_model.py:694-699
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu")
    state = model.load_training_state(str(path))
    assert state["iter"] == 2
    model.resume_training(state)
    for fused, adam, sch in ((model.drift_optimizer, adams[0], model.drift_lr_scheduler), (model.noise_optimizer, adams[1], model.noise_lr_scheduler)):
        f = fused._flat[0]
        first = adam.param_groups[0]["params"][0]
        n = first.numel()
        assert f["step"] == 2 and torch.equal(f["m"][:n].cpu(), adam.state[first]["exp_avg"].reshape(-1))
        assert fused.param_groups[0]["betas"] == (0.9, 0.99) and sch.last_epoch == 2
        assert abs(fused.param_groups[0]["lr"] - adam.param_groups[0]["lr"]) < 1e-12
    # this build's own layout goes through the same entry point
    model.save_training_state(0, 5, str(tmp_path))
    again = model.load_training_state(str(tmp_path / "5.state"))
    model.resume_training(again)
    # a pickle holding anything else is refused unless declared trusted
    evil = tmp_path / "9.state"
    torch.save({"iter": 9, "optimizers": [os.getcwd]}, evil)  # a global outside the allow-list
    with pytest.raises(RuntimeError):
        model.load_training_state(str(evil))
    assert model.load_training_state(str(evil), trusted=True)["iter"] == 9
    # a legacy-format (non-zip) file that is ONE pickle whose __reduce__ names a foreign callable: torch's legacy loader reads its
    # header with pickle_module.load(f), which must be the allow-listing unpickler too -- the callable is never called
    marker = tmp_path / "called"

    class Boom:
        def __reduce__(self):
            return (os.mkdir, (str(marker),))
    legacy = tmp_path / "7.state"
    legacy.write_bytes(pickle.dumps(Boom(), protocol=2))
    with pytest.raises((RuntimeError, pickle.UnpicklingError)):
        model.load_training_state(str(legacy))
    assert not marker.exists()


def test_bench_spawns_one_rank_per_gpu_with_the_rendezvous_environment(tmp_path):
    """`python bench.py --gpus N` started plainly becomes the launcher: N fresh rank processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set (the reference's one-command launch, trainUM.py:50-66); a failing rank stops the others and the exit code is its."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    child = ("import os, json, sys; open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write(json.dumps({k: os.environ.get(k) "
             "for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')}))")
    assert bench.spawn_ranks(3, [sys.executable, "-c", child, str(tmp_path)]) == 0
    envs = [json.loads((tmp_path / f"r{r}").read_text()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    # rank 1 fails at once, rank 0 would wait forever: the launcher stops it and reports rank 1's code
    hang = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(600)"
    import time
    t0 = time.time()
    assert bench.spawn_ranks(2, [sys.executable, "-c", hang]) == 7
    assert time.time() - t0 < 60
    # the real command line: without a GPU every rank refuses (exit 2, "needs a GPU") and so does the launcher -- but it did launch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    if not torch.cuda.is_available():
        assert r.returncode == 2 and r.stderr.count("needs a GPU") >= 1 and "rank" in r.stderr


def test_bench_train_mode_reaches_the_gradient_exchange_at_world_2(tmp_path):
    """`python bench.py --gpus 2 --mode train --grad-wire bf16` (BASELINE c3, the only configuration with a collective; reference
    launch README.md:35, trainUM.py:50-66) through spawn_ranks: both ranks rendezvous, build the train-phase model with its GradSync,
    take rank 0's weights and exchange the flat gradient buffers -- everything around the HIP compute, which needs a GPU
    (IDIFF_BENCH_DRYRUN=1 runs exactly that much and says so in its line)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(IDIFF_BENCH_DRYRUN="1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--grad-wire", "bf16", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=900)
    if torch.cuda.is_available():
        pytest.skip("dry run is the no-GPU hook; on a GPU box the rehearsal (IDIFF_BENCH_REHEARSAL=1) runs the real step")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["grad_sync_ok"] is True and line["grad_wire"] == "bf16" and line["scale"] == 0.5
    assert line["flat_gradient_floats"] > 60e6 and "dryrun" in line   # both nets' parameters in the flat buffers


def test_fused_adam_zero_grad_honours_set_to_none():
    """FusedAdam.zero_grad: set_to_none=True drops the per-parameter gradients (the backward's tensors are then taken as they are and
    gathered in one launch); set_to_none=False is torch.optim's meaning -- zero-filled gradients that a backward accumulates into,
    bound to the flat buffer.  (CPU: the host logic only.)"""
    from instancediff_amd.train_ops import FusedAdam
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    opt = FusedAdam(net.parameters(), lr=1e-3)
    x = torch.randn(5, 4)
    net(x).sum().backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    opt.zero_grad(set_to_none=False)
    flat = opt.flat_grads()[0]
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in net.parameters())
    assert next(net.parameters()).grad.data_ptr() == flat.data_ptr()
    net(x).sum().backward()
    net(x).sum().backward()          # accumulates
    for p, g in zip(net.parameters(), g1):
        assert torch.allclose(p.grad, 2 * g)
    assert torch.allclose(opt.flat_grads()[0], torch.cat([2 * g.reshape(-1) for g in g1]))


def test_bench_default_line_carries_the_training_exchange_at_world_2(tmp_path):
    """`python bench.py --gpus 2` -- the command the driver's scaling run issues -- must put the collective into ITS line: the default
    mode's N > 1 path runs BASELINE c3 across the ranks (train_leg_ranks) and reports it under "train" with the exchange measured for
    the fp32 wire and the bf16 wire variant.  Without a GPU the dry run goes through the same rendezvous / GradSync / timing-mark code
    for both wires and prints the line's shape (no rates)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(IDIFF_BENCH_DRYRUN="1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=900)
    if torch.cuda.is_available():
        pytest.skip("dry run is the no-GPU hook; on a GPU box the rehearsal (IDIFF_BENCH_REHEARSAL=1) runs the real step")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    tr = line["train"]
    assert line["n_gpus"] == 2 and line["metric"].startswith("denoising steps/sec") and tr["n_gpus"] == 2 and tr["batch_per_gpu"] == 32
    for leg, wire in ((tr, "fp32"), (tr["bf16_wire_variant"], "bf16")):
        assert leg["grad_sync_ok"] is True and leg["grad_wire"] == wire and leg["scale"] == 0.5 and leg["flat_gradient_floats"] > 60e6
        ex = leg["exchange"]
        assert ex["wire"] == wire and ex["span_ms"] >= ex["exposed_ms"] >= 0.0


def test_grad_sync_timing_marks_cover_one_step_per_finish():
    """GradSync(timing=True) on the CPU path without a process group is inactive and records nothing; the mark bookkeeping itself
    (first start() of a step opens it, finish() closes it, drain() forgets it) is exercised at world 2 by the dry-run tests."""
    from instancediff_amd.parallel import GradSync
    g = GradSync()
    g.timing = True
    g.start([torch.zeros(4)])
    assert g.finish() == 1.0 and g.timings() == []


def test_hidden_register_loads_of_conv_wino4_are_only_touched_behind_a_wait(tmp_path):
    """conv_wino4.hip (SPEC 2: GroupNorm/FiLM + SiLU prologue) requests its input patch with inline-asm buffer loads that hipcc does
    not count: scripts/lint_asm_loads.py checks the generated ISA -- no instruction may name a destination register between the load
    and a counted s_waitcnt of ours (a compiler copy or spill there would read a register whose data has not landed)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = tmp_path / "conv_wino4.s"
    src = os.path.join(ROOT, "instancediff_amd", "csrc", "conv_wino4.hip")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", str(asm)],
                   check=True, capture_output=True, timeout=900)
    syms = sorted({ln.split(":")[0] for ln in asm.read_text().splitlines() if ln.startswith("_ZN") and "conv_wino4_kernelILi0ELi2E" in ln.split(":")[0] and ":" in ln})
    assert len(syms) == 2, syms   # whole and partial patches
    for sym in syms:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_asm_loads.py"), str(asm), sym], capture_output=True, text=True)
        assert r.returncode == 0 and "violations: 0" in r.stdout, r.stdout[-2000:]
        assert "asm register loads: 0 " not in r.stdout   # the lint did see them
    # r05: the item-crossing pipeline COUNTS an epilogue's stores (s_waitcnt vmcnt(NL + 16)) instead of draining them --
    # scripts/lint_asm_stores.py walks the control-flow graph of every non-RAG instantiation (plain, prologue, two sources, upsample) and
    # requires exactly 16 + 4 buffer stores on every path of an item-loop iteration, for both wave-class bodies
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_asm_stores.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0 and "symbols: 4 violations: 0" in r.stdout, r.stdout[-3000:]


def test_pointer_rebuilt_from_a_signed_low_half_breaks_when_bit_31_is_set():
    """The arithmetic of the r03 GPU abort: __builtin_amdgcn_readfirstlane returns int; `(u64)hi << 32 | lo` with
    lo still an int converts lo to 64 bits by SIGN extension, so a wave-uniform pointer rebuilt that way for a buffer resource gets
    0xffff.... as its upper half whenever bit 31 of the address is set.  Every such rebuild in csrc/ goes through `unsigned`
    (scalar_ptr); this pins the lesson, and a source check keeps the pattern out."""
    import glob
    import re

    import numpy as np

    def rebuild_r03(addr):
        lo = np.int32(np.uint32(addr & 0xffffffff))
        hi = np.int32(np.uint32(addr >> 32))
        v = (np.uint64(np.uint32(hi)) << np.uint64(32)) | np.uint64(np.int64(lo))   # usual arithmetic conversions: int -> u64
        return int(v) & 0xffffffffffff                                                # 48-bit base of a buffer resource

    def rebuild_fixed(addr):
        lo, hi = np.uint32(addr & 0xffffffff), np.uint32(addr >> 32)
        return int((np.uint64(hi) << np.uint64(32)) | np.uint64(lo)) & 0xffffffffffff
    clear, set_ = 0x7f2a_1234_5600, 0x7f2a_9234_5600
    assert rebuild_r03(clear) == clear and rebuild_r03(set_) == 0xffff_9234_5600 != set_
    assert rebuild_fixed(clear) == clear and rebuild_fixed(set_) == set_
    # no kernel source ORs a readfirstlane result into a wider value without going through an unsigned variable first
    bad = []
    for path in glob.glob(os.path.join(ROOT, "instancediff_amd", "csrc", "*.h*")):
        for n, line in enumerate(open(path), 1):
            if re.search(r"<<\s*32\)?\s*\|\s*__builtin_amdgcn_readfirstlane", line) or re.search(r"\(unsigned long long\)\s*__builtin_amdgcn_readfirstlane", line):
                bad.append(f"{os.path.basename(path)}:{n}")
    assert not bad, bad


def test_variant_builders_carry_the_makefiles_per_object_flags():
    """scripts/build_variant.sh / build_variant2.sh rebuild ONE translation unit for A/B runs.  In r05 they lacked conv_wino4.o's
    -fno-slp-vectorize: every variant of that kernel was an SLP-packed build, 0.3-0.5 ms/step slower than the tree whatever it changed
    (profiles/r05/x_wino4_variants.txt).  Every `obj.o [obj.o ...]: FLAGS += ...` line of the Makefile must be mirrored in both scripts."""
    import re
    mk = open(os.path.join(ROOT, "instancediff_amd", "csrc", "Makefile")).read()
    per_obj = {}
    for objs, flags in re.findall(r"^([\w. ]+\.o)\s*:\s*FLAGS\s*\+=\s*(.+)$", mk, re.M):
        for o in objs.split():
            per_obj.setdefault(o, []).extend(flags.split())
    assert per_obj.get("conv_wino4.o") == ["-fno-slp-vectorize"] and "sde.o" in per_obj, per_obj
    for script in ("build_variant.sh", "build_variant2.sh"):
        txt = open(os.path.join(ROOT, "scripts", script)).read()
        cases = {}
        for objs, flag in re.findall(r"([\w.|]+\.o)\)\s*EXTRA=\"([^\"]+)\"", txt):
            for o in objs.split("|"):
                cases.setdefault(o, []).extend(flag.split())
        assert cases == per_obj, (script, cases, per_obj)
