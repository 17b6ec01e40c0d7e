"""The data-parallel path on the real RCCL backend ("nccl" on ROCm), at world size 1 on the one GPU of the test box: communicator
creation, broadcast, the flat-buffer all-reduces (fp32 and bf16 wire) started asynchronously during the backward of a full
training step, and a HIP-graph-captured sampling loop running while the process group is alive (RCCL's watchdog thread must not
invalidate the capture: driftSDE captures in thread-local mode).  Multi-rank semantics are covered on CPU with gloo at world size 2
(tests/test_host_cpu.py); the scaling curve itself is measured by the driver's 8-GPU run."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from instancediff_amd import ops, pipeline, train_ops  # noqa: E402
from instancediff_amd.parallel import GradSync  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def rccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def _train_inputs(model, sde, batch, t, eps):
    model.input = batch['input'].to(DEV)
    model.target = batch['target'].to(DEV)
    model.names = list(batch['names'])
    model.A_emb = batch['A_emb'].to(DEV)
    model.t, model.drift_noised_x, _, model.std_noise, _ = sde.forward_diffusion(model.target, model.input, t=t, eps=eps.to(DEV))


def _grads(model):
    return torch.cat([g.reshape(-1) for g in model.drift_optimizer.flat_grads() + model.noise_optimizer.flat_grads()]).clone()


def test_rccl_collectives_on_flat_buffers(rccl_world1):
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    gs = GradSync(single_rank_collectives=True)
    assert gs.active
    a = torch.arange(1 << 20, dtype=torch.float32, device=DEV) * 1e-3
    b = torch.randn(12345, device=DEV)
    a0, b0 = a.clone(), b.clone()
    assert gs.all_reduce_flat([a, b]) == 1.0
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)  # SUM over one rank
    p = torch.nn.Parameter(torch.randn(1000, device=DEV))
    p0 = p.detach().clone()
    gs.broadcast_parameters([p])
    torch.cuda.synchronize()
    assert torch.equal(p.detach(), p0)
    # bf16 wire: pack (round to nearest even) -> all-reduce in bf16 -> widen back into the fp32 master buffer
    gsb = GradSync(single_rank_collectives=True, wire="bf16")
    c = torch.randn(100003, device=DEV) * 3
    want = c.to(torch.bfloat16).to(torch.float32)
    gsb.all_reduce_flat([c])
    torch.cuda.synchronize()
    assert torch.equal(c, want)
    assert torch.equal(ops.bf16_to_f32(ops.f32_to_bf16(b0)), b0.to(torch.bfloat16).to(torch.float32))


def test_train_step_with_rccl_exchange_overlapped(rccl_world1):
    B, H, T_ = 2, 32, 20
    batch = make_batch(B, H, seed=3)
    g = torch.Generator().manual_seed(7)
    t = torch.tensor([[[[5]]], [[[17]]]])
    eps = torch.randn(batch['input'].shape, generator=g)
    ref_model, ref_sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0, score_map_dropout=0.0)
    ref_model.set_train()
    _train_inputs(ref_model, ref_sde, batch, t, eps)
    train_ops.forward_backward_inputRes(ref_model)
    ref = _grads(ref_model)
    for wire in ("fp32", "bf16"):
        model, sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0, dist=True, score_map_dropout=0.0)
        assert model.grad_sync is not None  # built because a process group is alive
        model.grad_sync = GradSync(single_rank_collectives=True, wire=wire)
        model.grad_sync.broadcast_parameters(list(model.drift_net.parameters()) + list(model.noise_net.parameters()))
        model.set_train()
        _train_inputs(model, sde, batch, t, eps)
        train_ops.forward_backward_inputRes(model)   # starts the drift net's exchange under the noise net's backward
        assert len(model.grad_sync._pending) == 2
        assert model.grad_sync.finish() == 1.0
        got = _grads(model)
        torch.cuda.synchronize()
        if wire == "fp32":
            assert torch.equal(got, ref)
        else:
            assert torch.equal(got, ref.to(torch.bfloat16).to(torch.float32))
        _train_inputs(model, sde, batch, t, eps)
        loss, _ = model.optimize_parameters()  # the full step: backward + started exchanges + finish + fused Adam
        assert loss == loss and loss > 0


def test_graph_captured_sampling_with_process_group_alive(rccl_world1):
    T, B, H = 6, 2, 32
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=3)
    model.set_eval()
    batch = make_batch(B, H, seed=11)
    tok = torch.ones(8, device=DEV)
    dist.all_reduce(tok)  # make sure the communicator (and its watchdog thread) is up before the capture
    outs = []
    for use_graph in (True, False):
        sde.hip_graph = use_graph
        sde.set_seed(99)
        model.feed_data(batch)
        sde.set_seed(99)
        model.test()
        assert sde.last_mode == ("graph" if use_graph else "eager"), sde.last_mode
        outs.append(torch.from_numpy(model.get_visuals()).clone())
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
