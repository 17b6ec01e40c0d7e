#!/usr/bin/env python3
"""Headline benchmark: denoising steps/sec of the InstanceDiff sampling loop on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank/GPU)

Workload (BASELINE.json configs[1]): 256x256 1-channel synthetic batch of 16, T=1000 reverse chain; one STEP =
one reverse-loop iteration of CLIPDriftModel.test()/driftSDE.reverse_ddpm = 2 UNet forwards (drift_net +
noise_net, each with its 4 ScoreMapModules and single-token image-context cross-attention) + the fused reverse
update, fp32, random-init weights (seed 0), inputs resident in HBM, on-device Philox noise.  Multi-GPU = N
independent replicas (sampling shards by image, no data-path collective) -> weak scaling.

Prints ONE JSON line: metric/value/... plus
  roofline     : dominant kernel = the Winograd F(2x2,3x3) conv on the f32 matrix cores; ALGORITHMIC (direct-form) FLOP per
                 launch / average launch duration (HIP events on the launch stream, second single-stream pass over the
                 same K steps), against the 157.3 TFLOP/s f32-MFMA peak (MI355X_MICROARCH.md).  The kernel executes
                 16/36 of the algorithmic multiply-adds, so `frac` can exceed 1; `executed_frac` is the matrix-pipe share.
  cpu_baseline : the oracle (plain PyTorch fp32 restatement of the same step) on the host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=1)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--mode", choices=["sample", "train", "irsde"], default="sample",
                    help="sample = headline denoising-steps/s metric (driftSDE: 2 UNet forwards + update per step); train = secondary "
                         "training-iterations/s line; irsde = secondary line for the IRSDE single-network loop (1 UNet forward + reverse_sde_step)")
    return ap.parse_args()


class StepRunner:
    """The product's loop body -- driftSDE.Stepper, the object reverse_ddpm itself runs: two UNet forwards on two HIP streams, the
    fused update and the device-side state advance, replayed as one captured HIP graph per step (IDIFF_HIP_GRAPH=0: eager)."""

    def __init__(self, model, sde, batch):
        from instancediff_amd import ops
        from instancediff_amd.models.SDEs.driftSDE import driftSDE
        self.ops, self.model, self.sde = ops, model, sde
        dev = model.device
        cond = batch['input'].to(dev).contiguous()
        ctx = batch['A_emb'].to(dev).contiguous()
        sde.set_seed(4321)
        x = ops.axpby(cond, sde._randn_like(cond), 1.0, sde.max_sigma)
        self.stepper = driftSDE.Stepper(sde, x, cond, batch['names'], model.text_encoder, ctx)

    @property
    def x(self):
        return self.stepper.x

    def prepare(self):
        return self.stepper.prepare()

    def run(self, n):
        return self.stepper.run(n)

    @torch.no_grad()
    def step(self):
        """one eager step (profiling passes)"""
        self.stepper._body()


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box exposes
    every host core to os.cpu_count() but grants a share; oversubscribing it makes the oracle crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("IDIFF_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, min(n, 64))


def cpu_baseline(args, batch_full):
    """Oracle step (2 oracle UNet forwards + oracle reverse update) on the host cores, on a bounded sample of the
    same workload: the first `cpu_sample_batch` images of the batch; steps/s is scaled to the full batch."""
    import torch.nn as nn
    from oracle import sde_ref, unet_ref
    from instancediff_amd import pipeline
    ncores = host_cores()
    torch.set_num_threads(ncores)
    bs = min(args.cpu_sample_batch, args.batch)
    opt = pipeline.load_options()
    mo = opt['models']['DriftNoise']
    nets = []
    torch.manual_seed(0)
    for key in ('dnet_settings', 'nnet_settings'):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m) for m in mo['score_map_ch_mult']])
        nets.append(unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s).eval())
    te = unet_ref.StubTextEncoder()
    sde = sde_ref.DriftSDERef(args.T, nets[0], nets[1], max_sigma=0.4)
    cond = batch_full['input'][:bs]
    names = batch_full['names'][:bs]
    ctx = batch_full['A_emb'][:bs]
    g = torch.Generator().manual_seed(4321)
    x = cond + 0.4 * torch.randn(cond.shape, generator=g)

    def one(x, t):
        tt = torch.full((bs,), t, dtype=torch.long)
        with torch.no_grad():
            rd = nets[0](x - cond, cond, tt, names, te, image_context=ctx)[0]
            rn = nets[1](x - cond, x, tt, names, te, image_context=ctx)[0]
        z = torch.randn(cond.shape, generator=g)
        return sde_ref.drift_reverse_update(x, rd, rn, z, sde.a[t], sde.b[t], sde.c[t])

    log("cpu oracle built, warm step (B=%d)" % bs)
    x = one(x, args.T)  # warm
    log("cpu warm step done")
    n, t0 = 0, time.time()
    while True:
        x = one(x, args.T - 1 - n)
        n += 1
        el = time.time() - t0
        if el > 12.0 or n >= 3:
            break
    sps_sample = n / el
    return {"value": sps_sample * bs / args.batch, "unit": "denoising steps/s (batch %d)" % args.batch, "cores": ncores, "kind": "port",
            "sample": "%d timed step(s) of the oracle on the first %d of %d images (%.2f s/step at B=%d), scaled by %d/%d; torch %d threads"
                      % (n, bs, args.batch, el / n, bs, bs, args.batch, ncores)}


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def train_bench(args, world, rank, dev):
    """Secondary line (BASELINE config c3): training iterations/sec of CLIPDriftModel.optimize_parameters --
    feed_data (forward diffusion) + 2 UNet forwards + losses + backward + flat RCCL all-reduce + fused Adam."""
    import torch.distributed as dist
    from instancediff_amd import pipeline
    from instancediff_amd.utils.synthetic import make_batch
    T = 100 if args.T == 1000 else args.T
    model, sde = pipeline.build(phase="train", device=dev, T=T, seed=0, dist=world > 1)
    model.set_train()
    sde.set_seed(1234 + rank)
    batch = make_batch(args.batch, args.size, seed=1234 + rank, mixed=True)
    torch.manual_seed(99 + rank)

    def it():
        model.feed_data(batch)
        return model.optimize_parameters()[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("train: model built; warmup")
    for _ in range(args.warmup):
        loss = it()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = it()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    if rank == 0:
        value = world * args.steps / el
        print(json.dumps({"metric": "training iterations/sec (%dx%d bs%d/GPU)" % (args.size, args.size, args.batch), "value": round(value, 4),
                          "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic", "last_loss": loss,
                          "config": {"workload": "%dx%d synthetic, batch %d per GPU, drift+noise UNet fwd/bwd, pyramid losses, Adam"
                                                 % (args.size, args.size, args.batch), "global_batch": args.batch * world,
                                     "parallelism": "dp%d (flat RCCL all-reduce)" % world,
                                     "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def irsde_bench(args, world, rank, dev):
    """IRSDE single-network mode (utils/sde_utils.py:244-261): one step = noise_net forward (as IRSDE's `model(x, mu, t)`) +
    the fused reverse_sde_step; same synthetic batch, T and batch as the headline line.  Secondary line."""
    import torch.distributed as dist
    from instancediff_amd import pipeline
    from instancediff_amd.utils.sde_utils import IRSDE
    from instancediff_amd.utils.synthetic import make_batch
    model, _ = pipeline.build(phase="test", device=dev, T=args.T, seed=0)
    model.set_eval()
    batch = make_batch(args.batch, args.size, seed=1234 + rank, mixed=True)
    lq = batch['input'].to(dev).contiguous()
    ctx = batch['A_emb'].to(dev).contiguous()
    sde = IRSDE(max_sigma=0.4, T=args.T, schedule='cosine', device=dev)
    sde.set_mu(lq)
    net = model.noise_net
    sde.set_model(lambda x, mu, t, **kw: net(x, mu, torch.full((x.shape[0],), float(t), device=dev), batch['names'], model.text_encoder,
                                             image_context=ctx))
    state = {"x": sde.noise_state(lq), "t": args.T}

    @torch.no_grad()
    def step():
        t = state["t"]
        noise = sde.noise_fn(state["x"], t, sde.sample_scale)
        noise = noise[0] if isinstance(noise, tuple) else noise
        state["x"] = sde.reverse_sde_step_from_noise(state["x"], noise, t)
        state["t"] = t - 1 if t > 1 else args.T

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    assert torch.isfinite(state["x"]).all()
    if rank == 0:
        value = world * args.steps / el
        print(json.dumps({"metric": "IRSDE denoising steps/sec (%dx%d bs%d)" % (args.size, args.size, args.batch), "value": round(value, 4),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "%dx%d 1-ch synthetic, IRSDE reverse_sde, batch %d per GPU, 1 UNet fwd + reverse_sde_step per step"
                                                 % (args.size, args.size, args.batch),
                                     "global_batch": args.batch * world, "parallelism": "replicas x%d (no collective)" % world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # IDIFF_BENCH_REHEARSAL=1: several ranks share the one GPU of a development box over gloo -- exercises the N>1 code path
    # (rendezvous, barriers, max-over-ranks timing, rank-0 reporting); its numbers mean nothing.
    rehearsal = bool(int(os.environ.get("IDIFF_BENCH_REHEARSAL", "0")))
    if rehearsal:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from instancediff_amd import ops, pipeline
    from instancediff_amd.utils.synthetic import make_batch

    if args.mode == "train":
        return train_bench(args, world, rank, dev)
    if args.mode == "irsde":
        return irsde_bench(args, world, rank, dev)

    model, sde = pipeline.build(phase="test", device=dev, T=args.T, seed=0)
    model.set_eval()
    batch = make_batch(args.batch, args.size, seed=1234 + rank, mixed=True)
    run = StepRunner(model, sde, batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    log("model built; graph capture + warmup")
    run.prepare()  # one eager step + HIP-graph capture of the step (untimed)
    if args.warmup:
        run.run(args.warmup)
    barrier()
    log("warmup done; timing %d steps" % args.steps)
    t0 = time.perf_counter()
    run.run(args.steps)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    assert torch.isfinite(run.x).all(), "non-finite state after the timed steps"
    log("timed region: %.3f s for %d steps" % (el, args.steps))

    roof = None
    if rank == 0 and not args.no_roofline:
        # second pass over the same K steps, on ONE stream (the timed region overlaps the two nets on two streams, which
        # would smear per-launch event times), every conv launch bracketed by HIP events on its launch stream
        two = sde.two_streams
        sde.two_streams = False
        ops.PROFILE = []
        for _ in range(args.steps):
            run.step()  # eager: per-launch events cannot be recorded inside a graph replay
        torch.cuda.synchronize()
        sde.two_streams = two
        # dominant kernel = conv_wino_kernel: every 3x3 conv whose shape tiles exactly (idiff_conv2d_last_algo() == 1)
        recs = [r for r in ops.PROFILE if r['algo'] == 1]
        allrecs = ops.PROFILE
        ops.PROFILE = None
        tot_ms = sum(r['e0'].elapsed_time(r['e1']) for r in recs)
        tot_fl = sum(r['flops'] for r in recs)
        all_ms = sum(r['e0'].elapsed_time(r['e1']) for r in allrecs)
        all_fl = sum(r['flops'] for r in allrecs)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_wino_kernel (3x3 conv, Winograd F(2x2,3x3) on f32 MFMA)", "achieved": round(ach, 2),
                "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "flop_count": "algorithmic = direct-form 2*Cin*Cout*9*H*W*B per launch; the kernel executes 16/36 of them",
                "executed_tflops": round(ach * 16 / 36, 2), "executed_frac": round(ach * 16 / 36 / F32_MFMA_PEAK_TFLOPS, 4),
                "launches": len(recs), "avg_launch_ms": round(tot_ms / max(len(recs), 1), 4),
                "gflop_per_launch": round(tot_fl / max(len(recs), 1) / 1e9, 3),
                "conv_ms_per_step_single_stream": round(all_ms / args.steps, 3),
                "wino_ms_per_step_single_stream": round(tot_ms / args.steps, 3),
                "all_conv_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2),
                "step_conv_gflop": round(all_fl / args.steps / 1e9, 1)}
        try:  # HBM bytes per launch of the same kernel from a separate rocprofv3 --pmc run of this command (profiles/)
            with open(os.path.join(ROOT, "profiles", "r01", "f_pmc_traffic.json")) as f:
                pmc = json.load(f)
            roof["traffic"] = pmc["hbm_bytes_per_launch"]
            roof["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, profiles/r01/f_pmc_traffic.json)"
        except (OSError, KeyError, ValueError):
            pass
    if world > 1:
        barrier()

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only: the other ranks of a multi-GPU run must not wait on it
            log("cpu baseline (oracle) ...")
            try:
                cpu = cpu_baseline(args, batch)
            except Exception as e:  # the baseline is a reported side measurement; never hide the GPU result
                cpu = {"error": repr(e)}
        value = world * args.steps / el
        line = {"metric": "denoising steps/sec (%dx%d bs%d)" % (args.size, args.size, args.batch), "value": round(value, 4),
                "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%dx%d 1-ch synthetic, %d-step reverse chain, batch %d per GPU, 2 UNet fwd + reverse update per step"
                                       % (args.size, args.size, args.T, args.batch),
                           "global_batch": args.batch * world, "parallelism": "replicas x%d (no collective)" % world,
                           "image_steps_per_s": round(value * args.batch, 2)},
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
