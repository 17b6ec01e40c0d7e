#!/usr/bin/env python3
"""Headline benchmark: denoising steps/sec of the InstanceDiff sampling loop on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) the process IS
a rank; started plainly (`python bench.py --gpus N`, the reference's one-command launch: README.md:35, trainUM.py:50-66) the parent
-- before anything touches the GPU -- starts N fresh rank processes of itself with that environment, waits for them, and exits
non-zero if any rank failed (spawn_ranks below; never os.exec*).  n_gpus in the line is the world size the process group reports.

Workload (BASELINE.json configs[1]): 256x256 1-channel synthetic batch of 16, T=1000 reverse chain; one STEP =
one reverse-loop iteration of CLIPDriftModel.test()/driftSDE.reverse_ddpm = 2 UNet forwards (drift_net +
noise_net, each with its 4 ScoreMapModules and single-token image-context cross-attention) + the fused reverse
update, fp32, random-init weights (seed 0), inputs resident in HBM, on-device Philox noise.  Multi-GPU = N
independent replicas (sampling shards by image, no data-path collective) -> weak scaling.

Prints ONE JSON line: metric/value/... plus
  roofline     : dominant kernel = the Winograd conv kernel with the largest share of the step (`kernels` lists all three: the
                 F(4x4,3x3) kernels conv_wino4_kernel / conv_wino4h_kernel and the F(2x2,3x3) conv_wino_kernel).  `achieved` = the matrix-core FLOP the kernel's algorithm EXECUTES per
                 launch (36 multiply-adds per 4x4 output tile and channel pair = 1/4 of the direct form; F(2x2,3x3): 16/36) /
                 average launch duration (HIP events on the launch stream, a second single-stream eager pass over the same K
                 steps); `peak` = 157.3 TFLOP/s f32 MFMA (MI355X_MICROARCH.md); `frac` = achieved / peak = the share of the matrix
                 pipe in use.  `effective_tflops` is the direct-form rate (x `algorithmic_speedup`), `step_frac` the executed
                 conv FLOP of a whole step / ms_per_step / peak.  `traffic`, `hbm_kernels` (every HBM-side kernel with >= 2 % of the
                 step: bytes per launch, GB/s, fraction of 8 TB/s) and `step_hbm_bytes` come from rocprofv3 --pmc passes stored
                 under profiles/ with a hash of the kernel sources; a stale file (sources changed since) is not reported.
  cpu_baseline : the oracle (plain PyTorch fp32 restatement of the same step) on the host cores, at the named batch when one
                 step fits the time budget (else a stated fraction of the batch), 1 warm + >= 1 timed step.
  graph / launches_per_step : whether the timed loop replayed a captured HIP graph, and how many kernels one step enqueues.
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--cpu-sample-batch", type=int, default=0, help="images per CPU-baseline step (0 = the full batch when one step "
                    "is estimated to fit --cpu-budget-s, else 4)")
    ap.add_argument("--cpu-budget-s", type=float, default=140.0, help="time budget of the CPU-baseline leg (warm + timed steps)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the `train` key of the default line (BASELINE c3 at N=1, ~40 s)")
    ap.add_argument("--attn", choices=["f32", "bf16", "f16"], default="f32", help="bf16 / f16 = REDUCED-PRECISION VARIANT line (self-attention "
                    "contractions on the bf16 / fp16 matrix cores; f16 with operands clamped to +-255 is the reference's own form, BASELINE "
                    "config c5): own metric label, PSNR delta vs the fp32 path stated")
    ap.add_argument("--grad-wire", choices=["fp32", "bf16"], default="fp32", help="--mode train: wire format of the gradient all-reduce")
    ap.add_argument("--mode", choices=["sample", "train", "irsde"], default="sample",
                    help="sample = headline denoising-steps/s metric (driftSDE: 2 UNet forwards + update per step); train = secondary "
                         "training-iterations/s line; irsde = secondary line for the IRSDE single-network loop (1 UNet forward + reverse_sde_step)")
    return ap.parse_args()


def spawn_ranks(n, cmd, env=None, poll_s=0.2):
    """Start `n` rank processes of `cmd` (a list), rank r with RANK = LOCAL_RANK = r, WORLD_SIZE = n and a rendezvous on
    127.0.0.1 at a free port; stdout / stderr are inherited (rank 0 prints the JSON line).  Returns the exit code: 0 when every
    rank exited 0; otherwise the first failing rank's code, after the ranks still running (which would wait for it at the next
    barrier forever) have been terminated -- by the exact PIDs started here.  The caller must not have initialised the GPU."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r))))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(poll_s)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    print("[bench] rank %d exited with code %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr, flush=True)
                    for q in live:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return rc


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
    """Oracle step (2 oracle UNet forwards + oracle reverse update) on the host cores: 1 warm + >= 1 timed step at the NAMED batch
    when that fits the budget; otherwise on the first 4 images, with the measured B=1 -> B=4 scaling stated next to it."""
    import torch.nn as nn
    from oracle import sde_ref, unet_ref
    from instancediff_amd import pipeline
    ncores = host_cores()
    torch.set_num_threads(ncores)
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
    g = torch.Generator().manual_seed(4321)

    def run(bs, nsteps, budget):
        """(seconds per step, steps timed) at batch bs: one warm step, then up to nsteps timed steps within the budget"""
        cond, names, ctx = batch_full['input'][:bs], batch_full['names'][:bs], batch_full['A_emb'][:bs]
        x = cond + 0.4 * torch.randn(cond.shape, generator=g)

        def one(x, t):
            tt = torch.full((bs,), t, dtype=torch.long)
            with torch.no_grad():
                rd = nets[0](x - cond, cond, tt, names, te, image_context=ctx)[0]
                rn = nets[1](x - cond, x, tt, names, te, image_context=ctx)[0]
            z = torch.randn(cond.shape, generator=g)
            return sde_ref.drift_reverse_update(x, rd, rn, z, sde.a[t], sde.b[t], sde.c[t])

        x = one(x, args.T)  # warm
        n, t0, per = 0, time.time(), []
        while True:
            ts = time.time()
            x = one(x, args.T - 1 - n)
            per.append(time.time() - ts)
            n += 1
            el = time.time() - t0
            if n >= nsteps or el + el / n > budget:
                break
        return el / n, n, per

    t_start = time.time()
    log("cpu oracle built (%d threads); probing B=1" % ncores)
    s1, _, _ = run(1, 1, 30.0)
    bs = args.cpu_sample_batch or args.batch
    est = s1 * bs * 3.0  # warm + two timed steps at linear scaling (batch-16 convs thread better than B=1: an upper estimate)
    if not args.cpu_sample_batch and est > args.cpu_budget_s:
        bs = min(4, args.batch)
    log("cpu: %.2f s/step at B=1; timing B=%d" % (s1, bs))
    sec, n, per = run(bs, 3, max(args.cpu_budget_s - (time.time() - t_start), 5.0))
    sps = 1.0 / sec * bs / args.batch
    sample = ("1 warm + %d timed step(s) of the oracle at batch %d of %d (mean %.2f s/step, min %.2f, max %.2f)"
              % (n, bs, args.batch, sec, min(per), max(per)))
    if bs < args.batch:
        sample += ", scaled by %d/%d" % (bs, args.batch)
    sample += "; B=1 probe %.2f s/step (B-scaling %.2fx per image vs B=1); torch %d threads" % (s1, (sec / bs) / s1, ncores)
    return {"value": round(sps, 5), "unit": "denoising steps/s (batch %d)" % args.batch, "cores": ncores, "kind": "port",
            "batch_timed": bs, "steps_timed": n, "s_per_step": [round(v, 3) for v in per], "sample": sample}


# translation units none of whose kernels a sampling step launches (profiles/r*/g_steady_state_step_single_stream.txt lists them all):
# the weight-gradient / backward / batched-GEMM kernels of the training path, the PSNR metrics, and the fp16 form of the ScoreMapModule
# decoder attentions (model option score_map_if_flash: a labelled variant, never in the default step)
TRAIN_ONLY_SOURCES = ("backward.hip", "conv_wgrad.hip", "conv_wgrad_args.h", "conv_wino_wgrad.hip", "conv_wino4_wgrad.hip", "gemm.hip", "metrics.hip",
                      "attention_flash.hip")


def kernel_source_hash():
    """sha256 over the HIP sources + headers behind the sampling step's kernels: the staleness key of stored PMC traffic figures"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "instancediff_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "instancediff_amd", "csrc", "*.h"))):
        if os.path.basename(path) in TRAIN_ONLY_SOURCES:
            continue
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def matrix_gflop_per_launch(kernel, B=16, size=256, heads=4, K=5):
    """Executed f32 matrix-core GFLOP per launch of the attention kernels at the default workload (None for other kernels): 4096 FLOP per
    v_mfma_f32_32x32x2.  ScoreMapModule cross-attention, per 32-key block: the wave-per-block form (72 rows) issues 36 + 48 MFMAs, the
    channel-split forms 4 waves x (XCW/2 + 16 * ceil(XCW/32)); the grouped launch covers the four levels.  Mid self-attention (4 heads x 64
    at 32x32 tokens): 4 N^2 dh heads B FLOP."""
    if kernel.startswith("smm_xattn_grouped_kernel"):
        mf = 0
        for lvl, C in enumerate((64, 64, 128, 256)):
            blocks = B * ((size >> lvl) ** 2) // 32
            Cm = 72 if C + 1 <= 72 else (136 if C + 1 <= 136 else 256)
            xcw = Cm // 4
            mf += blocks * (84 if Cm == 72 else 4 * (xcw // 2 + 16 * ((xcw + 31) // 32)))
        return mf * 4096 / 1e9
    if kernel.startswith("attn_self_kernel"):
        n = (size // 8) ** 2
        return 4.0 * n * n * 64 * heads * B / 1e9
    return None


def pmc_evidence(kernel):
    """Counter evidence from the newest profiles/r*/pmc_kernels.json (scripts/collect_pmc_table.sh: counter-free trace + separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command's single-stream eager step, reduced per kernel by
    scripts/pmc_table.py): `traffic` of the dominant kernel (launch-weighted mean over its template instantiations), the HBM-side
    kernels of the step with their measured bound, and the step's HBM bytes.  Reported only if measured on the same kernel sources."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_kernels.json")))
    if not cands:
        return {"traffic": None, "traffic_note": "no PMC pass stored"}
    src = os.path.relpath(cands[-1], ROOT)
    try:
        with open(cands[-1]) as f:
            doc = json.load(f)
        if doc.get("kernel_source_hash") != kernel_source_hash():
            return {"traffic": None, "traffic_note": "stored PMC passes (%s) are stale: kernel sources changed since (%s != %s)"
                    % (src, doc.get("kernel_source_hash"), kernel_source_hash())}
        short = kernel.split(" ")[0]
        mine = [k for k in doc["kernels"] if k["kernel"].startswith(short + "<") and "hbm_bytes_per_launch" in k]
        out = {"traffic_unit": "bytes/launch (%s)" % doc.get("formula", "2*FETCH_SIZE + WRITE_SIZE"), "traffic_source": src, "traffic_from_profile": True}
        if mine:
            n = sum(k["launches_per_step"] for k in mine)
            out["traffic"] = int(sum(k["hbm_bytes_per_launch"] * k["launches_per_step"] for k in mine) / n)
            # the profile's own clock for the same kernel (rocprofv3 kernel trace), next to this run's HIP-event avg_launch_ms
            out["profile_avg_launch_ms"] = round(sum(k["avg_us"] * k["launches_per_step"] for k in mine) / n / 1e3, 4)
            out["traffic_GBps"] = round(sum(k["hbm_bytes_per_launch"] * k["launches_per_step"] for k in mine) /
                                        sum(k["avg_us"] * 1e3 * k["launches_per_step"] for k in mine), 1)
        else:
            out["traffic"] = None
            out["traffic_note"] = "the stored PMC passes (%s) hold no launch of %s" % (src, short)
        # every kernel with >= 2 % of the step's kernel time that is not a 3x3 Winograd conv: measured HBM bound
        out["hbm_kernels"] = [{"kernel": k["kernel"], "bound": "hbm", "launches_per_step": k["launches_per_step"], "avg_us": k["avg_us"],
                               "share_of_step_kernel_time": k["share_of_step_kernel_time"], "bytes_per_launch": k["hbm_bytes_per_launch"],
                               "achieved": k["achieved_GBps"], "peak": k["peak_GBps"], "unit": "GB/s", "frac": k["frac"]}
                              for k in doc["kernels"] if "frac" in k and not k["kernel"].startswith("conv_wino") and k["share_of_step_kernel_time"] >= 0.02]
        # ... of which the attention kernels are bound by the f32 matrix rate, not by HBM: their executed matrix-core FLOP per launch at the
        # default workload (256x256, batch 16; query rows padded to the 32-row tile) against the 157.3 TFLOP/s peak
        for k in out["hbm_kernels"]:
            gf = matrix_gflop_per_launch(k["kernel"])
            if gf is not None:
                tf = gf / k["avg_us"] / 1e-3
                k.update(bound="mfma", executed_gflop_per_launch=round(gf, 2), achieved_tflops=round(tf, 1), peak_tflops=F32_MFMA_PEAK_TFLOPS,
                         frac_mfma=round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                         note="f32-MFMA-bound (20 query rows on 32-row tiles); `frac` / `achieved` = its HBM traffic, for reference")
        # matrix-pipe utilisation of the dominant kernel per layer shape: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), its own
        # --pmc passes (scripts/collect_mfma_util.sh), stored next to the traffic table and hash-gated like it
        mu = os.path.join(os.path.dirname(cands[-1]), "mfma_util.json")
        if os.path.exists(mu):
            with open(mu) as f:
                mdoc = json.load(f)
            if mdoc.get("kernel_source_hash") == kernel_source_hash() and short.startswith(mdoc.get("kernel", "?")):
                out["mfma_util_measured"] = {k: v["mfma_util"] for k, v in mdoc["shapes"].items()}
                out["mfma_util_source"] = os.path.relpath(mu, ROOT)
        out["step_hbm_bytes"] = doc.get("step_hbm_bytes")
        out["step_hbm_GBps_over_kernel_time"] = doc.get("step_hbm_GBps_over_kernel_time")
        return out
    except (OSError, KeyError, ValueError, ZeroDivisionError) as e:
        return {"traffic": None, "traffic_note": repr(e)}


def attention_variant_delta(args, dev, steps=8):
    """|dPSNR| and max|diff| of the bf16-attention variant against the fp32 path: the same `steps`-step chain (same weights,
    inputs, Philox seed) run once with each, at the benchmark's own size (batch 2)."""
    from instancediff_amd import ops, pipeline
    from instancediff_amd.utils.synthetic import make_batch
    from oracle import sde_ref  # PSNR helper only (the checker's formula, trainUM.py:319-323 semantics)
    outs = []
    batch = make_batch(2, args.size, seed=99, mixed=True)
    for dt in ("f32", args.attn):
        ops.ATTN_DTYPE = dt
        model, sde = pipeline.build(phase="test", device=dev, T=steps, seed=0)
        model.set_eval()
        sde.set_seed(7)
        model.feed_data(batch)
        sde.set_seed(7)
        model.test()
        outs.append(torch.from_numpy(model.get_visuals()).clone())
    ops.ATTN_DTYPE = "f32"
    d = abs(sde_ref.psnr(outs[0], batch['target']) - sde_ref.psnr(outs[1], batch['target']))
    return {"variant": "self-attention contractions on %s MFMA (fp32 softmax and accumulation); everything else fp32" % args.attn,
            "psnr_delta_db_vs_fp32_path": float("%.3g" % d), "max_abs_diff_vs_fp32_path": float("%.3g" % float((outs[0] - outs[1]).abs().max())),
            "measured_on": "%d-step chain, batch 2, %dx%d, same weights / inputs / noise" % (steps, args.size, args.size)}


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def train_roofline(it, iters=2):
    """Per-kernel-class roofline of the training iteration, measured live: `iters` more iterations on ONE stream with HIP events on the
    launch stream around every library call of a matrix-core kernel class (ops.PROFILE hooks in ops.conv2d and train_ops), the FLOP
    each call EXECUTES (F(4x4,3x3) forms: 36/144 of the direct 2*9*Cin*Cout*H*W*B, F(2x2,3x3): 16/36, the fused ScoreMapModule
    attention with its query rows padded to the 32-row tile) / its time / the 157.3 TFLOP/s f32 matrix peak; then one iteration under
    torch.profiler to count the device launches that are not this library's (ATen / runtime copies)."""
    from instancediff_amd import ops, train_ops
    two = train_ops.TRAIN_TWO_STREAMS
    train_ops.TRAIN_TWO_STREAMS = False
    try:
        it()
        torch.cuda.synchronize()
        ops.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t0) / iters * 1e3
        recs, ops.PROFILE = ops.PROFILE, None
    finally:
        ops.PROFILE = None
        train_ops.TRAIN_TWO_STREAMS = two
    CONV = {0: ("conv_igemm_kernel (direct form)", 1.0), 1: ("conv_wino_kernel F(2x2,3x3)", 16.0 / 36.0), 2: ("conv1x1 streaming f32", 1.0),
            3: ("conv_wino4_kernel F(4x4,3x3) forward + data gradient", 0.25), 4: ("conv_wino4h_kernel F(4x4,3x3) forward + data gradient", 0.25),
            5: ("conv1x1_x3_kernel (bf16x3 split; fp32-equivalent flops)", 1.0)}
    WG = {0: ("conv_wgrad_kernel (direct form)", 1.0), 1: ("wino_wgrad_kernel F(2x2,3x3) weight gradient", 16.0 / 36.0),
          2: ("wgrad1x1_kernel (streaming 1x1 weight gradient)", 1.0), 3: ("wino4_wgrad_kernel F(4x4,3x3) weight gradient", 0.25)}
    groups = {}
    for r in recs:
        kind = r.get("kind", "conv")
        if kind == "conv":
            name, f = CONV.get(r["algo"], ("conv algo %d" % r["algo"], 1.0))
        elif kind == "wgrad":
            name, f = WG.get(r.get("algo", 0), ("wgrad algo %s" % r.get("algo"), 1.0))
        else:
            name, f = {"bgemm": "bgemm_kernel (token-side batched GEMMs)", "smm_xattn_fwd": "smm_xattn_kernel<64> (ScoreMapModule attention forward)",
                       "smm_xattn_bwd": "smm_xattn_bwd_kernel (ScoreMapModule attention backward)"}[kind], 1.0
        g = groups.setdefault(name, {"n": 0, "ms": 0.0, "fl": 0.0})
        g["n"] += 1
        g["ms"] += r["e0"].elapsed_time(r["e1"])
        g["fl"] += r["flops"] * f
    rows = []
    for name, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
        tf = g["fl"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        rows.append({"kernel": name, "launches_per_it": g["n"] / iters, "ms_per_it_single_stream": round(g["ms"] / iters, 2),
                     "executed_gflop_per_it": round(g["fl"] / iters / 1e9, 1), "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4)})
    timed_ms = sum(g["ms"] for g in groups.values()) / iters
    exec_fl = sum(g["fl"] for g in groups.values()) / iters
    out = {"bound": "mfma", "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "kernels": rows[:5], "other_kernel_classes": rows[5:],
           "ms_per_it_single_stream_wall": round(single_ms, 2), "matrix_kernel_ms_per_it_single_stream": round(timed_ms, 2),
           "executed_gflop_per_it": round(exec_fl / 1e9, 1),
           "how": "HIP events on the launch stream around each library call, %d single-stream iterations after the timed ones" % iters}
    # launches outside the library: one iteration under torch.profiler; a kernel launched by an ATen op (or a runtime copy) hangs on
    # the op's CPU event (ev.kernels); the library's own launches go through ctypes and are counted by idiff_launch_count()
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            it()
            torch.cuda.synchronize()
        aten, names = 0, {}
        for ev in prof.events():
            ks = getattr(ev, "kernels", None) or []
            if not ks or not ev.name.startswith("aten::"):
                continue
            aten += len(ks)
            names[ev.name] = names.get(ev.name, 0) + len(ks)
        out["aten_launches_per_it"] = aten
        out["aten_ops"] = dict(sorted(names.items(), key=lambda kv: -kv[1])[:8])
        out["aten_note"] = ("what is left is the frozen context text encoder (a host-side torch module handed to the nets as a forward argument, "
                            "SURVEY 8 row f1: mm / mean / sum), the mid self-attention's q|k|v slicing and the host->device copies of feed_data")
    except Exception as e:  # a profiler that cannot attach (e.g. under rocprofv3) must not hide the line
        out["aten_launches_per_it"] = None
        out["aten_note"] = "torch.profiler unavailable: %r" % (e,)
    return out


def train_measure(args, world, rank, dev, batch_size, steps, warmup, wire=None, exchange=True, roofline=False):
    """feed_data (forward diffusion) + 2 UNet forwards + losses + backward + flat RCCL all-reduce + fused Adam, `steps` timed
    iterations after `warmup`; returns a dict: seconds (max over ranks), last loss, dropout and -- at world > 1 -- the data-parallel
    exchange as the step saw it (GradSync.timings(): span / exposed ms per step, mean over the timed steps of rank 0), the bytes one
    step puts on the wire per rank, and the same two all-reduces timed ALONE right after (nothing to overlap with).
    exchange=False: the same ranks without the exchange (GradSync switched off; a timing reference, the ranks' weights drift apart)."""
    import torch.distributed as dist
    from instancediff_amd import pipeline
    from instancediff_amd.utils.synthetic import make_batch
    T = 100 if args.T == 1000 else args.T
    os.environ["IDIFF_GRAD_WIRE"] = wire or args.grad_wire
    model, sde = pipeline.build(phase="train", device=dev, T=T, seed=0, dist=world > 1)
    model.set_train()
    sde.set_seed(1234 + rank)
    batch = make_batch(batch_size, args.size, seed=1234 + rank, mixed=True)
    torch.manual_seed(99 + rank)
    sync = model.grad_sync
    if sync is not None:
        sync.timing = True
        if not exchange:
            sync.active = False

    def it():
        model.feed_data(batch)
        return model.optimize_parameters()[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("train: model built; warmup")
    for _ in range(warmup):
        loss = it()
    barrier()
    if sync is not None:
        sync.timings()  # drop the warm-up marks
    lib = __import__("instancediff_amd.ops", fromlist=["_lib"])._lib.load()
    n0 = lib.idiff_launch_count()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = it()
    barrier()
    el = time.perf_counter() - t0
    launches = (lib.idiff_launch_count() - n0) / steps
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    out = {"el": el, "loss": loss, "dropout": float(getattr(model, "score_map_dropout", 0.1)), "library_launches_per_it": launches,
           "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    if sync is not None and sync.active:
        tm = sync.timings()
        flats = model.drift_optimizer.flat_grads() + model.noise_optimizer.flat_grads()
        nbytes = sum(f.numel() for f in flats) * (2 if sync.wire == "bf16" else 4)
        out["exchange"] = {"wire": sync.wire, "bytes_per_step_per_rank": nbytes, "all_reduces_per_step": len(flats),
                           "span_ms": round(sum(t["span_ms"] for t in tm) / max(len(tm), 1), 3),
                           "exposed_ms": round(sum(t["exposed_ms"] for t in tm) / max(len(tm), 1), 3),
                           "span_note": "first start() -> exchanges complete on the step's stream (overlaps the noise net's backward)",
                           "exposed_note": "finish() entered (both backwards + gather done) -> exchanges complete: what the step waited for"}
        # the same exchange alone: nothing on the GPU to hide behind
        reps = 5
        barrier()
        sync.timing = False
        t1 = time.perf_counter()
        for _ in range(reps):
            sync.start(flats)
            sync.finish()
        barrier()
        iso = (time.perf_counter() - t1) / reps
        out["exchange"]["isolated_ms"] = round(iso * 1e3, 3)
        out["exchange"]["isolated_busbw_GBps"] = round(2.0 * (world - 1) / world * nbytes / iso / 1e9, 1)
    if roofline and rank == 0 and world == 1:
        try:
            out["roofline"] = train_roofline(it)
        except Exception as e:  # a reported side measurement
            out["roofline"] = {"error": repr(e)}
    del model, sde
    torch.cuda.empty_cache()
    return out


def train_bench(args, world, rank, dev):
    """Secondary line (BASELINE config c3): training iterations/sec of CLIPDriftModel.optimize_parameters."""
    import torch.distributed as dist
    m = train_measure(args, world, rank, dev, args.batch, args.steps, args.warmup, roofline=not args.no_roofline)
    el, loss = m["el"], m["loss"]
    if rank == 0:
        value = world * args.steps / el
        label = "training iterations/sec (%dx%d bs%d/GPU)" % (args.size, args.size, args.batch)
        if args.grad_wire == "bf16":
            label += " -- VARIANT: bf16 gradient wire format (fp32 compute and fp32 master gradients)"
        print(json.dumps({"metric": label, "value": round(value, 4), "grad_wire": args.grad_wire,
                          "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic", "last_loss": loss, "exchange": m.get("exchange"),
                          "library_launches_per_it": m["library_launches_per_it"], "roofline": m.get("roofline"),
                          "config": {"workload": "%dx%d synthetic, batch %d per GPU, drift+noise UNet fwd/bwd, pyramid losses, Adam"
                                                 % (args.size, args.size, args.batch), "global_batch": args.batch * world,
                                     "parallelism": "dp%d (flat RCCL all-reduce)" % world,
                                     "peak_mem_GB": m["peak_mem_GB"]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def stored_n1_train():
    """(it/s, source) of the newest N=1 training line kept under profiles/ (profiles/rNN/e_bench_train_b32.json), for the weak-scaling
    efficiency of the N-rank training leg; (None, None) when there is none"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "e_bench_train_b32.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.loads(f.read().strip().splitlines()[-1])
            if doc.get("n_gpus", 1) == 1 and doc.get("value"):
                return float(doc["value"]), os.path.relpath(path, ROOT)
        except (OSError, ValueError, IndexError):
            continue
    return None, None


def train_leg_ranks(args, world, rank, dev):
    """BASELINE c3 across the ranks of a multi-GPU run, inside the default line (every rank calls this; rank 0 gets the dict): 256x256,
    batch 32 per GPU, fp32 compute, ONE flat RCCL all-reduce per optimizer per step started under the backward (parallel.GradSync) --
    with the fp32 wire, then the same without the exchange (what the collective costs the step), then the bf16 wire as a labelled
    variant.  The all-reduce is reported three ways: its span inside the step, the part the step waited for, and alone."""
    steps, warmup, bs = 6, 2, 32
    t0 = time.time()

    def one(wire, exchange=True):
        m = train_measure(args, world, rank, dev, bs, steps, warmup, wire=wire, exchange=exchange)
        return {"it_per_s": round(world * steps / m["el"], 4), "ms_per_step": round(m["el"] / steps * 1e3, 2), "exchange": m.get("exchange"),
                "last_loss": m["loss"], "peak_mem_GB": m["peak_mem_GB"]}
    f32 = one("fp32")
    none = one("fp32", exchange=False)
    b16 = one("bf16")
    n1, src = stored_n1_train()
    out = {"n_gpus": world, "batch_per_gpu": bs, "global_batch": bs * world, "steps": steps, "warmup": warmup, "dtype": "f32",
           "grad_wire": "fp32", "it_per_s": f32["it_per_s"], "ms_per_step": f32["ms_per_step"], "exchange": f32["exchange"],
           "ms_per_step_without_exchange": none["ms_per_step"],
           "exchange_cost_ms": round(f32["ms_per_step"] - none["ms_per_step"], 2),
           "n1_it_per_s": n1, "n1_source": src,
           "weak_scaling_efficiency_vs_n1": round(f32["it_per_s"] / (world * n1), 4) if n1 else None,
           "bf16_wire_variant": {"label": "VARIANT: bf16 gradient wire format (fp32 compute and fp32 master gradients)", "it_per_s": b16["it_per_s"],
                                 "ms_per_step": b16["ms_per_step"], "exchange": b16["exchange"],
                                 "weak_scaling_efficiency_vs_n1": round(b16["it_per_s"] / (world * n1), 4) if n1 else None},
           "last_loss": f32["last_loss"], "peak_mem_GB": f32["peak_mem_GB"], "leg_seconds": round(time.time() - t0, 1),
           "workload": "%dx%d synthetic, batch %d per GPU, drift+noise UNet fwd/bwd, pyramid losses, flat RCCL all-reduce of both optimizers' "
                       "gradient buffers, 2 fused Adam steps (BASELINE c3)" % (args.size, args.size, bs)}
    return out if rank == 0 else None


def train_leg(args, dev):
    """The BASELINE c3 workload as an extra key of the default line (N = 1): 256x256, batch 32, fp32, decoder dropout 0.1, a few timed
    iterations after the headline measurement (same process, the sampling model already freed by the caller)."""
    steps, warmup, bs = 6, 2, 32
    t0 = time.time()
    m = train_measure(args, 1, 0, dev, bs, steps, warmup, roofline=not args.no_roofline)
    el = m["el"]
    out = {"it_per_s": round(steps / el, 4), "ms_per_step": round(el / steps * 1e3, 2), "batch": bs, "dropout": m["dropout"], "steps": steps,
           "warmup": warmup, "dtype": "f32", "last_loss": m["loss"], "peak_mem_GB": m["peak_mem_GB"],
           "library_launches_per_it": m["library_launches_per_it"],
           "workload": "%dx%d synthetic, batch %d, drift+noise UNet fwd/bwd, pyramid losses, 2 fused Adam steps (BASELINE c3 at N=1, fp32)"
                       % (args.size, args.size, bs), "leg_seconds": round(time.time() - t0, 1)}
    out["roofline"] = m.get("roofline")
    return out


def train_dryrun(args, world, rank):
    """IDIFF_BENCH_DRYRUN=1 on a box WITHOUT a GPU (tests/test_host_cpu.py): what `bench.py --gpus N` (default line: the training leg
    across the ranks, key "train") and `bench.py --gpus N --mode train` do around the HIP compute -- rendezvous from the launcher's
    environment (gloo), train-phase model build with its GradSync, rank-0 parameter broadcast, one start()/finish() exchange of the
    optimizers' flat gradient buffers in the train step's order per wire format -- with rank-valued stand-in gradients and the
    exchange's timing marks.  No forward / backward runs (there is no CPU fallback for it) and no rate is reported."""
    import torch.distributed as dist
    from instancediff_amd import pipeline
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ok_all, legs = True, {}
    wires = [args.grad_wire] if args.mode == "train" else ["fp32", "bf16"]  # the default line's training leg reports both
    for wire in wires:
        os.environ["IDIFF_GRAD_WIRE"] = wire
        torch.manual_seed(1000 + rank)  # ranks start from DIFFERENT weights: the broadcast must make them equal
        model, _ = pipeline.build(phase="train", device=torch.device("cpu"), T=4, seed=1000 + rank, dist=world > 1)
        sync = model.grad_sync
        assert (sync is not None and sync.active and sync.world == world) if world > 1 else sync is None
        first = next(model.drift_net.parameters()).detach().reshape(-1)[:8].clone()
        ok, scale, tm = True, 1.0, []
        if world > 1:
            sync.timing = True
            got = [torch.empty_like(first) for _ in range(world)]
            dist.all_gather(got, first)
            ok = all(torch.equal(g, got[0]) for g in got)
            for opt in (model.drift_optimizer, model.noise_optimizer):  # the train step's order: drift first, then noise, one finish()
                for f in opt.flat_grads():
                    f.fill_(float(rank + 1))
                sync.start(opt.flat_grads())
            scale = sync.finish()
            tm = sync.timings()
            want = float(world * (world + 1) // 2)
            ok = ok and all(bool((f == want).all()) for opt in (model.drift_optimizer, model.noise_optimizer) for f in opt.flat_grads())
            ok = ok and abs(scale - 1.0 / world) < 1e-12 and len(tm) == 1 and tm[0]["span_ms"] >= tm[0]["exposed_ms"] >= 0.0
            dist.barrier()
        nparam = sum(f.numel() for opt in (model.drift_optimizer, model.noise_optimizer) for f in opt.flat_grads())
        legs[wire] = {"grad_wire": wire, "grad_sync_ok": ok, "flat_gradient_floats": nparam, "scale": scale,
                      "exchange": {"wire": wire, "span_ms": tm[0]["span_ms"] if tm else None, "exposed_ms": tm[0]["exposed_ms"] if tm else None}}
        ok_all = ok_all and ok
    if rank == 0:
        note = "no GPU: rendezvous + model build + parameter broadcast + flat gradient all-reduce only"
        if args.mode == "train":
            leg = legs[args.grad_wire]
            print(json.dumps({"metric": "training iterations/sec (%dx%d bs%d/GPU)" % (args.size, args.size, args.batch), "value": None,
                              "dryrun": note, "n_gpus": world, "grad_wire": args.grad_wire, "grad_sync_ok": leg["grad_sync_ok"],
                              "flat_gradient_floats": leg["flat_gradient_floats"], "scale": leg["scale"], "exchange": leg["exchange"]}), flush=True)
        else:
            n1, src = stored_n1_train()
            train = dict(legs["fp32"], n_gpus=world, batch_per_gpu=32, it_per_s=None, n1_it_per_s=n1, n1_source=src,
                         weak_scaling_efficiency_vs_n1=None, bf16_wire_variant=legs["bf16"])
            print(json.dumps({"metric": "denoising steps/sec (%dx%d bs%d)" % (args.size, args.size, args.batch), "value": None, "dryrun": note,
                              "n_gpus": world, "config": {"parallelism": "replicas x%d (no collective)" % world}, "train": train}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0 if ok_all else 1)


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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (no torch.cuda / HIP call has happened yet)
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("WORLD_SIZE=%d from the launcher overrides --gpus %d" % (world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        if os.environ.get("IDIFF_BENCH_DRYRUN") == "1" and args.mode in ("train", "sample"):
            return train_dryrun(args, world, rank)  # CPU test hook: everything of the N>1 training leg / line EXCEPT the HIP compute
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
        assert dist.get_world_size() == world

    from instancediff_amd import ops, pipeline
    from instancediff_amd.utils.synthetic import make_batch

    variant = None
    if args.attn != "f32" and args.mode == "sample":
        variant = attention_variant_delta(args, dev)  # measured first, on the fp32 path's own inputs
        ops.ATTN_DTYPE = args.attn
    os.environ["IDIFF_GRAD_WIRE"] = args.grad_wire

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
    launches_per_step = None
    graph_mode = run.stepper.mode == "graph"
    if rank == 0 and not args.no_roofline:
        # second pass over the same K steps, on ONE stream (the timed region overlaps the two nets on two streams, which
        # would smear per-launch event times), every conv launch bracketed by HIP events on its launch stream
        two = sde.two_streams
        sde.two_streams = False
        ops.PROFILE = []
        lib = ops._lib.load()
        n0 = lib.idiff_launch_count()
        for _ in range(args.steps):
            run.step()  # eager: per-launch events cannot be recorded inside a graph replay
        launches_per_step = (lib.idiff_launch_count() - n0) / args.steps
        torch.cuda.synchronize()
        sde.two_streams = two
        # The 3x3 convs run on two Winograd kernels (idiff_conv2d_last_algo(): 3 = F(4x4,3x3) conv_wino4_kernel, 1 = F(2x2,3x3)
        # conv_wino_kernel); the dominant kernel is the one with the larger share of the step.  "Executed" FLOP = what the
        # algorithm puts on the matrix pipe: 36 multiply-adds per 4x4 output tile (1/4 of the direct form's 144) resp. 16 per 2x2
        # tile (16/36 of 36).
        allrecs = ops.PROFILE
        ops.PROFILE = None
        KERN = {3: ("conv_wino4_kernel (3x3 conv, Winograd F(4x4,3x3) on f32 MFMA)", 36.0 / 144.0, 4.0),
                4: ("conv_wino4h_kernel (3x3 conv, Winograd F(4x4,3x3) on f32 MFMA, half-patch items, 2 workgroups per CU)", 36.0 / 144.0, 4.0),
                1: ("conv_wino_kernel (3x3 conv, Winograd F(2x2,3x3) on f32 MFMA)", 16.0 / 36.0, 2.25)}
        groups = {}
        for algo, (name, factor, speedup) in KERN.items():
            recs = [r for r in allrecs if r['algo'] == algo]
            if not recs:
                continue
            ms = sum(r['e0'].elapsed_time(r['e1']) for r in recs)
            fl = sum(r['flops'] for r in recs)  # direct-form count 2*Cin*Cout*9*H*W*B
            eff = fl / (ms * 1e-3) / 1e12
            groups[algo] = {"kernel": name, "launches_per_step": len(recs) / args.steps, "ms_per_step_single_stream": round(ms / args.steps, 3),
                            "avg_launch_ms": round(ms / len(recs), 4), "executed_share_of_direct": round(factor, 4),
                            "algorithmic_speedup": speedup, "effective_tflops": round(eff, 2), "achieved": round(eff * factor, 2),
                            "frac": round(eff * factor / F32_MFMA_PEAK_TFLOPS, 4),
                            "executed_gflop_per_launch": round(fl * factor / len(recs) / 1e9, 3), "_fl": fl, "_ms": ms}
        all_ms = sum(r['e0'].elapsed_time(r['e1']) for r in allrecs)
        all_fl = sum(r['flops'] for r in allrecs)
        wino_fl = sum(g["_fl"] for g in groups.values())
        exec_fl_step = (sum(g["_fl"] * KERN[a][1] for a, g in groups.items()) + (all_fl - wino_fl)) / args.steps  # all convs
        dom = max(groups, key=lambda a: groups[a]["_ms"])
        d = groups[dom]
        roof = {"bound": "mfma", "kernel": d["kernel"], "achieved": d["achieved"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": d["frac"], "traffic": None,
                "flop_count": "executed = %.4f of the direct-form 2*Cin*Cout*9*H*W*B per launch" % KERN[dom][1],
                "algorithmic_speedup": d["algorithmic_speedup"], "effective_tflops": d["effective_tflops"],
                "frac_of_direct_form_roofline": round(d["effective_tflops"] / F32_MFMA_PEAK_TFLOPS, 4),
                "frac_note": "frac counts only the multiply-adds the Winograd algorithm executes (1/4 of the direct form for F(4x4,3x3), "
                             "16/36 for the F(2x2,3x3) kernel that was dominant in round 1 at frac 0.62): frac x algorithmic_speedup is the "
                             "like-for-like throughput (r01: 0.62 x 2.25 = 1.40, now frac_of_direct_form_roofline)",
                "step_frac": round(exec_fl_step / (el / args.steps) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                "step_executed_gflop": round(exec_fl_step / 1e9, 1),
                "launches": int(d["launches_per_step"] * args.steps), "avg_launch_ms": d["avg_launch_ms"],
                "executed_gflop_per_launch": d["executed_gflop_per_launch"],
                "conv_ms_per_step_single_stream": round(all_ms / args.steps, 3),
                "wino_ms_per_step_single_stream": round(sum(g["_ms"] for g in groups.values()) / args.steps, 3),
                "step_conv_gflop_direct_form": round(all_fl / args.steps / 1e9, 1),
                "kernels": [{k: v for k, v in groups[a].items() if not k.startswith("_")} for a in sorted(groups, key=lambda a: -groups[a]["_ms"])]}
        roof.update(pmc_evidence(d["kernel"]))
    if world > 1:
        barrier()

    default_workload = args.size == 256 and args.batch == 16 and variant is None
    train_ranks = None
    if world > 1 and default_workload and not args.no_train_leg:
        # BASELINE c3 -- the one configuration with an exchange step -- across the ranks of THIS run: every rank takes part (the RCCL
        # all-reduce of the flat gradient buffers is collective); rank 0 reports it under "train"
        log("train leg across %d ranks (BASELINE c3: flat RCCL all-reduce under the backward) ..." % world)
        del run
        torch.cuda.empty_cache()
        try:
            train_ranks = train_leg_ranks(args, world, rank, dev)
        except Exception as e:
            if rank != 0:
                raise
            train_ranks = {"error": repr(e)}

    if rank == 0:
        # the training leg runs BEFORE the CPU baseline: measured right behind the oracle's ~110 s of 16-thread host work (GPU idle)
        # it read 4-5 % low (3.75 against 3.91-3.98 it/s, profiles/r04/README.md)
        train = None
        if world > 1:
            train = train_ranks if train_ranks is not None else "skipped (%s)" % ("--no-train-leg" if args.no_train_leg else "not the default 256x256 batch-16 fp32 line")
        elif args.no_train_leg or not default_workload:
            train = "skipped (%s)" % ("--no-train-leg" if args.no_train_leg else "not the default 256x256 batch-16 fp32 line")
        else:
            log("train leg (BASELINE c3 at N=1) ...")
            try:
                del run
                torch.cuda.empty_cache()
                train = train_leg(args, dev)
            except Exception as e:  # a reported side measurement; never hide the headline
                train = {"error": repr(e)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only: the other ranks of a multi-GPU run must not wait on it
            log("cpu baseline (oracle) ...")
            try:
                cpu = cpu_baseline(args, batch)
            except Exception as e:  # the baseline is a reported side measurement; never hide the GPU result
                cpu = {"error": repr(e)}
        elif world > 1:
            cpu = "N=1 line only"
        value = world * args.steps / el
        label = "denoising steps/sec (%dx%d bs%d)" % (args.size, args.size, args.batch)
        if variant is not None:
            label += " -- REDUCED-PRECISION VARIANT: %s MFMA self-attention" % args.attn
        line = {"metric": label, "value": round(value, 4),
                "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if variant is None else "f32 + %s attention" % args.attn, "data": "synthetic", "graph": graph_mode,
                "arithmetic": {"conv3x3": "f32 MFMA (v_mfma_f32_16x16x4_f32), Winograd F(4x4,3x3); F(2x2,3x3) / direct form on the shapes it does not tile",
                               "conv1x1": "bf16x3 split of fp32 operands, 6 bf16 MFMAs (v_mfma_f32_16x16x32_bf16) per product, fp32 accumulate: "
                                          "fp32-class result (2e-6 vs fp64, tests/test_ops_gpu.py); f32 MFMA where the pixels do not tile by 256",
                               "attention": "f32 MFMA" if variant is None else "%s MFMA self-attention (operands clamped to +-255 for f16), fp32 softmax" % args.attn,
                               "groupnorm_statistics": "fp32 partials, fp64 fixed-order reduce", "sde_update": "f32 (one rounding per op)"},
                "two_streams": bool(sde.two_streams), "variant": variant,
                "launches_per_step": launches_per_step,
                "config": {"workload": "%dx%d 1-ch synthetic, %d-step reverse chain, batch %d per GPU, 2 UNet fwd + reverse update per step"
                                       % (args.size, args.size, args.T, args.batch),
                           "global_batch": args.batch * world, "parallelism": "replicas x%d (no collective)" % world,
                           "image_steps_per_s": round(value * args.batch, 2)},
                "roofline": roof, "cpu_baseline": cpu, "train": train}
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
