#!/usr/bin/env python3
"""Headline benchmark: spectrograms/s of one full VQ-VAE train step (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = standardise -> encoder -> pre-VQ conv -> VQ -> decoder -> MSE -> backward -> [one RCCL all-reduce of
the flat gradient buffer] -> Adam, on a synthetic (B,201,500) batch already resident in HBM (speech ctor of
scripts/train_speech.py:152-153, B=64 per GPU, weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")
for _p in (ROOT, PKG, os.path.join(PKG, "src")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: exact-fp32 matrix rate (= vector rate)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (NOT the 2:1-sparsity headline)
PEAK = {"f32": F32_MFMA_PEAK_TFLOPS, "bf16": BF16_MFMA_PEAK_TFLOPS, "bf16x3": BF16_MFMA_PEAK_TFLOPS}
CONV_FAMILIES = {"f32": ("conv1d_f32_kernel", "conv1d_wgrad_f32_kernel"),
                 "bf16": ("conv1d_bf16_k3_kernel", "conv1d_bf16_v2_kernel", "conv1d_bf16_kernel", "conv1d_wgrad_bf16_v2_kernel"),
                 "bf16x3": ("conv1d_bf16x3_kernel", "conv1d_wgrad_bf16x3_kernel")}
SPEECH_CFG = (201, 1024, 128, 3, 1024, 0.25, 1024)          # scripts/train_speech.py:152-153
RIR_CFG = (500, 1024, 64, 2, 64, 0.25, 1024)                # scripts/train_rir.py:147-149


def algorithmic_gflop_per_spectrogram(cfg, L, out_channels=None):
    """train = 3*F_conv - (no dgrad into the input) + VQ distance GEMM (forward only)."""
    c, h, d, r, rh, _, k = cfg
    oc = c if out_channels is None else out_channels
    enc1 = 2 * c * h * 3 * L
    res = 2 * h * rh * 3 * L + 2 * rh * h * L
    pre = 2 * h * d * 3 * L
    dec1 = 2 * d * h * 3 * L
    up = 2 * h * h * 3 * L
    last = 2 * h * oc * 3 * L
    f_conv = enc1 + 2 * r * res + pre + dec1 + 2 * up + last
    return (3 * f_conv - enc1 + 2 * L * k * d) / 1e9


def cpu_baseline(seconds_budget=20.0):
    """The oracle restatement (kind "port") timed on this host: speech ctor, B=4 (BASELINE configs[0])."""
    from oracle import vqvae_oracle as O
    # the GPU box exposes 256 logical CPUs but gives a 1-GPU job a 16-CPU share; oversubscribing is 100x slower
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, int(os.environ.get("ALVQ_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    shapes = O.vqvae_param_shapes(201, 1024, 128, 1024, 1024)
    params = O.closed_form_params(shapes, codebook_scale=1.0, gain=0.5)
    tr = O.OracleTrainer(params, 3, 0.25, use_jitter=True)
    x = O.speech_preprocess(torch.randn(4, 201, 500))
    np.random.seed(0)
    tr.step(x)                                   # warm-up
    tr.step(x)
    times, t_end = [], time.time() + seconds_budget
    while len(times) < 3 or (time.time() < t_end and len(times) < 20):
        t0 = time.time()
        tr.step(x)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": 4.0 / med, "unit": "spectrograms/s", "cores": cores, "kind": "port",
            "sample": "%d train steps of the CPU oracle (same ATen op sequence as the reference), speech ctor, "
                      "B=4 x (201,500) fp32, jitter on, Adam; median %.3f s/step" % (len(times), med)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="spectrograms per GPU")
    ap.add_argument("--config", default="speech", choices=["speech", "rir", "echoed"],
                    help="speech = BASELINE configs[1] (the headline); rir = configs[2]; echoed = configs[4]")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3"],
                    help="bf16: BASELINE configs[1] (bf16 storage/MFMA, fp32 accumulate+master weights); f32: parity mode "
                         "on the exact-fp32 MFMA; bf16x3: split-bf16 parity mode (3 bf16 MFMAs per product)")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the secondary fp32 parity-mode measurement")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # ALVQ_BENCH_BACKEND=gloo rehearses the N>1 control flow with several ranks sharing one card (RCCL refuses
    # two ranks on one device); the driver's runs use the default: one rank per GPU over RCCL.
    backend = os.environ.get("ALVQ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from acoustic_locating_vq_vae import _native as N
    from acoustic_locating_vq_vae import _ops
    from acoustic_locating_vq_vae.vq_vae.convolutional_vq_vae import ConvolutionalVQVAE
    from acoustic_locating_vq_vae.train_step import Trainer

    _ops.set_compute_dtype(args.dtype)
    torch.manual_seed(0)                         # identical init on every rank (also broadcast by Trainer)
    np.random.seed(1234 + rank)                  # jitter: per-rank numpy stream
    B = args.batch
    if args.config == "speech":
        cfg, L, oc = SPEECH_CFG, 500, None
        model = ConvolutionalVQVAE(*cfg).cuda()
        kind = "speech"
    elif args.config == "rir":
        cfg, L, oc = RIR_CFG, 201, 1
        model = ConvolutionalVQVAE(*cfg, use_jitter=False, out_channels=1).cuda()
        kind = "rir"
    else:
        # scripts/train_echoed_speech.py:45-46 from two freshly initialised sub-models (no checkpoints ship)
        from acoustic_locating_vq_vae.vq_vae.echoed_speech_model import EchoedSpeechReconModel
        cfg, L, oc = SPEECH_CFG, 500, None
        rir = ConvolutionalVQVAE(*RIR_CFG, use_jitter=False, out_channels=1)
        sp = ConvolutionalVQVAE(*SPEECH_CFG)
        model = EchoedSpeechReconModel(rir, sp, 201, 1024, 2, 1024, True).cuda()
        kind = "echoed"
    model.train()
    trainer = Trainer(model, kind)
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + rank)
    raw = torch.randn(B, 201, 500, device="cuda", generator=g)      # synthetic spectrogram batch, resident in HBM
    wiener = torch.randn(B, 201, device="cuda", generator=g) if kind == "rir" else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(steps, warmup, use_timer):
        for _ in range(warmup):
            out = trainer.step(raw, wiener)
        barrier()
        timer = N.KernelTimer() if use_timer else None
        if timer is not None:
            timer.__enter__()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = trainer.step(raw, wiener)
        barrier()
        dt = time.perf_counter() - t0
        if timer is not None:
            timer.__exit__()
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        last = float(out[0])
        assert np.isfinite(last), "non-finite loss"
        return dt, last, (timer.summary() if timer is not None else None)

    def roofline(summ, dtype, steps):
        # the dominant kernel = the conv kernel with the largest share of the timed region
        fam = max((f for f in CONV_FAMILIES[dtype] if f in summ), key=lambda f: summ[f][1])
        n, secs, flops = summ[fam]
        ach = flops / secs / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(fam)
        return {"bound": "mfma", "kernel": fam, "achieved": ach, "peak": PEAK[dtype], "unit": "TFLOP/s",
                "frac": ach / PEAK[dtype], "traffic": traffic, "launches": n, "avg_launch_ms": 1e3 * secs / n,
                "algorithmic_gflop_per_launch": flops / n / 1e9}

    def families(summ, steps):
        return {k: {"launches": v[0], "ms_per_step": 1e3 * v[1] / steps, "tflops": (v[2] / v[1] / 1e12) if v[1] > 0 else None}
                for k, v in summ.items()}

    graph = "off"
    if not args.no_graph:
        try:
            trainer.capture(raw, wiener)
            graph = "hipGraph replay"
        except Exception as exc:                           # fall back to eager launches, and say so
            graph = "off (capture failed: %s)" % (str(exc).splitlines()[0][:120],)
            trainer._graph = None
    # timed region: K steps (graph replay when captured).  HIP events cannot bracket kernels inside a replay,
    # so the per-kernel durations behind `roofline` come from an instrumented eager pass of the same K steps
    # on the same model/batch right after it.
    elapsed, loss, summ = measure(args.steps, args.warmup, (not args.no_kernel_timer) and trainer._graph is None)
    if summ is None and not args.no_kernel_timer:
        g_saved, trainer._graph = trainer._graph, None
        _, _, summ = measure(args.steps, 1, True)
        trainer._graph = g_saved

    if rank == 0:
        gf = 61.73 if kind == "echoed" else algorithmic_gflop_per_spectrogram(cfg, L, oc)
        value = world * B * args.steps / elapsed
        line = {
            "metric": "spectrograms/sec (train step), %s VQ-VAE default config" % ("echoed-speech" if kind == "echoed" else kind),
            "value": value, "unit": "spectrograms/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s VQ-VAE train step (fwd+bwd+Adam), ctor %s, B=%d per GPU x (%s), %s, jitter %s"
                                   % (kind, list(cfg), B, "500,201" if kind == "rir" else "201,500",
                                      {"bf16": "bf16 storage + bf16 MFMA, fp32 accumulate / VQ / losses / master weights",
                                       "bf16x3": "split-bf16 (hi+lo planes, 3 bf16 MFMAs per product, fp32 accumulate)",
                                       "f32": "fp32 storage + exact-fp32 MFMA"}[args.dtype],
                                      "off" if kind == "rir" else "on"),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "algorithmic_gflop_per_spectrogram": gf},
            "model_tflops": value * gf / 1e3,
            "final_loss": loss,
            "launch": graph,
        }
        if summ is not None:
            line["roofline"] = roofline(summ, args.dtype, args.steps)
            line["kernel_families"] = families(summ, args.steps)
    if args.dtype == "bf16" and not args.no_f32_line:
        # secondary lines on the same model and batch (eager launches; the bf16 graph does not apply):
        #   f32    -- the parity mode on the exact-fp32 MFMA (1e-3 / bit-exact claims are made for this one);
        #   bf16x3 -- the split-bf16 parity mode (fp32-grade forward parity at 3 bf16 MFMAs per product)
        for mode, key in (("f32", "f32_parity_mode"), ("bf16x3", "bf16x3_parity_mode")):
            _ops.set_compute_dtype(mode)
            g_saved, trainer._graph = trainer._graph, None
            s2 = max(3, min(5, args.steps))
            e2, l2, _ = measure(s2, 2, False)                 # the rate: no per-launch events in the timed region
            summ2 = None if args.no_kernel_timer else measure(s2, 0, True)[2]   # the roofline: instrumented pass
            trainer._graph = g_saved
            _ops.set_compute_dtype(args.dtype)
            if rank == 0:
                v2 = world * B * s2 / e2
                line[key] = {"value": v2, "unit": "spectrograms/s", "ms_per_step": 1e3 * e2 / s2, "steps": s2,
                             "model_tflops": v2 * gf / 1e3, "launch": "eager"}
                if summ2 is not None:
                    line[key]["roofline"] = roofline(summ2, mode, s2)
    if rank == 0 and world == 1 and kind == "speech" and not args.no_f32_line:
        # what scripts/train_speech.py itself gets when its imports resolve to this build: the script's own loop body
        # (torch ops for |x| / standardise / MSE, loss.backward(), torch.optim.Adam) on the module API -- no
        # Trainer, no flat buffers, no graph.  Same model config, batch and dtype; eager launches.
        import torch.nn.functional as F
        m2 = ConvolutionalVQVAE(*cfg).cuda().train()
        opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3, amsgrad=False)

        def script_step():
            x = torch.abs(raw)
            x = (x - torch.mean(x, dim=1, keepdim=True)) / (torch.std(x, dim=1, keepdim=True) + 1e-8)
            opt2.zero_grad()
            vq_loss, recon, _ = m2(x)
            (F.mse_loss(recon, x, reduction="mean") + vq_loss).backward()
            opt2.step()

        for _ in range(3):
            script_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s3 = max(3, min(10, args.steps))
        for _ in range(s3):
            script_step()
        torch.cuda.synchronize()
        e3 = time.perf_counter() - t0
        line["script_loop_mode"] = {"value": B * s3 / e3, "unit": "spectrograms/s", "ms_per_step": 1e3 * e3 / s3, "steps": s3,
                                    "launch": "eager, module API + torch.optim.Adam (train_speech.py:62-74,88-91)"}
        del m2, opt2
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
