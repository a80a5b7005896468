#!/usr/bin/env python3
"""Headline benchmark: explanations/sec on synthetic 16 kHz 4 s clips (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one pass of the whole hot path over one batch of 64 clips already resident in HBM:
STFT -> wav2vec2-base embedder + logreg -> U-Net mask decoder -> masked ISTFT x2 -> embedder x2
(SURVEY.md §8d "1 explanation"), BASELINE config 2's batch and models.  Every rank runs its own
batches (utterances shard with no data-path collective, "weak" scaling); the only exchange is the
fixed-order gather of the per-clip probabilities for the LMAC metrics, done once inside the timed
region.  Rank 0 prints ONE JSON line.

`roofline` is for the dominant kernel (the implicit-GEMM tile with the largest total time, csrc/gemm.hip): algorithmic
FLOPs of its launches / their device time, measured with HIP events recorded on the launch stream
inside the timed steps.  `cpu_baseline` times the CPU oracle (plain torch, the reference arithmetic)
on a bounded sample of the same workload on this host's cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

BATCH = 64
AUDIO_LENGTH = 4
MFMA_F16_PEAK_TFLOPS = 2500.0        # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the 3B embedder batch is split over")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: gloo backend, every rank on cuda:0 (checks the sharding / "
                         "gather / timing logic where no multi-GPU node is available; not a measurement)")
    ap.add_argument("--tune", action="store_true",
                    help="time every candidate GEMM tile per launch first and keep the fastest (default: the 128x128 tile)")
    ap.add_argument("--no-tune", action="store_true", help=argparse.SUPPRESS)       # kept for old command lines
    ap.add_argument("--cpu-clips", type=int, default=32)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline (a 1-GPU box owns 16)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = 0 if args.rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.rehearse else "nccl", rank=rank, world_size=world)   # "nccl" = RCCL on ROCm

    from addvisor_hip import gemm as G, pipeline as P, synthetic as syn
    torch.set_grad_enabled(False)

    cfg = syn.base_config()
    emb_sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    unet_sd = syn.unet_weights()
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, dev, audio_length=AUDIO_LENGTH, streams=args.streams)

    B, L = args.batch, AUDIO_LENGTH * 16000
    n_batches = max(1, min(args.steps, 8))
    # clip indices are global and disjoint per rank: rank r, batch j -> clips [(j*world + r)*B, ...)
    batches = [syn.make_clips(B, L, first=(j * world + rank) * B).to(dev) for j in range(n_batches)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.tune:
        pipe.tune(B)                                   # per-launch GEMM tile selection, outside the timed region
        if args.verbose and rank == 0:
            for rec in G.TUNER.log:
                print("tuned", rec, file=sys.stderr)
    for i in range(args.warmup):
        pipe.explain(batches[i % n_batches])
    barrier()

    G.PROFILE.reset(enabled=os.environ.get("ADDVISOR_BENCH_NO_EVENTS", "0") == "0")     # 1: measure the cost of the per-launch events
    probs = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = pipe.explain(batches[i % n_batches])
        probs.append(torch.cat([out["predictions"], out["theta_out"], out["masked_predictions"]], 1))
    local = torch.cat(probs, 0)
    if args.rehearse and world > 1:                                        # gloo has no CUDA all_gather: stage through the host
        allp = P.gather_probabilities(local.cpu(), local.shape[0] * world).to(dev)
    else:
        allp = P.gather_probabilities(local, local.shape[0] * world)       # the one exchange step (RCCL)
    metrics = P.lmac_metrics(allp[:, 0].contiguous(), allp[:, 1].contiguous(), allp[:, 2].contiguous())
    barrier()
    elapsed = time.perf_counter() - t0
    G.PROFILE.enabled = False
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_expl = world * B * args.steps
    value = n_expl / elapsed
    gemm_ms, gemm_flops, gemm_n = G.PROFILE.summary()                      # the tile with the largest total time
    gemm_kernel = getattr(G.PROFILE, "kernel", "gemm_f16_kernel")         # the instantiation rocprofv3 lists under this name
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
    flops_step = pipe.flops(B)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, emb_sd, coef, icpt, unet_sd, args.cpu_clips, L, args.cpu_threads)

    traffic = None                       # HBM bytes per launch of the dominant kernel, from the committed PMC passes
    tpath = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
    if os.path.exists(tpath) and not args.tune:
        with open(tpath) as fh:
            for k, v in json.load(fh).items():
                if gemm_kernel in k:
                    traffic = round(v["hbm_bytes_per_launch"])
    if rank == 0:
        line = {
            "metric": "explanations/sec (16 kHz, 4 s clips)", "value": round(value, 2), "unit": "explanations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "BASELINE config 2 extended to the full explanation: batch 64 x 4 s clips, "
                                   "STFT + wav2vec2-base embedder (to hidden layer 9) + logreg + U-Net mask decoder "
                                   "+ masked ISTFT x2 + embedder re-forward x2 + LMAC metrics",
                       "batch_per_gpu": B, "clip_samples": L, "embedder": "wav2vec2-base (seeded random weights)",
                       "precision": "fp16 operands, fp32 accumulate / norms / residual", "sharding": f"utterance x{world}",
                       "gflop_per_explanation": round(flops_step / B / 1e9, 1)},
            "lmac": {k: round(v, 6) for k, v in metrics.items()},
            "pipeline_tflops": round(flops_step * args.steps * world / elapsed / 1e12, 1),
            "roofline": {"kernel": gemm_kernel + (" (tuned tile choice)" if args.tune else ""), "bound": "mfma",
                         "achieved": None if achieved is None else round(achieved, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / MFMA_F16_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": "profiles/r01_gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                         "FETCH_SIZE x2 per MI355X_MICROARCH.md; bytes per launch)" if traffic else None, "launches": gemm_n,
                         "avg_launch_us": None if not gemm_n else round(1e3 * gemm_ms / gemm_n, 2),
                         "gflop_per_launch": None if not gemm_n else round(gemm_flops / gemm_n / 1e9, 2)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(cfg, emb_sd, coef, icpt, unet_sd, n_clips, L, threads):
    """The CPU oracle (plain torch = the reference arithmetic) on a bounded sample, this host's cores."""
    from addvisor_hip import synthetic as syn
    from oracle import lmac_ref
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    w = syn.make_clips(n_clips, L)
    t0 = time.perf_counter()
    with torch.no_grad():
        lmac_ref.explain(w, emb_sd, cfg, coef, icpt, unet_sd, audio_length=AUDIO_LENGTH)
    dt = time.perf_counter() - t0
    return {"value": round(n_clips / dt, 3), "unit": "explanations/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_clips} clips of the same workload through oracle/lmac_ref.explain (plain torch fp32), {dt:.1f} s"}


if __name__ == "__main__":
    main()
