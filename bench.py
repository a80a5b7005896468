#!/usr/bin/env python3
"""Headline benchmark: explanations/sec on synthetic 16 kHz 4 s clips (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher: this script starts the N ranks itself (child processes through
        `python -m torch.distributed.run`, before anything touches the GPU) and relays rank 0's JSON line;
        under torchrun (WORLD_SIZE set) it is one of the ranks.

A *step* is one pass of the whole hot path over one batch of 64 clips already resident in HBM:
STFT -> wav2vec2-base embedder + logreg -> U-Net mask decoder -> masked ISTFT x2 -> embedder x2
(SURVEY.md §8d "1 explanation"), BASELINE config 2's batch and models.  Every rank runs its own
batches (utterances shard with no data-path collective, "weak" scaling); the only exchange is the
fixed-order gather of the per-clip probabilities for the LMAC metrics, done once inside the timed
region.  Rank 0 prints ONE JSON line.

Precision: the headline runs the fp32-class mode (`"dtype": "f32"`: split-format operands, three fp16 MFMAs per
product, fp32 accumulate -- the arithmetic class of the fp32 reference; its matrix peak is 2.5 PFLOP/s / 3).  The
fp16-operand mode is measured right after it and reported under `"f16"`.  `--precision f16` makes fp16 the headline.

`roofline` is for the dominant kernel (the implicit-GEMM tile with the largest total time, csrc/gemm.hip): algorithmic
FLOPs of its launches / their device time, measured with HIP events recorded on the launch stream inside the timed
steps.  `cpu_baseline` times the CPU oracle (plain torch, the reference arithmetic) on a bounded sample of the same
workload on this host's cores (rank 0, N = 1 only).  The default N = 1 invocation also measures BASELINE config 3
(`"hifigan"`: HiFi-GAN V1, 256 x 251 mel frames), config 5's per-GPU share (`"ig"`: IntegratedGradients, 50 steps x
16 clips, wav2vec2-large, fp32-class gradient chain; the fp16 chain under its `"f16"` sub-key), the vocoder variant of the step (`"explain_vocoder"`) and one rank's data-set loop of config 4
through the drop-in `LMAC_metrics.run_addvisor_metrics` itself (`"dataset"`: 8 905 clips from an in-memory Dataset in pinned host memory,
DataLoader + per-item upload + collate inside the timed region, ragged last batch) and SURVEY 8(f)'s training step (`"train"`: batch 64,
U-Net forward / backward + LMAC-loss forward / backward + Adam through the drop-in modules, fp32-class chain; fp16 under `"f16"`) and
appends them as extra keys; `--workload hifigan|ig|dataset|xlsr2b|train` runs one of them as the headline.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "xai-audio-deepfakes_amd"))
sys.path.insert(0, ROOT)

BATCH = 64
AUDIO_LENGTH = 4
DATASET_CLIPS = 71237                       # BASELINE config 4: the ASVspoof2019-LA evaluation partition's size
DATASET_SHARE = -(-DATASET_CLIPS // 8)      # one of eight ranks' contiguous block (8 905 clips, last batch 9)
MFMA_F16_PEAK_TFLOPS = 2500.0        # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK = {"f16": MFMA_F16_PEAK_TFLOPS, "f32": MFMA_F16_PEAK_TFLOPS / 3.0}     # fp32-class: 3 fp16 MFMAs per product
PEAK_NOTE = {"f16": "dense fp16 MFMA peak", "f32": "dense fp16 MFMA peak / 3: the fp32-class mode issues three fp16 MFMAs per product "
             "(the native fp32 MFMA peak is 157 TFLOP/s)"}


DEFAULT_STEPS = 40


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=DEFAULT_STEPS, help="timed steps (default 40: ~2 s of timed work, past the clock ramp of the first second)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", choices=("explain", "hifigan", "ig", "dataset", "xlsr2b", "train"), default="explain")
    ap.add_argument("--precision", choices=("f32", "f16"), default="f32", help="headline precision of the explain workload")
    ap.add_argument("--vocoder", action="store_true", help="explain workload with both resyntheses re-rendered by the HiFi-GAN V1 vocoder (the north-star variant) as the headline")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (f16 / hifigan / ig keys)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the 3B embedder batch is split over")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: gloo backend, every rank on cuda:0 (checks the launch / sharding / "
                         "gather / timing logic where no multi-GPU node is available; not a measurement)")
    ap.add_argument("--tune", action="store_true",
                    help="time every candidate GEMM tile per launch first and keep the fastest (default: the 128x128 tile)")
    ap.add_argument("--no-tune", action="store_true", help=argparse.SUPPRESS)       # kept for old command lines
    ap.add_argument("--cpu-clips", type=int, default=64)
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = every core this process may use)")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two child rocprofv3 --pmc passes that measure the dominant kernel's HBM bytes per launch")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (never exec: this process stays the
    parent and has not touched the GPU) and relay their output; rank 0 prints the JSON line."""
    port = args.master_port or (29500 + os.getpid() % 20000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def usable_cpus() -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:                                                   # cgroup v2 quota of the box ("max 100000" = unlimited)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
    torch.set_grad_enabled(False)
    ctx = dict(args=args, dev=dev, rank=rank, world=world, dist=dist)

    if args.workload == "hifigan":
        line = hifigan_line(ctx, bench_hifigan(ctx, args.batch if args.batch != BATCH else 256, args.steps, args.warmup))
    elif args.workload == "ig":
        line = ig_line(ctx, bench_ig(ctx, 16, 160, args.precision))
    elif args.workload == "xlsr2b":
        r = bench_xlsr2b(ctx, args.precision, args.steps if args.steps != DEFAULT_STEPS else 3, min(args.warmup, 1) or 1)
        line = {"metric": "explanations/sec (16 kHz, 4 s clips), XLS-R-2B-width embedder", "value": r["value"],
                "unit": "explanations/s", "n_gpus": world, "steps": r["steps"], "warmup": 1, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic", "config": {"workload": r["workload"]},
                "lmac": r["lmac"], "roofline": r["roofline"], "pipeline_tflops": r["pipeline_tflops"], "cpu_baseline": None}
    elif args.workload == "train":
        r = bench_train(ctx, args.precision, args.batch)
        line = {"metric": "mask-decoder training clips/sec (4 s clips, wav2vec2-base frozen)", "value": round(r["value"] * world, 1), "unit": "clips/s",
                "n_gpus": world, "steps": r["steps"], "warmup": 2, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": r["dtype"], "data": "synthetic", "config": {"workload": r["workload"]}, "f16": r.get("f16"),
                "cpu_baseline": None}
    elif args.workload == "dataset":
        r = bench_dataset(ctx, args.precision)
        line = {"metric": "explanations/sec over a host-resident data set (PCIe upload included)", "value": r["value"], "unit": "explanations/s",
                "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": round(1e3 * r["seconds"], 1), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.precision, "data": "synthetic", "config": {"workload": r["workload"]}, "lmac": r["lmac"],
                "cpu_baseline": None}
    else:
        line = explain_line(ctx)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ the explanation step
def run_explain(ctx, precision, steps, warmup, vocoder=False, cfg=None):
    """Time `steps` explanation steps in one precision; returns the numbers of that run.  `vocoder`: the north-star variant
    in which both resyntheses are re-rendered by the HiFi-GAN V1 vocoder (mel front end + generator, same precision as the rest
    of the path) before the classifier re-forward."""
    import torch
    from addvisor_hip import gemm as G, pipeline as P, synthetic as syn
    args, dev, rank, world, dist = ctx["args"], ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    cfg = cfg or syn.base_config()
    emb_sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    unet_sd = syn.unet_weights()
    voc = None
    if vocoder:
        from addvisor_hip.hifigan import HipHifigan
        hcfg = syn.HifiganConfig()
        voc = HipHifigan(hcfg, syn.hifigan_weights(hcfg), dev, precision=precision)      # the vocoder runs at the path's precision
    pipe = P.ExplainPipeline(cfg, emb_sd, coef, icpt, unet_sd, dev, audio_length=AUDIO_LENGTH, streams=args.streams, precision=precision,
                             vocoder=voc)
    B, L = args.batch, AUDIO_LENGTH * 16000
    n_batches = max(1, min(steps, 8))
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
    for i in range(warmup):
        pipe.explain(batches[i % n_batches])
    barrier()

    G.PROFILE.reset(enabled=os.environ.get("ADDVISOR_BENCH_NO_EVENTS", "0") == "0")     # 1: measure the cost of the per-launch events
    probs = []
    t0 = time.perf_counter()
    for i in range(steps):
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
    res = dict(precision=precision, elapsed=elapsed, steps=steps, value=world * B * steps / elapsed, metrics=metrics,
               flops_step=pipe.flops(B), roofline=gemm_roofline(G, precision, args.tune, committed=(cfg.hidden_size == 768 and not vocoder and B == BATCH)),
               cfg=cfg, weights=(emb_sd, coef, icpt, unet_sd))
    del pipe, batches
    torch.cuda.empty_cache()
    return res


HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E (MI355X_MICROARCH.md)


def stft_roofline(dev, B=BATCH, reps=20):
    """HBM roofline of the two signal kernels north_star names first (csrc/stft.hip), as the explanation step runs them:
    framed rFFT `advh_stft_forward` (X + |X|, no phase) and the fused masked inverse `advh_istft_masked_c64` (mask application
    + irFFT + overlap-add, mask-in AND mask-out).  `achieved` = bytes each launch actually moves (every operand and result once)
    / average launch time from HIP events on the launch stream; `algorithmic` = SURVEY.md §8(d)'s per-clip accounting
    (1.89 MB forward with the phase output, 2 x 1.47 MB inverse with magnitude + phase inputs)."""
    import torch
    from addvisor_hip import _lib, ops, synthetic as syn
    L, hop, win = AUDIO_LENGTH * 16000, 322, 644
    w = syn.make_clips(B, L).to(dev)
    spec, mag, _ = ops.stft_forward(w, L, hop, win, want_complex=True, want_phase=False)
    T, Fm, Tm = mag.shape[2], 512, (mag.shape[2] // 4) * 4
    mask = torch.rand(B, Fm, Tm, device=dev)
    o_in, o_out = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    xr = torch.view_as_real(spec)

    def inverse():
        _lib.check(_lib.lib().advh_istft_masked_c64(xr.data_ptr(), mask.data_ptr(), Fm, Tm, 2, o_in.data_ptr(), o_out.data_ptr(), L, B, T, L,
                                                    hop, win, None, st), "advh_istft_masked_c64")

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    t_f = timed(lambda: ops.stft_forward(w, L, hop, win, want_complex=True, want_phase=False))
    t_i = timed(inverse)
    nb = 513 * T
    moved_f = B * (L * 4 + nb * 8 + nb * 4)                          # wave in; X (c64) + |X| out
    moved_i = B * (nb * 8 + Fm * Tm * 4 + 2 * L * 4)                 # X (c64) + mask in; two waveforms out
    alg_f, alg_i = B * 1.89e6, B * 2 * 1.47e6
    rec = lambda kern, t, moved, alg: {"kernel": kern, "bound": "hbm", "us_per_launch": round(t * 1e6, 1), "bytes_moved": int(moved),
                                       "achieved": round(moved / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(moved / t / 1e9 / HBM_PEAK_GBS, 4),
                                       "algorithmic_GBs_survey_8d": round(alg / t / 1e9, 1)}
    return {"workload": f"{B} clips x 4 s, as inside the explanation step", "stft_forward": rec("stft_fwd_kernel (X + |X|)", t_f, moved_f, alg_f),
            "istft_masked": rec("istft_kernel (complex source, mask-in + mask-out, log1p domain)", t_i, moved_i, alg_i)}


def gemm_roofline(G, precision, tuned=False, committed=False):
    """`committed`: attach the committed PMC traffic figure (profiles/*gemm_traffic.json) -- only meaningful for the default explain
    workload those passes ran; every other workload reports traffic null unless it measures its own."""
    ms, flops, n = G.PROFILE.summary()                                     # the tile instantiation with the largest total time
    kernel = getattr(G.PROFILE, "kernel", "gemm_f16_kernel")              # ... under the name rocprofv3 lists it
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else None
    peak = PEAK[precision]
    traffic, source = committed_traffic(kernel) if (committed and not tuned) else (None, None)
    return {"kernel": kernel + (" (tuned tile choice)" if tuned else ""), "bound": "mfma",
            "achieved": None if achieved is None else round(achieved, 1), "peak": round(peak, 1), "peak_note": PEAK_NOTE[precision],
            "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / peak, 4),
            "traffic": traffic, "traffic_source": source, "launches": n,
            "avg_launch_us": None if not n else round(1e3 * ms / n, 2),
            "gflop_per_launch": None if not n else round(flops / n / 1e9, 2)}


def measured_traffic(kernel, precision, batch=BATCH, streams=1):
    """HBM bytes per launch of `kernel`, measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes; FETCH_SIZE x2:
    gfx950 counts 128-byte requests as 64; both counters in KiB), started as child processes after the timed region (the
    parent never execs).  Returns (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tot = {}
    tmp = tempfile.mkdtemp(prefix="advh_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-traffic",
                   "--precision", precision, "--batch", str(batch), "--streams", str(streams)]      # the SAME workload as the timed run
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, env=dict(os.environ, TMPDIR="/tmp"))
            vals = []
            for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(fn) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                            vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}, {len(vals)} samples)"
            tot[counter] = sum(vals) / len(vals)
        nbytes = 2.0 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024
        return round(nbytes), ("measured in this invocation: two child `rocprofv3 --pmc` passes of `bench.py --steps 2` (FETCH_SIZE x2 + WRITE_SIZE, "
                               "KiB -> bytes, mean per launch of the dominant kernel)")
    except Exception as e:                               # profiling must never take the benchmark line down
        return None, f"traffic measurement failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in
    separate runs of this same command, tools/pmc_bench_traffic.sh).  PMC counters cannot be read from inside an unprofiled
    run, so this is NOT a measurement of the present run: the source file is named next to the number, else null."""
    for name in ("r02_gemm_traffic.json", "r01_gemm_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as fh:
                for k, v in json.load(fh).items():
                    if kernel in k:
                        return round(v["hbm_bytes_per_launch"]), (f"profiles/{name}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                                                  "command (FETCH_SIZE x2 per MI355X_MICROARCH.md), bytes per launch; not re-measured in this run")
    return None, None


def explain_line(ctx):
    args, rank, world = ctx["args"], ctx["rank"], ctx["world"]
    head = run_explain(ctx, args.precision, args.steps, args.warmup, vocoder=args.vocoder)
    B, L = args.batch, AUDIO_LENGTH * 16000
    prec_txt = {"f32": "fp32-class: split-format (hi + lo * 2^-11) operands, 3 fp16 MFMAs per product, fp32 accumulate / norms / residual",
                "f16": "fp16 operands, fp32 accumulate / norms / residual"}
    line = {
        "metric": "explanations/sec (16 kHz, 4 s clips)", "value": round(head["value"], 2), "unit": "explanations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * head["elapsed"] / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "BASELINE config 2 extended to the full explanation: batch 64 x 4 s clips, "
                               "STFT + wav2vec2-base embedder (to hidden layer 9) + logreg + U-Net mask decoder "
                               "+ masked ISTFT x2 + embedder re-forward x2 + LMAC metrics",
                   "batch_per_gpu": B, "clip_samples": L, "embedder": "wav2vec2-base (seeded random weights)",
                   "precision": prec_txt[args.precision], "sharding": f"utterance x{world}",
                   **({"vocoder": "HiFi-GAN V1 (mel front end + generator) re-renders both resyntheses before the classifier re-forward"} if args.vocoder else {}),
                   "gflop_per_explanation": round(head["flops_step"] / B / 1e9, 1)},
        "lmac": {k: round(v, 6) for k, v in head["metrics"].items()},
        "pipeline_tflops": round(head["flops_step"] * args.steps * world / head["elapsed"] / 1e12, 1),
        "roofline": head["roofline"],
        "cpu_baseline": None,
    }
    line["roofline_hbm"] = stft_roofline(ctx["dev"], B)        # every rank runs it (identical, a few ms): no rank leaves the others waiting
    extras = world == 1 and not args.no_extras and not args.tune and not args.vocoder
    if rank == 0 and world == 1 and not args.no_traffic and not args.tune and head["roofline"]["launches"]:
        kern = head["roofline"]["kernel"]
        nbytes, src = measured_traffic(kern, args.precision, args.batch, args.streams)
        if nbytes is not None:
            line["roofline"]["traffic"], line["roofline"]["traffic_source"] = nbytes, src
        else:
            line["roofline"]["traffic_source"] = f"{src}; falling back to " + str(line["roofline"]["traffic_source"])
    if extras:
        other = "f16" if args.precision == "f32" else "f32"
        o = run_explain(ctx, other, args.steps, args.warmup)
        line[other] = {"value": round(o["value"], 2), "unit": "explanations/s", "ms_per_step": round(1e3 * o["elapsed"] / args.steps, 3),
                       "precision": prec_txt[other], "lmac": {k: round(v, 6) for k, v in o["metrics"].items()},
                       "pipeline_tflops": round(o["flops_step"] * args.steps / o["elapsed"] / 1e12, 1), "roofline": o["roofline"]}
        v = run_explain(ctx, args.precision, 3, 1, vocoder=True)
        line["explain_vocoder"] = {"workload": "the same step with both resyntheses re-rendered by the HiFi-GAN V1 vocoder (mel front end + generator, "
                                               "128 clips x 251 frames per step, at the path's precision) before the classifier re-forward",
                                   "value": round(v["value"], 2), "unit": "explanations/s", "ms_per_step": round(1e3 * v["elapsed"] / v["steps"], 3),
                                   "steps": v["steps"], "dtype": args.precision,
                                   "lmac": {k: round(x, 6) for k, x in v["metrics"].items()}}
        line["xlsr2b"] = bench_xlsr2b(ctx, args.precision)
        line["dataset"] = bench_dataset(ctx, args.precision)
        line["hifigan"] = bench_hifigan(ctx, 256, 5, 1)
        line["ig"] = bench_ig(ctx, 16, 160)
        line["train"] = bench_train(ctx, args.precision)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cfg = head["cfg"]
        emb_sd, coef, icpt, unet_sd = head["weights"]
        line["cpu_baseline"] = cpu_baseline(cfg, emb_sd, coef, icpt, unet_sd, args.cpu_clips, L, args.cpu_threads, args.cpu_repeats)
    return line


# ------------------------------------------------------------------------------------------ the reference's own embedder shape
def bench_xlsr2b(ctx, precision, steps=3, warmup=1):
    """The explanation step with the embedder the reference itself loads (classifier_embedder.py:13-16, 25: XLS-R-2B -- hidden
    1920, 16 heads x 120, FFN 7680, layer-norm feature extractor, pre-LN encoder -- truncated to the layers hidden_states[9]
    needs, `nn.Linear(1920, 1)` head) instead of BASELINE config 2's wav2vec2-base: same 64 x 4 s batch, same U-Net, three
    embedder passes per explanation.  Seeded random weights (the checkpoint is private)."""
    from addvisor_hip import synthetic as syn
    cfg = syn.xlsr2b_config(num_hidden_layers=10)
    r = run_explain(ctx, precision, steps, warmup, cfg=cfg)
    B = ctx["args"].batch
    return {"workload": f"explanation step with the reference's embedder shape: XLS-R-2B width (hidden 1920, 16 x 120 heads, FFN 7680, 9 encoder "
                        f"layers to hidden_states[9]) + logreg(1920) + U-Net, batch {B} x 4 s, one GPU",
            "value": round(r["value"], 2), "unit": "explanations/s", "ms_per_step": round(1e3 * r["elapsed"] / r["steps"], 2), "steps": r["steps"],
            "dtype": precision, "gflop_per_explanation": round(r["flops_step"] / B / 1e9, 1),
            "pipeline_tflops": round(r["flops_step"] * r["steps"] / r["elapsed"] / 1e12, 1),
            "lmac": {k: round(v, 6) for k, v in r["metrics"].items()}, "roofline": r["roofline"]}


# ------------------------------------------------------------------------------------------ BASELINE config 4 (one GPU's loop)
def bench_dataset(ctx, precision, n_clips=DATASET_SHARE, pool=512):
    """BASELINE config 4 as one rank runs it, through the reference driver's OWN entry point: the drop-in
    `LMAC_metrics.run_addvisor_metrics` (LMAC_metrics.py:117-172) over an in-memory `Dataset` of `n_clips` 4 s clips per
    rank -- `DataLoader(batch_size=64, collate_fn=LMAC_metrics.collate_fn)`, every item uploaded from pinned HOST memory
    inside `__getitem__` (`.to(device)`, LMAC_metrics.py:106), ragged last batch, the fused explanation step per batch, one
    gather of the per-clip probabilities and the five metrics at the end.  The rate INCLUDES the loader, the per-item PCIe
    uploads and the collate -- unlike the headline, whose batches are resident.  Under `--gpus N` the function shards the
    `N * n_clips` clips itself (block partition -> RCCL all_gather), every rank times its own walk and the line reports
    total clips / max-over-ranks time."""
    import contextlib
    import torch
    from addvisor_hip import runtime as rt, synthetic as syn
    args, dev, rank, world, dist = ctx["args"], ctx["dev"], ctx["rank"], ctx["world"], ctx["dist"]
    saved = {k: os.environ.get(k) for k in ("ADDVISOR_PRECISION", "ADDVISOR_EMBEDDER")}
    os.environ["ADDVISOR_PRECISION"], os.environ["ADDVISOR_EMBEDDER"] = precision, "base"
    rt.reset()
    import LMAC_metrics
    LMAC_metrics._model = None
    LMAC_metrics._pipes.clear()
    old_len = LMAC_metrics.audio_processor.audio_length
    LMAC_metrics.audio_processor.audio_length = AUDIO_LENGTH
    B, L = BATCH, AUDIO_LENGTH * 16000
    host = syn.make_clips(pool, L).pin_memory()                                  # the "files": cycled through to reach the data-set size
    n_total = n_clips * world

    copy_s = torch.cuda.Stream(device=dev)

    class InMemory(torch.utils.data.Dataset):
        def __len__(self):
            return n_total

        def __getitem__(self, i):
            # LMAC_metrics.py:101-106: per-item upload.  The copy is issued on a copy stream, so the items of batch k+1 (the loader
            # runs ahead of the GPU: nothing in the loop synchronises) cross PCIe while batch k computes; the compute stream
            # waits for the copy stream, never the other way round.
            cur = torch.cuda.current_stream()
            with torch.cuda.stream(copy_s):
                t = host[(i * 7) % pool].to(dev, non_blocking=True)
            cur.wait_stream(copy_s)
            t.record_stream(cur)
            return t, f"clip{i}.wav"

    class Head(torch.utils.data.Dataset):                                            # warm-up: one full and one ragged batch build both workspaces
        def __len__(self):
            return B + (n_clips % B or B)

        def __getitem__(self, i):
            return host[i % pool].to(dev), f"w{i}.wav"

    try:
        with contextlib.redirect_stdout(sys.stderr):                                 # the driver prints its five lines; stdout carries the JSON line only
            if dist is None:
                LMAC_metrics.run_addvisor_metrics("", "", batch_size=B, dataset=Head())
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            metrics = LMAC_metrics.run_addvisor_metrics("", "", batch_size=B, dataset=InMemory())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
    finally:
        LMAC_metrics.audio_processor.audio_length = old_len
        LMAC_metrics._model = None
        LMAC_metrics._pipes.clear()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        rt.reset()
    out = {"workload": f"BASELINE config 4 through LMAC_metrics.run_addvisor_metrics (the reference driver's entry point): {n_clips} clips x 4 s per rank "
                       f"(one of eight ranks' share of an ASVspoof2019-LA-sized set of {DATASET_CLIPS}) from an in-memory Dataset in pinned host memory, "
                       f"DataLoader batches of {B} (last batch {n_clips % B or B}), per-item upload in __getitem__, fused explanation step, LMAC metrics "
                       "at the end; wav2vec2-base + U-Net",
           "value": round(n_total / dt, 1), "unit": "explanations/s (loader + PCIe upload included)", "seconds": round(dt, 3), "clips": n_total,
           "n_gpus": world, "dtype": precision, "upload_GB": round(n_total * L * 4 / 1e9, 3), "lmac": {k: round(v, 6) for k, v in metrics.items()}}
    del host
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------ BASELINE config 3
def bench_hifigan(ctx, B, steps, warmup, precision="f16", both=True):
    """HiFi-GAN V1 vocoder, B x 251 mel frames (4 s) -> waveform.  fp16 operands by default (the vocoder's stated tolerance is on
    waveforms: 1.5e-3 measured); the fp32-class mode of the explanation path (1.7e-6) is timed too and reported under "f32"."""
    import torch
    from addvisor_hip import gemm as G, synthetic as syn
    from addvisor_hip.hifigan import HipHifigan
    dev = ctx["dev"]
    T = 251
    cfg = syn.HifiganConfig()
    net = HipHifigan(cfg, syn.hifigan_weights(cfg), dev, precision=precision)
    g = torch.Generator().manual_seed(7)
    mel = (torch.randn(B, 80, T, generator=g) * 2 - 4).to(dev)
    for _ in range(max(1, warmup)):
        net.decode_batch(mel)
    torch.cuda.synchronize()
    G.PROFILE.reset(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        wav = net.decode_batch(mel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    G.PROFILE.enabled = False
    fl = net.flops(B, T)
    out = {"workload": f"BASELINE config 3: HiFi-GAN V1 decode_batch, {B} x {T} mel frames (4 s), one GPU", "value": round(B / dt, 1),
           "unit": "clips/s", "ms_per_batch": round(dt * 1e3, 2), "steps": steps, "dtype": precision,
           "gflop_per_clip": round(fl / B / 1e9, 1), "tflops": round(fl / dt / 1e12, 1), "finite": bool(torch.isfinite(wav).all().item()),
           "roofline": gemm_roofline(G, precision)}
    del net, mel, wav
    torch.cuda.empty_cache()
    if both and precision == "f16":
        r = bench_hifigan(ctx, B, 2, 1, "f32", both=False)
        out["f32"] = {k: r[k] for k in ("value", "unit", "ms_per_batch", "tflops", "finite", "roofline")}
    return out


def hifigan_line(ctx, r):
    args, world = ctx["args"], ctx["world"]
    return {"metric": "HiFi-GAN V1 vocoder clips/sec (4 s, 251 mel frames)", "value": round(r["value"] * world, 1), "unit": "clips/s", "n_gpus": world,
            "steps": r["steps"], "warmup": args.warmup, "ms_per_step": r["ms_per_batch"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic", "config": {"workload": r["workload"]}, "roofline": r["roofline"],
            "tflops": r["tflops"], "cpu_baseline": None}


# ------------------------------------------------------------------------------------------ BASELINE config 5
def bench_ig(ctx, B, chunk, precision="f32", both=True):
    """IntegratedGradients, n_steps = 50, wav2vec2-large, B clips = one GPU's share of config 5's batch of 128 over 8 GPUs;
    path-batched forward + dgrad-only backward.  Headline precision: the fp32-class chain (split-format activations and
    gradients, three fp16 MFMAs per product, attention backward on the fp32 MFMA -- the reference differentiates with fp32
    autograd, captum_saliency.py:131-135); the fp16-operand chain is timed too and reported under "f16".
    `chunk` = Captum's internal_batch_size (the reference leaves it unset): 160 = ten whole steps per chunk, five equal chunks; 64 pads the
    50 steps to 13 chunks of 4 (1 496 against 1 632 path points/s, profiles/r03_ig_internal_batch.txt)."""
    import torch
    from addvisor_hip import gemm as G, synthetic as syn
    from addvisor_hip.attribution import HipAttribution
    from addvisor_hip.embedder import HipEmbedder
    dev = ctx["dev"]
    cfg = syn.large_config()
    sd = syn.embedder_weights(cfg)
    coef, icpt = syn.logreg_weights(cfg.hidden_size)
    att = HipAttribution(HipEmbedder(cfg, sd, coef, icpt, dev, precision=precision))
    w = syn.make_clips(B, 64000).to(dev)
    att.integrated_gradients(w, n_steps=max(1, chunk // B), internal_batch_size=chunk)      # warm-up: builds the one chunk-shaped workspace
    torch.cuda.synchronize()
    G.PROFILE.reset(True)
    t0 = time.perf_counter()
    attr = att.integrated_gradients(w, n_steps=50, internal_batch_size=chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    G.PROFILE.enabled = False
    rows = min(chunk, 50 * B) // B * B if chunk >= B else B
    fwd = att.emb.flops(rows, 64000) / rows
    out = {"workload": f"BASELINE config 5, one GPU's share: IntegratedGradients n_steps=50, wav2vec2-large, {B} clips x 4 s, internal batch {chunk}",
           "value": round(50 * B / dt, 1), "unit": "path points/s", "clips_per_s": round(B / dt, 3), "seconds": round(dt, 3), "dtype": precision,
           "fwd_gflop_per_point": round(fwd / 1e9, 1), "approx_tflops_fwd_plus_dgrad": round(2 * fwd * 50 * B / dt / 1e12, 1),
           "finite": bool(torch.isfinite(attr).all().item()), "roofline": gemm_roofline(G, precision)}
    del att, attr, w
    torch.cuda.empty_cache()
    if both and precision == "f32":
        r = bench_ig(ctx, B, chunk, "f16", both=False)
        out["f16"] = {k: r[k] for k in ("value", "unit", "seconds", "approx_tflops_fwd_plus_dgrad", "finite", "roofline")}
    return out


def bench_train(ctx, precision="f32", B=64, steps=5, warmup=2, both=True):
    """one step of the mask-decoder training loop (train_addvisor.py:364-381) through the drop-in modules -- U-Net
    forward / backward on the HIP training kernels, LMAC loss forward and backward (masked ISTFT x2, frozen wav2vec2-base x2 with saves,
    their input-gradient chain, ISTFT adjoint x2), two Adam steps -- at the path's precision; the fp16-operand chain under "f16"."""
    import torch
    from addvisor_hip import runtime as rt, synthetic as syn
    saved = {k: os.environ.get(k) for k in ("ADDVISOR_PRECISION", "ADDVISOR_EMBEDDER", "ADDVISOR_AUDIO_LENGTH")}
    os.environ.update(ADDVISOR_PRECISION=precision, ADDVISOR_EMBEDDER="base", ADDVISOR_AUDIO_LENGTH=str(AUDIO_LENGTH))
    rt.reset()
    try:
        import addvisor, loss_function
        dev = ctx["dev"]
        ap = loss_function.audio_processor
        ap.audio_length = AUDIO_LENGTH
        w = syn.make_clips(B, AUDIO_LENGTH * 16000, seed=3).to(dev)
        _, mag, ph = ap.compute_stft(w)
        _, p = ap.classify(w)
        T4 = 4 * (mag.shape[2] // 4)
        x = mag[:, :512, :T4].unsqueeze(1).contiguous()
        net = addvisor.UNet().to(dev)
        net.train()
        loss = loss_function.LMACLoss().to(dev)
        opt_m, opt_w = torch.optim.Adam(net.parameters(), lr=3e-5), torch.optim.Adam(loss.parameters(), lr=1e-4)
        vals = []

        def step():
            with torch.enable_grad():
                total, _, _ = loss.loss_function(net(x).float(), mag, ph, p)
                opt_m.zero_grad(); opt_w.zero_grad()
                total.backward()
            opt_m.step(); opt_w.step()
            return total

        dist = ctx["dist"]

        def barrier():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            vals.append(step())
        barrier()
        dt = (time.perf_counter() - t0) / steps
        if dist is not None:                                   # every rank trains its own replica on its own clips: the slowest rank's time
            t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        vals = [float(v) for v in vals]
        out = {"workload": f"mask-decoder training step (train_addvisor.py:364-381): batch {B} x {AUDIO_LENGTH} s, U-Net forward / backward + LMAC loss "
                           "forward / backward through the frozen wav2vec2-base + two Adam steps, one GPU", "value": round(B / dt, 1), "unit": "clips/s",
               "ms_per_step": round(1e3 * dt, 2), "steps": steps, "dtype": precision, "finite": all(v == v and abs(v) != float("inf") for v in vals),
               "loss_first_last": [round(vals[0], 5), round(vals[-1], 5)]}
        del net, loss, opt_m, opt_w, x, mag, ph, w
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        rt.reset()
        torch.cuda.empty_cache()
    if both and precision == "f32":
        r = bench_train(ctx, "f16", B, steps, warmup, both=False)
        out["f16"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "finite", "loss_first_last")}
    return out


def ig_line(ctx, r):
    args, world = ctx["args"], ctx["world"]
    return {"metric": "IntegratedGradients path points/sec (50 steps, wav2vec2-large, 4 s clips)", "value": round(r["value"] * world, 1),
            "unit": "path points/s", "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": round(1e3 * r["seconds"], 1), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": r["dtype"], "data": "synthetic", "config": {"workload": r["workload"]},
            "roofline": r["roofline"], "f16": r.get("f16"), "cpu_baseline": None}


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(cfg, emb_sd, coef, icpt, unet_sd, n_clips, L, threads, repeats):
    """The CPU oracle (plain torch = the reference arithmetic) on a bounded sample of the same workload on this host's
    cores: `n_clips` clips in chunks of 16, `repeats` repetitions, median."""
    import torch
    from addvisor_hip import synthetic as syn
    from oracle import lmac_ref
    nthreads = threads if threads > 0 else usable_cpus()
    torch.set_num_threads(max(1, nthreads))
    w = syn.make_clips(n_clips, L)
    times = []
    with torch.no_grad():
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            for i in range(0, n_clips, 16):
                lmac_ref.explain(w[i:i + 16], emb_sd, cfg, coef, icpt, unet_sd, audio_length=AUDIO_LENGTH)
            times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    return {"value": round(n_clips / dt, 3), "unit": "explanations/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model(), "host_logical_cpus": os.cpu_count(),
            "sample": f"{n_clips} clips of the same workload through oracle/lmac_ref.explain (plain torch fp32) in chunks of 16, "
                      f"{len(times)} repetitions, median {dt:.1f} s (all: {', '.join(f'{t:.1f}' for t in times)} s)"}


if __name__ == "__main__":
    main()
