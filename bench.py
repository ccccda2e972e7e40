#!/usr/bin/env python3
"""Headline benchmark: frames/s of uPIT BLSTM TRAINING on WSJ0-2mix-shaped synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the training hot path over one batch already resident in HBM:
random h0/c0 -> BLSTM (input-projection GEMMs + persistent recurrence) -> BatchNorm -> Linear+sigmoid ->
PIT-MSE loss -> full backward -> [N>1: RCCL all-reduce of the flat gradient] -> clip_grad_norm_(0.25) + Adam.
Workload (BASELINE.json configs[1]): uPIT 3x896 BLSTM, 2 speakers, 512-pt STFT features (257 bins),
batch 32 x 400 frames PER GPU (weak scaling), fp32.  frames = sum of valid STFT frames per step.
The features are produced before the timed region by the STFT kernel from synthetic 8 kHz PCM.
--ragged: the WSJ0-2mix-SHAPED variant of the same workload (SURVEY.md 8d's variable-length set): utterance lengths
U(24 000, 64 000) samples from a fixed seed, 32 utterances per step, a DIFFERENT batch every step, frames = the valid
frames only; not the headline line (that stays the T = 400 configuration BASELINE.json quotes the metric on).

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline      the dominant kernel (fp32 MFMA GEMM): algorithmic FLOP / HIP-event time, live in the timed region
  cpu_baseline  the CPU oracle's train step (torch-CPU port of the reference loop) on a bounded sample
  secondary     (default run on one GPU only) the other one-GPU BASELINE configurations, timed in the same process after the
                headline's timed region: the variable-length set in fp32 and bf16, bf16 3-speaker, RSH 4-speaker, the headline
                workload on the fp32-MFMA kernels throughout (the reference's literal arithmetic) and the reference's own default
                model and batch size, 2 x 600 / 100 utterances (SECONDARY)
  aux           (default run on one GPU only) the HBM-bound kernels `north_star` names -- STFT, mask-apply + iSTFT, PIT loss forward
                and backward: us per launch, GB/s of algorithmic bytes, fraction of the HBM peak
  library       which libsepkern.so ran and its sk_build_flags() (0 = the product build; a diagnostic build needs --diagnostic)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "speech-separation_amd")
for p in (ROOT, PKG, os.path.join(PKG, "archs")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 matrix peak (same guide)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* at 64 FLOP/clk/SIMD, 256 CUs, 2.4 GHz
# fp32 products formed on the bf16 matrix pipe by the three-way split of both operands, SIX bf16 piece products per fp32 product
# (gemm_f32_kernel_split3 / the split form of the stream-K kernel; the forward recurrence): the pipe that bounds them is the
# bf16 one, at a sixth of its rate in fp32-equivalent FLOP
PEAK_F32_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=896)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--num-spk", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: BASELINE configs[3] arithmetic (bf16 matrix-core inputs for the non-recurrent GEMMs, "
                         "fp32 accumulate); the headline metric is quoted on f32")
    ap.add_argument("--ragged", action="store_true",
                    help="variable-length utterances, U(24k, 64k) samples (188..501 frames), a different batch every step")
    ap.add_argument("--padded-rows", action="store_true",
                    help="diagnostic A/B for --ragged: run every product and recurrence step over the zero-PADDED (T_max, B) grid "
                         "(the r03 layout's cost) instead of the packed rows; frames still count the valid ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--aux", action="store_true", help="time the streaming kernels (STFT, mask-iSTFT, PIT forward / backward) "
                                                       "also on a non-default run (the default run on one GPU always does)")
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--diagnostic", action="store_true",
                    help="accept a library whose sk_build_flags() is not 0 (a timing-only / ablation build named by SEPKERN_LIB, "
                         "with SEPKERN_ALLOW_DIAGNOSTIC_LIB=1): the line then carries its flags; without this flag bench.py exits")
    ap.add_argument("--no-power-probe", action="store_true")
    ap.add_argument("--arch", choices=["upit", "rsh"], default="upit",
                    help="rsh: the recurrent-selective-hearing arch (BASELINE configs[4]); not the headline metric")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default run on one GPU only: skip the secondary BASELINE configurations (ragged set, bf16 3-spk, "
                         "bf16 ragged, RSH 4-spk) that are timed after the headline and reported under \"secondary\"")
    ap.add_argument("--secondary-only", type=str, default="",
                    help="comma-separated names from SECONDARY: time only these (diagnostics)")
    return ap.parse_args(argv)


def make_batch(torch, ops, synth, B, T, S, first_utt, nsamp=None, padded_rows=False):
    """Synthetic PCM -> STFT magnitudes on the GPU as PACKED rows (PackedSequence.data layout, sepkern.packing): mix (R,257),
    sources [(R,257)], their Packing, the padded mix (T,B,257) and the PCM.  nsamp = None: all utterances T frames; else the
    per-utterance sample counts (sorted longest first, as the reference's collator sorts a batch, archs/uPIT.py:39)."""
    from sepkern.packing import Packing
    if nsamp is None:
        nsamp = [128 * (T - 1) + 64] * B
    nsamp = sorted((int(v) for v in nsamp), reverse=True)
    pcms = synth.pcm_batch(B, num_spk=S, first_utt=first_utt, lengths=nsamp)
    frames = [1 + n // 128 for n in nsamp]
    pk = Packing.from_lens(frames, "cuda")
    F = 257
    feats, padded_mix = [], None
    for k in range(S + 1):
        out = torch.zeros(pk.T, B, F, device="cuda")
        ops.stft_batch([torch.from_numpy(p[k]).cuda() for p in pcms], out=out, out_offs=[b * F for b in range(B)],
                       stride_t=[B * F] * B, stride_f=[1] * B)
        padded_mix = out if k == 0 else padded_mix
        feats.append(out.view(pk.T * B, F) if padded_rows else pk.pack(out))
    if padded_rows:        # the network sees B utterances of T_max frames each (zeros past an utterance's end)
        grid = Packing.from_lens([pk.T] * B, "cuda")
        grid.R_valid = pk.R
        pk = grid
    return feats[0], feats[1:], pk, padded_mix, pcms


def cpu_baseline(H, L, S, B, T, budget_s=300.0):
    """The oracle's train step (same torch-CPU ops as the reference loop, steps/train_qsub.py:116-122: nn.LSTM,
    BatchNorm1d, Linear, PIT-MSE, clip_grad_norm_, Adam) timed at the WORKLOAD'S OWN batch shape, B x T frames:
    3 steps after a short warm-up on a small batch (thread pool, allocator) -- fewer only if the next one
    would not fit in `budget_s` (at least one); min / median per step reported."""
    import numpy as np
    import torch
    from oracle import upit as OU
    torch.manual_seed(0)
    # the GPU box gives a 1-GPU job a 16-CPU share of a much larger host: os.cpu_count() would oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))
    torch.set_num_threads(threads)
    model = OU.OracleSepDNN(num_spk=S, hidden_dim=H, num_layers=L)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)

    def batch(nb, nt):
        samples = []
        for _ in range(nb):
            d = {"mix": np.abs(rng.standard_normal((nt, 257))).astype(np.float32)}
            for s in range(S):
                d["source%d" % (s + 1)] = np.abs(rng.standard_normal((nt, 257))).astype(np.float32)
            samples.append(d)
        return OU.collate(samples)
    OU.train_step(model, opt, batch(4, 20), model.init_hidden(4))        # warm-up, not the measured shape
    full = batch(B, T)
    t0 = time.time()
    per = []
    # torch's CPU nn.LSTM backward is slow at this size (measured: 76-84 s per 32 x 400 step on the GPU box's 16-thread
    # share): three steps are about four minutes
    while len(per) < 3 and (not per or (time.time() - t0) * (len(per) + 1) / len(per) <= budget_s):
        t1 = time.time()
        OU.train_step(model, opt, full, model.init_hidden(B))
        per.append(time.time() - t1)
        log("CPU baseline step %d: %.1f s" % (len(per), per[-1]))
    dt, n = time.time() - t0, len(per)
    return {"value": round(n * B * T / dt, 1), "unit": "frames/s", "cores": threads, "kind": "port",
            "seconds_per_step": round(dt / n, 2), "seconds_per_step_min": round(min(per), 2),
            "seconds_per_step_median": round(sorted(per)[n // 2], 2), "steps": n,
            "sample": "oracle train step (torch-CPU nn.LSTM/BN/Linear + PIT-MSE + clip + Adam), same %dx%d model, the "
                      "workload's own batch of %d x %d frames, %d steps after a small-batch warm-up" % (L, H, B, T, n)}


def launch_ranks(n):
    """`python bench.py --gpus N` called WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the
    launcher.  It has not imported torch and never touches the GPU; it starts one fresh child process per rank (the same
    command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's stdout -- the
    one JSON line -- and exits with the worst return code.  Nothing is exec'ed.  When one rank fails the others are
    terminated by pid (they would otherwise wait in a collective until its timeout)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SEPKERN_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    log("launcher: started %d ranks (pids %s), rendezvous 127.0.0.1:%d" % (n, [p.pid for p in procs], port))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode("utf-8", "replace").splitlines()))
    reader.start()
    worst, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                log("launcher: rank %d exited with code %d" % (r, rc))
                worst = worst or rc
        if worst and live:
            time.sleep(5.0)                               # let the others report their own error first
            for r in sorted(live):
                if procs[r].poll() is None:
                    procs[r].terminate()
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            live.clear()
        time.sleep(0.2)
    reader.join()
    for line in out0:
        print(line, flush=True)
    sys.exit(worst if 0 <= worst < 256 else 1)


def measure(args, env, standalone_pass=True):
    """One workload (model, optimizer, resident batches) timed as the module docstring says: W warm-up steps, K timed
    steps between barrier + synchronize pairs, max over ranks.  Returns (the JSON line's dict without `cpu_baseline` /
    `aux` / `secondary`, what aux_kernels() needs).  Called once for the headline and -- default run on one GPU -- once
    per secondary BASELINE configuration, in the same process."""
    torch, dist, skdist = env.torch, env.dist, env.skdist
    world, rank, local = env.world, env.rank, env.local
    from sepkern import ops, synth, _lib
    from sepkern.optim import ClipAdam
    import uPIT
    if args.arch == "rsh":
        import RSH as arch_mod
    else:
        arch_mod = uPIT

    H, L, S, B, T = args.hidden, args.layers, args.num_spk, args.batch, args.frames
    DT = "bf16" if args.dtype == "bf16" else "fp32"
    torch.manual_seed(0)                                  # identical initial weights on every rank
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # SepDNN prints its conf keys like the reference does
        if args.arch == "rsh":
            model = arch_mod.SepDNN(local, hidden_dim=str(H), num_layers=str(L), dtype=DT)
        else:
            model = uPIT.SepDNN(local, num_spk=str(S), hidden_dim=str(H), num_layers=str(L), dtype=DT)
    model.cuda()
    model.train()
    model.hidden_generator = torch.Generator(device="cuda")
    model.hidden_generator.manual_seed(1234 + rank)       # per-rank h0/c0 stream
    opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
    log("model ready (%dx%d, %d speakers); building the synthetic batch%s" % (L, H, S, "es" if args.ragged else ""))
    if args.ragged and args.arch != "upit":
        sys.exit("bench: --ragged is the uPIT workload")
    import numpy as np
    if args.ragged:
        # SURVEY.md 8d's variable-length set: U(24 000, 64 000) samples (3-8 s at 8 kHz = 188..501 frames), fixed seed; one
        # batch per step of the run, all of them resident in HBM before the timed region starts
        rng = np.random.default_rng(2024 + rank)
        n_batches = min(64, args.steps + args.warmup)
        pool = [make_batch(torch, ops, synth, B, T, S, (rank * n_batches + i) * B, nsamp=rng.integers(24000, 64001, B),
                           padded_rows=args.padded_rows) for i in range(n_batches)]
    else:
        pool = [make_batch(torch, ops, synth, B, T, S, rank * B)]
    pcms, mix_padded = pool[0][4], pool[0][3]
    log("%d batch(es) resident in HBM: %d utterances, %s frames each" %
        (len(pool), B, "/".join(str(b[2].R) for b in pool[:4]) + ("/..." if len(pool) > 4 else "")))
    loss_acc = torch.zeros(2, device="cuda")
    counters = {"i": 0, "frames": 0, "T": 0}

    if args.arch == "rsh":
        lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
        srcs_padded = [pool[0][2].unpack(s_) for s_ in pool[0][1]]
        combos = torch.cat([mix_padded, torch.ones_like(mix_padded)], 2).contiguous()   # [mixture | attention = 1] (archs/RSH.py:104-106)
        groups = [(S, combos, srcs_padded, lens)]

    def step():
        mix, srcs, pk = pool[counters["i"] % len(pool)][:3]
        counters["i"] += 1
        counters["frames"] += getattr(pk, "R_valid", pk.R)
        counters["T"] += pk.T
        if args.arch == "rsh":
            loss, norm = arch_mod.compute_loss_padded(model, groups)
        else:
            loss, norm = uPIT.compute_loss_packed(model, mix, srcs, pk)   # packed rows, as the collator's PackedSequences hold them
        loss_acc[0] += loss.detach() * norm               # epoch loss bookkeeping stays on the device
        loss_acc[1] += norm
        loss.backward()
        opt.step()

    def warm_up():
        for i in range(args.warmup):
            step()
            torch.cuda.synchronize()
            log("warm-up step %d done" % (i + 1))

    warm_up()
    # A persistent recurrence launch that could not keep its grid co-resident times out (bounded spins), flags the
    # step through the gradient buffer -- on EVERY rank, the flag is all-reduced with the gradients -- and the fused
    # optimizer skips it.  Then continue in this same process with one launch per step (never re-exec a process
    # that has touched the GPU) and say so in the JSON; a second failure is fatal.
    lstm_fallback = None
    if opt.skipped() > 0:
        log("a persistent recurrence launch timed out during warm-up: falling back to SEPKERN_LSTM_MODE=2")
        try:
            model.check_status()
        except _lib.SepkernError:
            pass
        model._engine.lstm_mode = 2
        lstm_fallback = "per-step launches (mode 2) after a timed-out persistent launch"
        skipped_before = opt.skipped()
        warm_up()
        if opt.skipped() > skipped_before:
            sys.exit("bench: the recurrence fails in per-step mode too")
    skipped_before = opt.skipped()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_events:
        ops.PROF = {}
    if world > 1:
        skdist.TIMING = []
    counters["frames"] = counters["T"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames_timed, T_mean = counters["frames"], counters["T"] / float(args.steps)     # this rank's valid frames in the timed region
    log("timed region done: %.3f ms/step" % (1000.0 * dt / args.steps))
    prof = ops.prof_summary()
    ops.PROF = None
    # Untimed extra pass for the roofline's "standalone" figure: the same step with the engine's co-scheduling of
    # weight-gradient GEMMs and recurrences switched off, so every launch has the device to itself.
    prof_alone = None
    if standalone_pass and not args.no_kernel_events and model._engine is not None and model._engine.overlap:
        model._engine.overlap = False
        step()
        torch.cuda.synchronize()
        ops.PROF = {}
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        prof_alone = ops.prof_summary()
        ops.PROF = None
        model._engine.overlap = True
    if opt.skipped() > skipped_before:
        sys.exit("bench: %d step(s) of the timed region were skipped (timed-out recurrence launch)" % (opt.skipped() - skipped_before))
    dist_info = None
    if world > 1:
        mine = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        ar = skdist.TIMING or []
        skdist.TIMING = None
        ar_ms = sum(a.elapsed_time(b) for a, b in ar) / max(1, len(ar))
        # which physical device every rank computed on: "did the collective see N GPUs" is answerable from the line
        import hashlib
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = "%s|%s|%s|%d" % (getattr(props, "uuid", ""), getattr(props, "pci_bus_id", ""), getattr(props, "pci_device_id", ""),
                                 torch.cuda.current_device())
        h = int.from_bytes(hashlib.sha256(ident.encode()).digest()[:7], "little")
        mine_dev = torch.tensor([h], device="cuda", dtype=torch.int64)
        every_dev = [torch.zeros_like(mine_dev) for _ in range(world)]
        dist.all_gather(every_dev, mine_dev)
        fr = torch.tensor([frames_timed], device="cuda", dtype=torch.int64)
        dist.all_reduce(fr)
        frames_timed = int(fr.item())
        # per-rank recurrence mode: a rank that fell back to per-step launches sets the pace of the whole job
        fb = torch.tensor([1 if model._engine.lstm_mode == 2 else 0], device="cuda", dtype=torch.int64)
        every_fb = [torch.zeros_like(fb) for _ in range(world)]
        dist.all_gather(every_fb, fb)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "lstm_per_step_launches_by_rank": [int(t.item()) for t in every_fb],
                     "distinct_devices": len({int(t.item()) for t in every_dev}),
                     "launcher": os.environ.get("SEPKERN_BENCH_LAUNCHER", "external (torch.distributed.run)"),
                     "grad_allreduce": skdist.overlap_mode_name(),
                     "ms_per_step_by_rank": [round(1000.0 * t / args.steps, 3) for t in per_rank],
                     "allreduce_ms_per_step": round(ar_ms, 3), "allreduce_calls": len(ar),
                     "allreduce_bytes": int(model._engine.grad_full.numel() * 4),
                     "allreduce_busbw_GBs": round(2.0 * (world - 1) / world * model._engine.grad_full.numel() * 4 /
                                                  (ar_ms * 1e-3) / 1e9, 1) if ar_ms > 0 else None}
        dt = max(per_rank)
    frames_per_step = frames_timed / float(args.steps)                               # all ranks
    final_loss = float(loss_acc[0] / loss_acc[1])
    if not (final_loss == final_loss) or final_loss <= 0:
        sys.exit("bench: loss is not finite/positive (%r)" % final_loss)

    res = {
        "metric": "frames/sec uPIT BLSTM training on WSJ0-2mix-shaped synth" if args.arch == "upit" else
                  "utterance frames/sec RSH training on CHiME-5-shaped synth (each frame goes through num_spk passes)",
        "value": round(frames_per_step * args.steps / dt, 1),
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("RSH %dx%d BLSTM over [mix|attention], %d-spk (%d passes per step), " % (L, H, S, S)
                                if args.arch == "rsh" else "uPIT %dx%d BLSTM, %d-spk, " % (L, H, S)) +
                               "512-pt STFT (257 bins), batch %s per GPU, fwd + %s + bwd + clip 0.25 + Adam, random-init weights" %
                               (("%d utterances of U(24000, 64000) samples = 188..501 frames (mean %.0f valid frames per step, "
                                 "longest %.0f on average), a different batch every step, %s" %
                                 (B, frames_per_step / world, T_mean, "zero-PADDED rows (diagnostic)" if args.padded_rows else "packed rows"))
                                if args.ragged else "%d x %d frames" % (B, T),
                                "greedy-assignment MSE" if args.arch == "rsh" else "PIT-MSE"),
                   "numerics": numerics_note(model, args),
                   "global_batch": B * world, "frames_per_step": round(frames_per_step, 1),
                   "parallelism": "dp%d" % world if world > 1 else "single",
                   "mean_loss": round(final_loss, 6)},
    }
    if dist_info:
        res["distributed"] = dist_info
    if lstm_fallback:
        res["lstm_fallback"] = lstm_fallback
    # HBM-bound classes recorded in the timed region (the PIT loss kernels run in every step): bytes, not FLOP
    streaming = {k: prof.pop(k) for k in list(prof) if k.split("@")[0] in STREAM_CLASSES}
    if streaming:
        res["streaming_in_step"] = {k: stream_row(v) for k, v in streaming.items()}
    if prof_alone:
        for k in list(prof_alone):
            if k.split("@")[0] in STREAM_CLASSES:
                del prof_alone[k]
    if prof:
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS     # the dtype's own dense MFMA peak

        def pipe_peak(cls):
            """The peak of the matrix pipe a kernel class runs on (fp32-equivalent TFLOP/s)."""
            if args.dtype == "bf16":
                return PEAK_BF16_MFMA_TFLOPS
            if cls.startswith("gemm_f32_split") or (cls.startswith("lstm_fwd") and model._engine is not None and model._engine.split3_fwd):
                return PEAK_F32_SPLIT_TFLOPS
            return PEAK_F32_MFMA_TFLOPS
        # launches recorded on the side stream are the weight-gradient GEMMs that the engine co-schedules with the
        # next layer's recurrence on the same CUs (sepkern/engine.py): that shortens the step but lengthens THEIR
        # durations, so the figure over all launches is reported next to the one over the launches that had the
        # device to themselves
        merged = {}
        for k, v in prof.items():
            b = k.split("@")[0]
            m = merged.setdefault(b, [0, 0.0, 0.0])
            m[0] += v[0]; m[1] += v[1]; m[2] += v[2]
        # the dominant GEMM class by launch time: fp32 runs split products (bf16 pipe) wherever the operands are aligned and
        # fp32-MFMA kernels elsewhere; bf16 the NT kernel (hidden sizes that are no multiple of 8: the r01 kernel)
        if args.dtype == "bf16":
            kname = "gemm_bf16_nt_kernel" if "gemm_bf16_nt_kernel" in merged else "gemm_bf16_kernel"
        else:
            kname = max((k for k in merged if k.startswith("gemm_f32")), key=lambda k: merged[k][1])
        n, ms, fl = merged[kname]
        ach = fl / (ms * 1e-3) / 1e12
        kpeak = pipe_peak(kname)
        res["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2),
                           "peak": round(kpeak, 1), "unit": "TFLOP/s", "frac": round(ach / kpeak, 4),
                           "traffic": pmc_traffic(kname), "launches_per_step": n // args.steps,
                           "avg_launch_ms": round(ms / n, 4), "ms_per_step": round(ms / args.steps, 3)}
        if kname.startswith("gemm_f32_split"):
            res["roofline"]["peak_note"] = ("fp32 products formed on the bf16 matrix pipe (three-way split of both operands, six bf16 piece "
                                            "products per fp32 product): peak = the bf16 dense MFMA peak / 6 in fp32-equivalent FLOP; achieved "
                                            "= 2MNK / time.  The fp32-MFMA pipe (157.3 TFLOP/s) does not bound these launches")
            res["roofline"]["achieved_over_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 4)
        if prof_alone and kname in prof_alone:
            n1, ms1, fl1 = prof_alone[kname]
            ach1 = fl1 / (ms1 * 1e-3) / 1e12
            res["roofline"]["standalone"] = {
                "achieved": round(ach1, 2), "frac": round(ach1 / kpeak, 4), "avg_launch_ms": round(ms1 / n1, 4),
                "note": "same launches in an untimed pass of 2 steps with co-scheduling off: in the timed region the "
                        "weight-gradient launches of every layer but the first run co-resident with the next recurrence "
                        "on its CUs, which shortens the step and lengthens the launches it overlaps"}
        res["kernels"] = {k: {"launches_per_step": v[0] // args.steps, "ms_per_step": round(v[1] / args.steps, 3),
                              "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in prof.items()}
        # Every MFMA-bound kernel class against the same roofline, largest share of the step first, so that this line and
        # the first rows of profiles/*_summary.txt (rocprofv3 --stats of the same command) name the same kernels.  The
        # recurrences carry a second bound: a time step cannot be shorter than its products at the pipe's rate
        # (mfma_floor_us: 2 waves per SIMD x 4H*H/(16*16*4*... ) MFMAs of 32 cycles) plus one cross-CU hand-off of h_t / dG_t
        # (handoff_us = what the step takes beyond that floor: store -> drain -> flag -> poll -> LDS-DMA).
        by = {}
        for k, v in sorted(merged.items(), key=lambda kv: -kv[1][1]):
            if v[2] <= 0:
                continue
            a_ = v[2] / (v[1] * 1e-3) / 1e12
            row = {"launches_per_step": v[0] // args.steps, "avg_launch_ms": round(v[1] / v[0], 4),
                   "ms_per_step": round(v[1] / args.steps, 3), "achieved": round(a_, 2), "peak": round(pipe_peak(k), 1),
                   "frac": round(a_ / pipe_peak(k), 4)}
            if k.startswith("lstm_"):
                us = 1e3 * v[1] / v[0] / T_mean
                # per workgroup and time step: 16 batch rows x 64 gate columns x H, on 4 SIMDs, at the rate of the pipe the
                # products run on (FLOP per clock and SIMD: fp32 MFMA 64, six bf16 piece products 1024 / 6, bf16 1024)
                rate = 1024.0 if args.dtype == "bf16" else (1024.0 / 6.0 if pipe_peak(k) == PEAK_F32_SPLIT_TFLOPS else 64.0)
                floor = (2.0 * 16 * 64 * H / 4) / rate / 2.4e3
                row.update({"us_per_time_step": round(us, 3), "mfma_floor_us": round(floor, 3),
                            "handoff_us": round(us - floor, 3)})
            by[k] = row
        res["roofline"]["by_kernel"] = by
        res["roofline"]["top_kernel"] = next(iter(by)) if by else None
        # whole-step figure against the same roofline: 6 x MACs per frame
        if args.arch == "rsh":      # per pass: I = 2F inputs, F outputs; num_spk passes per frame
            P = S * (sum(2 * 4 * H * ((514 if l == 0 else 2 * H) + H) for l in range(L)) + 2 * H * 257)
        else:
            P = sum(2 * 4 * H * ((257 if l == 0 else 2 * H) + H) for l in range(L)) + 2 * H * 257 * S
        res["step_tflops"] = round(6.0 * P * frames_per_step / world / (dt / args.steps) / 1e12, 2)
        # against the compute dtype's OWN dense MFMA peak (fp32: 157.3 TFLOP/s -- a yardstick, not a bound, for a step whose
        # large products run on the bf16 pipe as split products: see roofline.peak_note)
        res["step_frac_of_mfma_peak"] = round(res["step_tflops"] / peak, 4)
    return res, (pcms, T_mean)


STREAM_CLASSES = ("stft_kernel", "istft_kernel", "pit_fwd", "pit_bwd")


def stream_row(v):
    """(launches, ms, algorithmic bytes) of an HBM-bound class -> us per launch, GB/s, fraction of the HBM peak."""
    n, ms, by = v
    gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"launches": n, "us_per_launch": round(1e3 * ms / max(1, n), 2), "MB_algorithmic_per_launch": round(by / max(1, n) / 1e6, 2),
            "GBs_algorithmic": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4)}


# The other BASELINE.json configurations that fit one GPU, timed by the DEFAULT run after the headline's timed region (same
# process, same harness, their own model / optimizer / resident batches), so that the driver's one line witnesses them too:
#   ragged        SURVEY.md 8d's variable-length set (the WSJ0-2mix-SHAPED batches `north_star` names), fp32, packed rows
#   bf16_3spk     BASELINE configs[3]: 3-speaker uPIT (6-permutation PIT loss), bf16, 32 x 400
#   bf16_ragged   the same arithmetic on the variable-length set (2 speakers, as the ragged set is defined)
#   rsh_4spk      BASELINE configs[4]'s one-GPU shape: RSH 2 x 600, 4 speakers, 32 x 400
#   fp32_mfma     the headline workload in the reference's LITERAL arithmetic: every GEMM on the fp32-MFMA kernels (variants 8 / 1:
#                 fp32 products, accumulation rounded to nearest) and the plain fp32-MFMA product in the forward recurrence -- what
#                 nn.LSTM / nn.Linear on fp32 tensors compute (/root/reference/archs/uPIT.py:115,132,141) up to summation order
#   ref_default_2x600_b100   the reference's OWN default model and batch size (archs/uPIT.py:115-119: 2 x 600; run_train.sh:18 and
#                 steps/train_qsub.py:39-41: 100 utterances per batch), 400 frames each: the only path through "several batch groups
#                 per workgroup" (G = 3) of the persistent recurrences
SECONDARY = (
    ("ragged", dict(ragged=True)),
    ("bf16_3spk", dict(dtype="bf16", num_spk=3)),
    ("bf16_ragged", dict(dtype="bf16", ragged=True)),
    ("rsh_4spk", dict(arch="rsh", hidden=600, layers=2, num_spk=4)),
    ("fp32_mfma", dict(env={"SEPKERN_GEMM_VARIANTS": "8,1", "SEPKERN_LSTM_FWD": "0,1,1,0,0,0,0,0"})),
    ("ref_default_2x600_b100", dict(hidden=600, layers=2, batch=100)),
)
SECONDARY_STEPS, SECONDARY_WARMUP = 20, 3


def is_headline(args):
    """True when the command line asks for BASELINE configs[1] itself (no workload flag given)."""
    d = parse([])
    return all(getattr(args, k) == getattr(d, k) for k in
               ("hidden", "layers", "num_spk", "batch", "frames", "dtype", "ragged", "padded_rows", "arch"))


def secondary_workloads(args, env, only=None):
    """{name: {value, ms_per_step, frames_per_step, dtype, step_frac_of_mfma_peak, by_kernel, ...}} for SECONDARY.  A
    workload that fails is reported as {"error": ...}: the headline line is printed either way."""
    out = {}
    for name, over in SECONDARY:
        if only and name not in only:
            continue
        a = argparse.Namespace(**vars(args))
        a.steps, a.warmup, a.no_kernel_events, a.aux = SECONDARY_STEPS, SECONDARY_WARMUP, False, False
        for k, v in over.items():
            if k != "env":
                setattr(a, k, v)
        log("secondary workload %s" % name)
        # (the engine reads its switches when it is built: the workload's own environment holds for that long only)
        saved_env = {k: os.environ.get(k) for k in over.get("env", {})}
        os.environ.update(over.get("env", {}))
        try:
            r, _ = measure(a, env, standalone_pass=False)
        except (Exception, SystemExit) as e:            # noqa: BLE001 -- reported, not swallowed
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("secondary workload %s FAILED: %s" % (name, out[name]["error"]))
            continue
        finally:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            import gc
            gc.collect()
            env.torch.cuda.empty_cache()
        rf = r.get("roofline", {})
        out[name] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                     "warmup": r["warmup"], "frames_per_step": r["config"]["frames_per_step"], "dtype": r["dtype"],
                     "workload": r["config"]["workload"], "numerics": r["config"]["numerics"],
                     "mean_loss": r["config"]["mean_loss"], "step_tflops": r.get("step_tflops"),
                     "step_frac_of_mfma_peak": r.get("step_frac_of_mfma_peak"),
                     "roofline_kernel": rf.get("kernel"), "roofline_frac": rf.get("frac"),
                     "by_kernel": {k: {f: v[f] for f in ("ms_per_step", "achieved", "frac", "us_per_time_step", "handoff_us")
                                       if f in v} for k, v in rf.get("by_kernel", {}).items()}}
        if "lstm_fallback" in r:
            out[name]["lstm_fallback"] = r["lstm_fallback"]
        log("secondary workload %s: %.3f ms/step, %.0f frames/s" % (name, r["ms_per_step"], r["value"]))
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                           # never returns
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d under a launcher that started %d ranks" % (args.gpus, world))
    # which library: before anything touches the GPU.  The loader refuses a diagnostic build (sk_build_flags() != 0: timing-only /
    # ablation / tuning builds of csrc/Makefile's variant targets) unless SEPKERN_ALLOW_DIAGNOSTIC_LIB=1; with it, this program still
    # wants --diagnostic, and the line names the library and its flags either way.
    from sepkern import _lib
    _lib.load()
    library = _lib.library_info()
    if library["build_flags"] and not args.diagnostic:
        sys.exit("bench: %s is a diagnostic build (sk_build_flags() = 0x%x: %s); its numbers are not the product's -- pass "
                 "--diagnostic to time it anyway" % (library["path"], library["build_flags"], "; ".join(library["build_flag_names"])))
    # rehearsal on a 1-GPU box: SEPKERN_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 (use with
    # SEPKERN_DIST_BACKEND=gloo and SEPKERN_LSTM_MODE=2, since two processes cannot both keep a
    # persistent grid resident on one GPU)
    if os.environ.get("SEPKERN_BENCH_ONE_DEVICE") == "1":
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local)
    # rehearsals only: SEPKERN_LSTM_MODE_BY_RANK="0,2" gives every rank its own recurrence mode (one persistent grid beside
    # ranks that launch per step: the chunked gradient exchange then runs next to a persistent kernel on a one-GPU box)
    by_rank = os.environ.get("SEPKERN_LSTM_MODE_BY_RANK")
    if by_rank:
        os.environ["SEPKERN_LSTM_MODE"] = by_rank.split(",")[rank % len(by_rank.split(","))]
    from sepkern import dist as skdist
    skdist.init_from_env()                                # nccl (= RCCL over xGMI) unless SEPKERN_DIST_BACKEND says otherwise

    env = argparse.Namespace(torch=torch, dist=dist, skdist=skdist, world=world, rank=rank, local=local)
    res, aux_in = measure(args, env)
    res["library"] = library
    default_run = world == 1 and is_headline(args)
    if rank == 0 and world == 1 and not args.no_aux and (args.aux or default_run):
        from sepkern import ops
        res["aux"] = aux_kernels(torch, ops, aux_in[0], args.batch, args.num_spk)
    if rank == 0 and default_run and not args.no_power_probe and "roofline" in res:
        res["roofline"]["power_note"] = power_probe(torch)
    if world == 1 and not args.no_secondary and is_headline(args):
        res["secondary"] = secondary_workloads(args, env, only=[s for s in args.secondary_only.split(",") if s])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU baseline (bounded sample)")
        res["cpu_baseline"] = cpu_baseline(args.hidden, args.layers, args.num_spk, args.batch, args.frames)
        log("CPU baseline done: %s frames/s on %d threads" % (res["cpu_baseline"]["value"], res["cpu_baseline"]["cores"]))
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def kernel_source_id():
    """sha256 over the HIP sources the library is built from: ties a PMC pass to the code it measured (the GPU box has
    no .git, so a commit id cannot be read there)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(PKG, "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same bench, gfx950 read correction applied) -- the
    counters cannot be collected from inside this process.  The file records the id of the kernel sources it was
    measured on (and the commit); when that differs from the sources of THIS run the figure is stale: None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        if doc.get("kernel_source_id") != kernel_source_id():
            return None
        return doc["kernels"][kernel]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def numerics_note(model, args):
    """What the `dtype` field does not say (a reader of "f32" is owed this)."""
    eng = model._engine
    if args.dtype == "bf16":
        return "bf16 matrix-core inputs for every product incl. the recurrences, fp32 accumulate / state / optimizer"
    gemms = ("products with aligned operands (the large GEMMs: 99 % of the GEMM FLOP) are formed on the bf16 matrix pipe by the three-way "
             "bf16 split of both fp32 operands -- x = hi + mid + lo exactly, pieces by rounding (|mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|); "
             "the six piece products of relative size >= 2^-16, each exact, are added into fp32 accumulators (sign phases cancel the "
             "bf16 MFMA's truncation offset); the three left out are together <= 2^-23 |a||b| in the worst case: a single product may "
             "be off by ~1 ulp where an fp32 FMA is exact, on sums the error vs fp64 is not above the fp32-MFMA kernels' "
             "(tests/test_gpu_kernels.py, test_gpu_signed_error.py, test_gpu_fullsize.py); secondary.fp32_mfma = the same step on "
             "fp32-MFMA kernels throughout")
    if os.environ.get("SEPKERN_GEMM_SPLIT", "1") == "0" or (eng is not None and eng.var_main not in (0, 2, 9)):
        gemms = "GEMMs on the fp32-MFMA kernels"
    if eng is not None and eng.split3_fwd:
        return ("fp32 storage, fp32 accumulation, no operand perturbed; " + gemms + "; the forward recurrence forms h W_hh^T the same way "
                "(SEPKERN_LSTM_FWD=0,1,1,0,0,0,0,0 = plain fp32-MFMA product); backward recurrences: fp32-MFMA products")
    if eng is not None and eng.tagged_fwd and eng.lstm_mode != 2:
        return ("fp32 storage and accumulation; " + gemms + "; forward-recurrence hand-off 'tagged': the operand h entering h W_hh^T "
                "carries a 2-bit epoch in its low mantissa bits (<= 3 ulp), all stored values exact; SEPKERN_LSTM_FWD=0,1,1,0,0,0,0,0 = "
                "exact hand-off")
    return "fp32 storage and accumulation; " + gemms + "; recurrences: fp32-MFMA products, exact hand-off (flags)"


def aux_kernels(torch, ops, pcms, B, S, reps=50):
    """The HBM-bound kernels `north_star` names, at the workload's own shape: STFT (train layout, (S + 1) B utterances), mask-apply +
    iSTFT (S B source signals), PIT loss forward (pairwise SSE + S! sums + arg-min) and backward on the packed rows.  Each is
    enqueued `reps` times back to back between two HIP events on the launch stream (ops._timed); bytes = SURVEY 8d's algorithmic
    bytes per frame.  Returns {name: {us_per_launch, MB_algorithmic_per_launch, GBs_algorithmic, frac_of_hbm_peak}}."""
    from sepkern.packing import Packing
    wavs = [torch.from_numpy(p[k]).cuda() for p in pcms for k in range(S + 1)]
    specs = ops.stft_batch([torch.from_numpy(p[0]).cuda() for p in pcms], want_complex=True, layout="FT")
    masks = [[torch.rand(257, sp.shape[1], device="cuda") for _ in range(S)] for sp in specs]
    frames = [int(sp.shape[1]) for sp in specs]
    pk = Packing.from_lens(sorted(frames, reverse=True), "cuda")
    F = 257
    mix = torch.rand(pk.Rp, F, device="cuda")
    srcs = [torch.rand(pk.Rp, F, device="cuda") for _ in range(S)]
    mask = torch.rand(pk.Rp, S * F, device="cuda")
    gscale = torch.ones(1, device="cuda")

    def run():
        ops.stft_batch(wavs, want_complex=False, layout="TF", repeat=reps)
        ops.mask_istft(specs, masks, want_float=False, repeat=reps)
        fw = ops.pit_mse_fwd(mask, mix, srcs, pk.lens, packing=pk, repeat=reps)
        ops.pit_mse_bwd(mask, mix, srcs, fw["best_perm"], fw["out"], gscale, packing=pk, repeat=reps)
    run()                                                  # warm-up (allocations, descriptor uploads)
    torch.cuda.synchronize()
    ops.PROF = {}
    run()
    torch.cuda.synchronize()
    prof = ops.prof_summary()
    ops.PROF = None
    out = {}
    for cls, (n, ms, by) in prof.items():
        out[cls] = stream_row((n * reps, ms, by))
    out["note"] = ("%d launches each, back to back between HIP events; bytes = algorithmic (SURVEY 8d): STFT 128 int16 samples in + 257 fp32 "
                   "bins out per frame; mask-iSTFT 257 complex64 + 257 fp32 in, 128 int16 out per frame and source; PIT forward (2S+1) F fp32 "
                   "per frame, backward (3S+1) F" % reps)
    return out


def power_probe(torch, seconds=2.5):
    """The dominant GEMM class is POWER-bound (r05: 181 TFLOP/s sustained at 1.89 GHz and 1375 W of the 1400 W cap, where the same
    kernel runs 186-212 for a few launches): a sustained run of the projection-shaped product on the planes kernel while the
    device's own sensors are read from sysfs (no child process): package power, its cap, and the shader clock.  Returns a dict for
    roofline.power_note -- the practical ceiling of the class next to `peak`."""
    import glob
    import threading
    from sepkern import ops
    note = {"measured_r05": "181 TFLOP/s sustained at 1.89 GHz, 1375 W of the 1400 W cap (profiles/r05_gemm_split6.txt): the split-product "
                            "GEMMs are power-bound; ~0.45-0.5 of the bf16-pipe peak / 6 is the practical ceiling of this class"}
    try:
        M, N, K = 12800, 7168, 1792
        A = torch.randn(M, K, device="cuda")
        Bm = torch.randn(N, K, device="cuda")
        C = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm(A, Bm, C, M, N, K, K, K, N, transB=True)
        torch.cuda.synchronize()
        hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))

        def read(path):
            try:
                with open(path) as f:
                    return float(f.read().strip())
            except (OSError, ValueError):
                return None
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                row = []
                for h in hw:
                    pw = read(h + "/power1_average")
                    pw = read(h + "/power1_input") if pw is None else pw
                    row.append((pw, read(h + "/freq1_input"), read(h + "/power1_cap")))
                samples.append(row)
                time.sleep(0.2)
        th = threading.Thread(target=sampler)
        th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.time(), 0
        e0.record()
        while time.time() - t0 < seconds:
            for _ in range(40):
                ops.gemm(A, Bm, C, M, N, K, K, K, N, transB=True)
            n += 40
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        ms = e0.elapsed_time(e1) / n
        note["sustained"] = {"shape": "12800 x 7168 x 1792 N/T (a layer's input projection)", "launches": n, "seconds": round(time.time() - t0, 2),
                             "tflops_fp32_equivalent": round(2.0 * M * N * K / ms / 1e9, 1),
                             "frac_of_split_peak": round(2.0 * M * N * K / ms / 1e9 / PEAK_F32_SPLIT_TFLOPS, 4)}
        # the card under load: the hwmon whose power reading is highest in the second half of the run
        if hw and samples:
            late = samples[len(samples) // 2:]
            best, best_w = None, -1.0
            for i in range(len(hw)):
                ws = [r[i][0] for r in late if r[i][0] is not None]
                if ws and sum(ws) / len(ws) > best_w:
                    best, best_w = i, sum(ws) / len(ws)
            if best is not None:
                fr = [r[best][1] for r in late if r[best][1] is not None]
                cap = [r[best][2] for r in late if r[best][2] is not None]
                note["sensors"] = {"watts": round(best_w / 1e6, 1), "sclk_mhz": round(sum(fr) / len(fr) / 1e6, 1) if fr else None,
                                   "cap_watts": round(cap[0] / 1e6, 1) if cap else None, "samples": len(late), "source": hw[best]}
        if "sensors" not in note:
            note["sensors"] = None
    except Exception as e:                                # noqa: BLE001 -- a probe must not cost the line
        note["error"] = "%s: %s" % (type(e).__name__, e)
    return note


if __name__ == "__main__":
    main()
