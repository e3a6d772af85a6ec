#!/usr/bin/env python
"""Headline benchmark: training throughput (audio-seconds per second) of the full hot path --
fused mel frontend -> encoder forward/backward -> fused CTC loss/grad -> clip + Adam -- on
BASELINE.json configs[1]: synthetic 16 kHz audio, batch 32 x 15 s per GPU, 80 mels, SentencePiece-
sized vocabulary (999 pieces + blank), bf16 activations.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          # starts N rank processes itself (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Inputs (waveforms, labels) are resident in HBM before the timed
region; the SpecAugment draws (a [B,4] int32 tensor) are made on the host every step like the
reference does.  `roofline` is measured live with HIP events on the kernels' own stream;
`cpu_baseline` times the CPU oracle (oracle/) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
# HBM-side bytes per launch from the PMC counters at exactly these sizes (C2, bf16), collected with
# `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes over tools/pmc_kernels.py and
# corrected as MI355X_MICROARCH.md "HBM" prescribes (FETCH_SIZE x2 for 16-byte-per-lane reads on
# gfx950, KiB units): profiles/r01_pmc_traffic.md holds the raw numbers.  None = not collected.
PMC_TRAFFIC_BYTES = {
    ("sru_bwd_kernel", "bf16", 32, 15.0): 2 * 144257.0 * 1024 + 193024.0 * 1024,
    ("sru_fwd_kernel", "bf16", 32, 15.0): 2 * 96184.9 * 1024 + 96128.0 * 1024,
    # round 3 (profiles/r03_gemm_pmc.md, tools/pmc_gemm.py, per launch): bytes that left the L2s -- the Infinity Cache
    # serves part of them, the counters cannot tell.  Forward: A (49 MB) is re-streamed by every XCD once per group
    # of four tile columns, B (6 MB) once per XCD and group; algorithmic 55.5 MB read + 147.7 MB written.
    ("gemm_nt(forward projection U = LN(x) W)", "bf16", 32, 15.0): 2 * 127609.6 * 1024 + 144248.8 * 1024,
    ("gemm_nt(data gradient gX = gU W^T)", "bf16", 32, 15.0): 2 * 123362.9 * 1024 + 48064.0 * 1024,
    ("gemm_tn(weight gradient dW = LN(x)^T gU, split-K)", "bf16", 32, 15.0): 2 * 137164.3 * 1024 + 61440.5 * 1024,
    # round 2 (profiles/r02_pmc_traffic.md; kernel unchanged since)
    ("melspec_kernel", "bf16", 32, 15.0): 2 * 16019.0 * 1024 + 18300.0 * 1024,
    # round 3 (profiles/r03_pmc_traffic.md): rows + lattice / dense half (16-byte loads: x2) + sparse half (4-byte loads: raw
    # counter, a lower bound), packed lattice rows
    ("ctc_fwd_bwd(3 kernels)", "bf16", 32, 15.0): (2 * 48157.0 + 2 * 52896.0 + 13632.0 + 4976.0 + 112238.0 + 29633.0) * 1024,
}
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--seconds", type=float, default=15.0)
    p.add_argument("--vocab", choices=["sp", "char"], default="sp")
    p.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--kernel-iters", type=int, default=20)
    return p.parse_args()


def _config_name(B, seconds):
    """Which BASELINE.json config the arguments are."""
    if B == 32 and seconds == 15.0:
        return "configs[1]"
    if B == 64 and seconds == 30.0:
        return "configs[3] (long-form)"
    return "custom shape"


def synth_batch(B, N, V, seconds, device, vocab):
    """0.1*randn clamp + one sine per utterance; labels U{1..V-1}, length
    U[0.5,1] * (15 chars/s or 4 pieces/s) * duration, some adjacent repeats forced."""
    g = torch.Generator(device="cpu").manual_seed(7)
    t = torch.arange(N, dtype=torch.float32) / 16000.0
    freqs = torch.randint(80, 4000, (B, 1), generator=g).float()
    wav = (0.1 * torch.randn(B, N, generator=g) + 0.2 * torch.sin(2 * torch.pi * freqs * t)).clamp(-1, 1)
    per_s = 15.0 if vocab == "char" else 4.0
    Lmax = int(per_s * seconds)
    lens = (torch.rand(B, generator=g) * 0.5 + 0.5).mul(Lmax).long().clamp(min=1)
    lens[0] = Lmax
    labels = torch.randint(1, V, (B, Lmax), generator=g)
    labels[:, 1::9] = labels[:, 0::9][:, : labels[:, 1::9].shape[1]]
    offs = torch.arange(B + 1, dtype=torch.int64) * N
    return (wav.reshape(-1).to(device), offs.to(device), labels.to(device),
            lens.to(torch.int32).to(device), Lmax)


def time_kernel(fn, iters, spacer=None):
    """Average device time of fn() in ms, HIP events on the current stream (the stream the C-ABI
    launches use: speech_amd passes torch's current stream).  With `spacer` every launch is timed by its
    own event pair and an HBM-bound launch runs between two of them: back-to-back dense-MFMA launches
    pull the clock down within milliseconds, which the training step (where every GEMM sits between
    streaming kernels) never sees."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if spacer is None:
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            fn()
        end.record()
        torch.cuda.synchronize()
        return start.elapsed_time(end) / iters
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in pairs:
        spacer()
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / iters


def kernel_rooflines(args, model, wav, offs, labels, lens, Lmax, V, iters):
    """Per-kernel HBM roofline lines for the hand-written north-star kernels, measured live: the
    C-ABI entry points are called directly on pre-allocated buffers (no host-side checks, copies or
    allocations inside the timed loop), HIP events on the stream the kernels are launched on."""
    from speech_amd import _lib, ops
    from speech_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    dev = wav.device
    B, N = args.batch, int(args.seconds * 16000)
    out = {}

    def line(name, by, ms, launches):
        out[name] = {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": by / ms / 1e6 / HBM_PEAK_GBS, "ms": ms, "algorithmic_bytes": by,
                     "launches_per_step": launches}

    # frontend: 4*sum(N) + 4*80*B*Tmax bytes
    masks = torch.tensor([[10, 10, 100, 30]] * B, dtype=torch.int32, device=dev)
    T = 1 + N // 160
    spec_buf = torch.empty((B, 1, 80, T), dtype=torch.float32, device=dev)
    plan = ops.default_plan(dev)
    st = stream_ptr()
    ms = time_kernel(lambda: check(L.sp_melspec_fwd(ptr(wav), ptr(offs), B, T, ptr(plan.buf), ptr(masks), 0,
                                                    ptr(spec_buf), st), "mel"), iters)
    line("melspec_kernel", 4.0 * B * N + 4.0 * 80 * B * T, ms, 1)
    Tp = (T - 1) // 2 + 1
    logits = torch.randn(B, Tp, V, device=dev) * 3
    il = torch.full((B,), Tp, dtype=torch.int32, device=dev)
    nll = torch.empty(B, device=dev); loss = torch.empty(1, device=dev); grad = torch.empty_like(logits)
    ws = torch.empty(L.sp_ctc_workspace_bytes(Tp, B, V, Lmax), dtype=torch.uint8, device=dev)
    ms = time_kernel(lambda: check(L.sp_ctc_fwd_bwd(ptr(logits), logits.stride(1), logits.stride(0), ptr(labels),
                                                    labels.stride(0), ptr(il), ptr(lens), Tp, B, V, 0, Lmax, 1.0,
                                                    ptr(nll), ptr(loss), ptr(grad), 0, ptr(ws), ws.numel(), st), "ctc"), iters)
    S = (2 * lens.float() + 1).sum().item()
    # SURVEY 8(d)'s algorithmic figure: scores read + gradient written + an f32 alpha lattice written and read
    # (the kernels spill f64 alpha AND beta -- 4x that term -- for the precision DESIGN.md section 4 explains;
    # the roofline fraction is quoted against the algorithmic bytes, so that choice costs fraction, as it should)
    line("ctc_fwd_bwd(3 kernels)", 2 * 4.0 * Tp * B * V + 2 * 4.0 * Tp * S, ms, 1)
    # the form the training step runs: time-major logits, gradient stored as bf16 for the classifier's bf16 GEMM
    logits_t = logits.transpose(0, 1).contiguous()
    grad16 = torch.empty_like(logits_t, dtype=torch.bfloat16)
    ms = time_kernel(lambda: check(L.sp_ctc_fwd_bwd(ptr(logits_t), logits_t.stride(0), logits_t.stride(1), ptr(labels),
                                                    labels.stride(0), ptr(il), ptr(lens), Tp, B, V, 0, Lmax, 1.0,
                                                    ptr(nll), ptr(loss), ptr(grad16), 2, ptr(ws), ws.numel(), st), "ctc"), iters)
    line("ctc_fwd_bwd(as in the step: time-major, bf16 gradient)", (4.0 + 2.0) * Tp * B * V + 2 * 4.0 * Tp * S, ms, 0)   # counted above
    del logits_t, grad16
    # decode: scores read + ids write
    ids = torch.empty((B, Tp), dtype=torch.int32, device=dev); nids = torch.empty(B, dtype=torch.int32, device=dev)
    ms = time_kernel(lambda: check(L.sp_greedy_decode(ptr(logits), _lib.dtype_code(logits), B, Tp, V, 0, None, ptr(ids),
                                                      ptr(nids), st), "decode"), iters)
    line("greedy_decode(2 kernels)", 4.0 * B * Tp * V + 4.0 * B * Tp, ms, 0)   # eval path only
    # SRU recurrence, one k=3 layer: forward reads U + x', writes h + c; backward reads U, x', c, gh, writes gU, gx
    d = 512
    e = 2 if args.dtype == "bf16" else 4
    adt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dc = _lib.dtype_code(torch.empty(0, dtype=adt))
    U = torch.randn(Tp, B, 3, 2, d, device=dev).to(adt)
    xr = torch.randn(Tp, B, 2 * d, device=dev).to(adt)
    wc = torch.randn(4 * d, device=dev) * 0.5
    bias = torch.zeros(4 * d, device=dev)
    h = torch.empty(Tp, B, 2 * d, device=dev, dtype=adt); c = torch.empty(Tp, B, 2 * d, device=dev, dtype=adt)
    ms = time_kernel(lambda: check(L.sp_sru_fwd(ptr(U), ptr(xr), ptr(wc), ptr(bias), None, Tp, B, d, 3, dc, ptr(h), ptr(c), st),
                                   "sru_fwd"), iters)
    line("sru_fwd_kernel", Tp * B * 2 * d * (3 * e + e + e + e), ms, 10)          # c is saved in the activation dtype
    gh = torch.randn_like(h); gU = torch.empty_like(U); gx = torch.empty_like(xr)
    gwc = torch.zeros(4 * d, device=dev); gb = torch.zeros(4 * d, device=dev)
    ms = time_kernel(lambda: check(L.sp_sru_bwd(ptr(U), ptr(xr), ptr(wc), ptr(bias), None, ptr(c), ptr(gh), Tp, B, d, 3, dc,
                                                ptr(gU), ptr(gx), ptr(gwc), ptr(gb), st), "sru_bwd"), iters)
    line("sru_bwd_kernel", Tp * B * 2 * d * (3 * e + e + e + e + 3 * e + e), ms, 10)
    return out


def _in_step_events(step_fn, name, key, n_steps=2):
    """Brackets every call of speech_amd.ops.<name> made by `n_steps` extra (untimed) training steps with
    its own HIP event pair on the launch stream; returns {key(args): [ms, ...]}."""
    from speech_amd import ops
    rec, orig = [], getattr(ops, name)

    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **kw)
        e1.record()
        rec.append((key(*a, **kw), e0, e1))
        return r
    setattr(ops, name, timed)
    try:
        for i in range(n_steps):
            step_fn(10 ** 6 + i)
        torch.cuda.synchronize()
    finally:
        setattr(ops, name, orig)
    out = {}
    for k, e0, e1 in rec:
        out.setdefault(k, []).append(e0.elapsed_time(e1))
    return out


def gemm_in_step(step_fn, n_steps=2):
    """Live durations of the projection GEMMs INSIDE the training step, one HIP event pair per launch on the
    launch stream (a back-to-back microbenchmark of a dense-MFMA kernel is not representative: the clock sags
    within milliseconds of continuous MFMA load, and the caches hold other data than in the step).  Returns
    {name: roofline line} for the three 1024 <-> 3072 products of the nine k = 3 SRU layers."""
    def key(a, b, out=None, out_dtype=None, accumulate=False, split_k=1):
        form = "nt" if a.stride(-1) == 1 and b.stride(-1) == 1 else ("tn" if a.stride(0) == 1 and b.stride(0) == 1 else "other")
        return (form, a.shape[0], b.shape[0], a.shape[1])
    by_shape = _in_step_events(step_fn, "gemm", key, n_steps)
    names = {("nt", 3072, 1024): "gemm_nt(forward projection U = LN(x) W)", ("nt", 1024, 3072): "gemm_nt(data gradient gX = gU W^T)",
             ("tn", 3072, None): "gemm_tn(weight gradient dW = LN(x)^T gU, split-K)"}
    out = {}
    for (form, M, N, K), ts in by_shape.items():
        name = names.get((form, N, K)) or (names.get((form, N, None)) if form == "tn" and M == 1024 else None)
        if name is None:
            continue
        ms, fl = sum(ts) / len(ts), 2.0 * M * N * K
        out[name] = {"bound": "mfma", "achieved": fl / ms / 1e9, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                     "frac": fl / ms / 1e9 / MFMA_BF16_PEAK_TF, "ms": ms, "algorithmic_flops": fl,
                     "algorithmic_bytes": 2 * (M * K + N * K + M * N), "launches_per_step": len(ts) // n_steps,
                     "shape_mnk": [M, N, K], "timed": "in-step, one HIP event pair per launch"}
    return out


def sru_in_step(step_fn, which, n_steps=2):
    """Average duration of the k = 3 SRU recurrence launches (which = "sru_fwd" / "sru_bwd") inside the
    training step (the event pair brackets the one kernel the call launches)."""
    if which == "sru_bwd":
        key = lambda U, xres, wc, bias, mask, c, gh, T, B, d, k, *r, **kw: k      # noqa: E731
    else:
        key = lambda U, xres, wc, bias, mask, T, B, d, k, *r, **kw: k            # noqa: E731
    ts = _in_step_events(step_fn, which, key, n_steps).get(3)
    return (sum(ts) / len(ts), len(ts) // n_steps) if ts else None


def cpu_baseline(args, V):
    """The CPU oracle (oracle.mel + oracle.net + torch CPU CTC + torch Adam: the reference's own
    CPU arithmetic) on a bounded sample: ONE training step on 2 utterances of the same length."""
    from oracle import mel as omel
    from oracle import net as onet
    from speech_amd import net as pnet
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    Bs = 2
    N = int(args.seconds * 16000)
    g = torch.Generator().manual_seed(7)
    wavs = [(0.1 * torch.randn(1, N, generator=g)).clamp(-1, 1) for _ in range(Bs)]
    L = int((15.0 if args.vocab == "char" else 4.0) * args.seconds)
    labels = torch.randint(1, V, (Bs, L), generator=g)
    ll = torch.full((Bs,), L, dtype=torch.int32)
    model = pnet.SRModel(10, 512, V - 1, 80, 0.1, 0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    t0 = time.time()
    spec = omel.collate(wavs, masks=[(10, 10, 100, 30)] * Bs)
    loss, grads, _ = onet.loss_and_grads(sd, spec, labels, ll, 10, 512, training=True)
    params = [sd[k].requires_grad_(False) for k in grads]
    opt = torch.optim.Adam([torch.nn.Parameter(p) for p in params], lr=3e-4)
    for p, k in zip(opt.param_groups[0]["params"], grads):
        p.grad = grads[k]
    torch.nn.utils.clip_grad_norm_(opt.param_groups[0]["params"], 2.0)
    opt.step()
    dt = time.time() - t0
    return {"value": Bs * args.seconds / dt, "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"1 training step on {Bs} x {args.seconds:g} s utterances (fp32, CPU oracle: torch.stft mel, "
                      f"torch conv/BN/GEMM, python-loop SRU, torch CTC, torch Adam), {dt:.1f} s wall",
            "parts": cpu_parts(args, V, cores)}


def cpu_parts(args, V, cores):
    """the north-star stages on the host alone, as the reference runs them -- (a) the frontend one
    utterance at a time like collate_fn (oracle.mel = torch.stft + filterbank matmul), (b) F.ctc_loss forward +
    backward on [T', B, V] log-probs, (c) argmax + the Python collapse loop of decoder.py.  Bounded: 8 utterances for
    (a), the bench's batch for (b), 64 utterances for (c); median of 3 after a warm-up."""
    import statistics
    from oracle import decode as odec
    from oracle import mel as omel

    def med(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    g = torch.Generator().manual_seed(11)
    N = int(args.seconds * 16000)
    B = args.batch
    Tp = ((1 + N // 160) - 1) // 2 + 1
    wavs = [(0.1 * torch.randn(1, N, generator=g)).clamp(-1, 1) for _ in range(8)]
    t_mel = med(lambda: omel.collate(wavs))
    L = int((15.0 if args.vocab == "char" else 4.0) * args.seconds)
    logits = (torch.randn(Tp, B, V, generator=g) * 3)
    labels = torch.randint(1, V, (B, L), generator=g)
    il = torch.full((B,), Tp, dtype=torch.int32)
    ll = torch.full((B,), L, dtype=torch.int32)

    def ctc():
        z = logits.clone().requires_grad_(True)
        torch.nn.functional.ctc_loss(torch.log_softmax(z, 2), labels, il, ll, blank=0).backward()
    t_ctc = med(ctc)
    nd = min(64, 2 * B)
    dlog = torch.randn(nd, Tp, V, generator=g)
    t_dec = med(lambda: odec.greedy_decode_ids(dlog))
    return {"frontend_audio_s_per_s": 8 * args.seconds / t_mel,
            "ctc_fwd_bwd_ms": 1e3 * t_ctc, "ctc_shape_tbv": [Tp, B, V],
            "greedy_decode_utt_per_s": nd / t_dec, "decode_shape_btv": [nd, Tp, V], "threads": cores}


def comm_profile(dp, opt, step, args, world, ms_step_dp):
    """The gradient exchange of the data-parallel step, measured after the timed region (all ranks take part):
    every bucket's all-reduce alone (HIP events on the launch stream around a blocking collective, median of 5),
    and the step without any exchange (each rank stepping on its own gradients).  exposed = step with the exchange
    - step without; overlap_frac = the share of the stand-alone all-reduce time that backward hid."""
    from speech_amd.train import all_reduce_sum
    per_bucket = []
    for lo, hi, _ in dp.buckets:
        view = opt.flat_grad[lo:hi]
        ts = []
        for _r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            all_reduce_sum(view, dp.group)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        per_bucket.append(ts[len(ts) // 2])
    # the step with the exchange switched off (replicas drift apart from here on: nothing is measured after this)
    launch, finish = dp._launch, dp.finish
    dp._launch = lambda b: None
    try:
        for i in range(2):
            step(2 * 10 ** 6 + i)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(max(args.steps // 2, 2)):
            step(2 * 10 ** 6 + 2 + i)
        torch.cuda.synchronize()
        ms_nocomm = 1000.0 * (time.perf_counter() - t0) / max(args.steps // 2, 2)
    finally:
        dp._launch, dp.finish = launch, finish
    tm = torch.tensor([ms_nocomm], dtype=torch.float64, device=opt.flat_grad.device)
    all_reduce_sum(tm, dp.group, dist.ReduceOp.MAX)
    ms_nocomm = float(tm.item())
    total = sum(per_bucket)
    exposed = max(ms_step_dp - ms_nocomm, 0.0)
    return {"buckets_mb": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi, _ in dp.buckets],
            "allreduce_ms": [round(t, 3) for t in per_bucket], "allreduce_ms_total": total,
            "ms_per_step_without_exchange": ms_nocomm, "exposed_ms": exposed,
            "overlap_frac": (1.0 - exposed / total) if total > 0 else None,
            "backend": dist.get_backend(dp.group)}


def spawn_ranks(n, deadline_s=None, log_dir=None, cmd=None):
    """`python bench.py --gpus N` typed as is: start N fresh rank processes of this script (one per GPU, the
    launcher's environment contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and relay rank 0's
    JSON line.  Nothing in this process has touched the GPU, and nothing is exec'ed: the ranks are children.
    Fails fast: the children are polled; the first one that exits non-zero (a rank that dies at RCCL init leaves the
    others blocked in a collective) ends the rest, and so does the overall deadline (BENCH_DEADLINE_S, default 540 s:
    below the driver's limit, so that a line about the failure still gets out).  Every rank's stderr goes to its own
    file (BENCH_LOG_DIR, default gpurun_out/ or the temp dir) and the failing rank's tail is printed.
    cmd: the rank command line (default: this script with this process's arguments; the launcher's test passes its own)."""
    import signal
    import socket
    import subprocess
    import tempfile
    deadline_s = float(os.environ.get("BENCH_DEADLINE_S", "540")) if deadline_s is None else deadline_s
    log_dir = log_dir or os.environ.get("BENCH_LOG_DIR") or (
        os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else tempfile.gettempdir())
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        path = os.path.join(log_dir, f"bench_rank{r}.stderr")
        logs.append(path)
        with open(path, "wb") as err:
            procs.append(subprocess.Popen(cmd or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]), env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL, stderr=err,
                                          start_new_session=True))      # its own process group: ended as a whole

    def end_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        t_end = time.time() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                q.wait()

    t0, rc, why = time.time(), 0, None
    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc, why = bad[0][1], f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > deadline_s:
                rc, why = 124, f"deadline of {deadline_s:.0f} s passed with ranks {[r for r, c in enumerate(codes) if c is None]} still running"
                break
            time.sleep(0.05)
    finally:
        end_all()
    if why:
        sys.stderr.write(f"bench.py: {why}; the other ranks were terminated\n")
        for r, path in enumerate(logs):
            try:
                tail = open(path, "rb").read()[-1500:].decode("utf-8", "replace")
            except OSError:
                tail = ""
            if tail.strip():
                sys.stderr.write(f"--- rank {r} stderr (tail of {path}) ---\n{tail}\n")
    return rc if rc else 0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world == 1 and "RANK" not in os.environ and (args.gpus > 1 or os.environ.get("BENCH_FORCE_SPAWN")):
        raise SystemExit(spawn_ranks(args.gpus))      # BENCH_FORCE_SPAWN: exercise the launcher on a 1-GPU box
    # BENCH_SHARE_GPU / BENCH_BACKEND exist for the test that rehearses the N > 1 path on a one-GPU lease
    # (two ranks on one card over gloo: RCCL refuses two ranks per device); the driver never sets them
    if os.environ.get("BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from speech_amd import data, net, ops
    from speech_amd.train import BucketedAllReduce, FlatAdam, lr_lambda_linear_warmup, train_step

    V = 1000 if args.vocab == "sp" else 29
    B, N = args.batch, int(args.seconds * 16000)
    adt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(7)
    model = net.SRModel(10, 512, V - 1, 80, 0.1, 0, compute_dtype=adt).to(dev)
    model.train()
    opt = FlatAdam(model.parameters(), lr=3e-4, weight_decay=1e-3)
    dp = BucketedAllReduce(opt) if world > 1 else None
    torch.manual_seed(7 + rank)
    wav, offs, labels, lens, Lmax = synth_batch(B, N, V, args.seconds, dev, args.vocab)
    T = 1 + N // 160
    offs_host = offs.cpu()

    def step(i):
        masks = data.draw_masks([T] * B)                      # host RNG, like the reference
        spec = ops.melspec(wav, offs, masks=masks, offsets_host=offs_host)   # a loader knows the lengths on the host
        opt.param_groups[0]["lr"] = 3e-4 * lr_lambda_linear_warmup(i + 1, 3500, 10 ** 6)
        return train_step(model, opt, spec, labels, lens, 2.0, dp, max_label_len=Lmax)

    for i in range(args.warmup):
        loss = step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        from speech_amd.train import all_reduce_sum
        all_reduce_sum(tmax, None, dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(loss)

    # N > 1: the same per-kernel timings and the gradient exchange's cost.  Every rank runs the same extra
    # (untimed) steps and collectives in the same order; rank 0 reports.
    ks_dp, comm = None, None
    if world > 1:
        # (the headline number above is already taken: a failure in these extras is reported in the line, not raised)
        try:
            ks_dp = {}
            if args.dtype == "bf16":
                ks_dp.update(gemm_in_step(step))
                for which in ("sru_fwd", "sru_bwd"):
                    sb = sru_in_step(step, which)
                    if sb is not None:
                        e = 2
                        Tp_ = ((1 + N // 160) - 1) // 2 + 1
                        by = Tp_ * B * 1024 * ((3 * e + 3 * e) if which == "sru_fwd" else (3 * e + e + e + e + 3 * e + e))
                        ks_dp[which + "_kernel"] = {"bound": "hbm", "achieved": by / sb[0] / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": by / sb[0] / 1e6 / HBM_PEAK_GBS, "ms": sb[0], "algorithmic_bytes": by,
                                                    "launches_per_step": sb[1] + 1, "timed": "in-step, one HIP event pair per launch"}
            comm = comm_profile(dp, opt, step, args, world, 1000.0 * dt / args.steps)
        except Exception as ex:      # noqa: BLE001 -- every rank runs the same code on the same shapes
            comm = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        value = B * args.seconds * world * args.steps / dt
        line = {
            "metric": "train audio-sec/sec", "value": value, "unit": "audio-sec/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{_config_name(B, args.seconds)}: synthetic 16 kHz, batch {B} x {args.seconds:g} s per GPU, 80 mels, "
                                   f"{'sentencepiece-sized vocab (999+blank)' if args.vocab == 'sp' else 'char vocab (29)'}"
                                   ", full training step (mel frontend + SRModel fwd/bwd + CTC + clip + Adam)",
                       "global_batch": B * world, "seconds_per_utt": args.seconds, "vocab": V,
                       "parallelism": f"dp{world}", "final_loss": loss_val},
        }
        if world == 1:
            ks = kernel_rooflines(args, model, wav, offs, labels, lens, Lmax, V, args.kernel_iters)
            if args.dtype == "bf16":
                ks.update(gemm_in_step(step))
                for which in ("sru_fwd", "sru_bwd"):      # replace the stand-alone timings by the in-step ones
                    sb = sru_in_step(step, which)
                    if sb is not None:
                        k = ks[which + "_kernel"]
                        k["ms_standalone"], k["ms"], k["timed"] = k["ms"], sb[0], "in-step, one HIP event pair per launch"
                        k["achieved"] = k["algorithmic_bytes"] / sb[0] / 1e6
                        k["frac"] = k["achieved"] / HBM_PEAK_GBS
                        k["launches_per_step"] = sb[1] + 1      # + the k = 4 first layer
            # dominant = the kernel with the largest share of the training step, over ALL kernels timed here
            # (every kernel of the step is hand-written; the three projection GEMMs lead)
            dom = max(ks, key=lambda k: ks[k]["ms"] * ks[k]["launches_per_step"])
            r = dict(ks[dom])
            r["kernel"] = dom
            r["traffic"] = PMC_TRAFFIC_BYTES.get((dom, args.dtype, args.batch, float(args.seconds)))
            line["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}
            line["kernels"] = ks
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args, V)
        else:
            if ks_dp:
                dom = max(ks_dp, key=lambda k: ks_dp[k]["ms"] * ks_dp[k]["launches_per_step"])
                r = dict(ks_dp[dom])
                r["kernel"], r["traffic"] = dom, None
                line["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}
                line["kernels"] = ks_dp
            line["comm"] = comm
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
