"""GEMM lab: times sp_gemm on the training shapes (random bf16 operands, padded pitches as the model uses
them), every launch between two HBM-bound spacer launches (as in the training step: back-to-back dense-MFMA
launches pull the clock down), against an f32 reference and beside the vendor library's kernel on the same
operands.  Interleaved rounds in one process; median and minimum.
    python tools/gemm_lab.py [nt|tn|all] [rounds]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib, ops          # noqa: E402

dev = torch.device("cuda")
spacer_buf = torch.empty(64 << 20, dtype=torch.float32, device=dev)


def spacer():
    spacer_buf.mul_(1.0001)


def timed(fns, rounds=5, reps=4):
    res = {k: [] for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            ev = []
            for _ in range(reps):
                spacer()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            res[k] += [a.elapsed_time(b) * 1e3 for a, b in ev]
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def padded(rows, cols, scale=1.0, pad=64):
    buf = torch.empty(rows, cols + pad, device=dev, dtype=torch.bfloat16)
    buf[:, :cols] = (torch.randn(rows, cols, device=dev) * scale).bfloat16()
    return buf[:, :cols]


def report(name, shape, fl, t, errs):
    for k, (med, mn) in t.items():
        e = errs.get(k)
        err = "" if e is None else f"  relerr {e:.2e}" + ("  WRONG" if e > 1e-2 else "")
        print(f"{name:8s} {shape:18s} {k:10s} median {med:7.1f} us  min {mn:7.1f} us  {fl / med / 1e6:7.1f} TF/s{err}", flush=True)


def lab_nt(name, M, N, K, rounds, out_dtype=torch.bfloat16):
    a, b = padded(M, K), padded(N, K, 0.05)
    ref = a.float() @ b.float().t()
    c_own = torch.zeros(M, N, device=dev, dtype=out_dtype)
    c_lib = torch.zeros(M, N, device=dev, dtype=out_dtype)
    fns = {"sp_gemm": lambda: ops.gemm(a, b, out=c_own)}
    if out_dtype == torch.bfloat16:
        fns["library"] = lambda: torch.mm(a, b.t(), out=c_lib)
    else:
        fns["library"] = lambda: torch.mm(a, b.t(), out_dtype=torch.float32, out=c_lib)
    t = timed(fns, rounds)
    report(name, f"{M}x{N}x{K}", 2.0 * M * N * K, t, {"sp_gemm": float((c_own.float() - ref).abs().max()) / float(ref.abs().max())})


def lab_tn(name, Mi, Nj, K, rounds):
    x, gu = padded(K, Mi), padded(K, Nj, 0.05, pad=0)
    ref = x.float().t() @ gu.float()
    S = _lib.lib().sp_gemm_tn_splits(Mi, Nj, K, _lib.SP_BF16)
    P = torch.zeros(S, Mi, Nj, device=dev)
    Kc = K // 4 * 4
    xa, ga = x[:Kc].view(4, Kc // 4, Mi).transpose(1, 2), gu[:Kc].view(4, Kc // 4, Nj)
    fns = {"sp_gemm": lambda: ops.gemm(x.t(), gu.t(), out=P, out_dtype=torch.float32, split_k=S),
           "lib_bmm4": lambda: torch.bmm(xa, ga, out_dtype=torch.float32)}
    t = timed(fns, rounds)
    report(name, f"{Mi}x{Nj}x{K}/S{S}", 2.0 * Mi * Nj * K, t, {"sp_gemm": float((P.sum(0) - ref).abs().max()) / float(ref.abs().max())})


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    M = 24032
    if what in ("nt", "all"):
        lab_nt("fwd", M, 3072, 1024, rounds)
        lab_nt("dX", M, 1024, 3072, rounds)
        lab_nt("L0fwd", M, 4096, 1280, rounds)
        lab_nt("L0dX", M, 1280, 4096, rounds)
        lab_nt("logits", M, 1024, 1024, rounds, torch.float32)
    if what in ("tn", "all"):
        lab_tn("dW", 1024, 3072, M, rounds)
        lab_tn("L0dW", 1280, 4096, M, rounds)
        lab_tn("cls_dW", 1024, 1024, M, rounds)
