"""GEMM lab: times the projection GEMM kernels on the training shapes (random bf16 operands, padded
pitches as the model uses them), every launch between two HBM-bound spacer launches (as in the training
step), against an f32 reference and beside the vendor library's kernel on the same operands.
    python tools/gemm_lab.py [nt|tn|all] [rounds]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib, ops          # noqa: E402
from speech_amd._lib import check, ptr, stream_ptr          # noqa: E402

dev = torch.device("cuda")
L = _lib.lib()
H = ctypes.CDLL(_lib.LIB_PATH)
vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
H.sp_lab_gemm_nt.argtypes = [vp, vp, vp, ci, ci, ci, i64, i64, i64, ci, ci, vp]
if hasattr(H, "sp_lab_gemm_tn"):
    H.sp_lab_gemm_tn.argtypes = [vp, vp, vp, ci, ci, ci, i64, i64, ci, ci, vp]

spacer_buf = torch.empty(64 << 20, dtype=torch.float32, device=dev)


def spacer():
    spacer_buf.mul_(1.0001)


def timed(fns, rounds=5, reps=4):
    """fns: {name: callable}; interleaved rounds in one process; returns {name: (median_us, min_us)}."""
    res = {k: [] for k in fns}
    for k, f in fns.items():
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
    out = {}
    for k, v in res.items():
        v = sorted(v)
        out[k] = (v[len(v) // 2], v[0])
    return out


def padded(rows, cols, scale=1.0, pad=64):
    buf = torch.empty(rows, cols + pad, device=dev, dtype=torch.bfloat16)
    buf[:, :cols] = (torch.randn(rows, cols, device=dev) * scale).bfloat16()
    return buf[:, :cols]


def lab_nt(name, M, N, K, rounds, out_dtype=torch.bfloat16, variants=(0, 1, 2, 3)):
    a = padded(M, K)
    b = padded(N, K, 0.05)
    ref = a.float() @ b.float().t()
    fl = 2.0 * M * N * K
    fns = {}
    outs = {}
    bn = {0: 256, 1: 256, 2: 192, 3: 128}
    knobs = [tuple(int(x) for x in kv.split(":")) for kv in os.environ.get("LAB_KNOBS", "0:0").split(",")]
    for v in variants:
        if N % bn[v % 10]:
            continue
        for gc, nts in knobs:
            c = torch.zeros(M, N, device=dev, dtype=out_dtype)
            key = f"rs{v}/g{gc}n{nts}"
            outs[key] = c

            def run(v=v, c=c, gc=gc, nts=nts):
                H.sp_lab_set(0, gc); H.sp_lab_set(1, nts)
                check(H.sp_lab_gemm_nt(ptr(a), ptr(b), ptr(c), M, N, K, a.stride(0), b.stride(0), c.stride(0),
                                       _lib.dtype_code(c), v, stream_ptr()), "lab")
                H.sp_lab_set(0, 0); H.sp_lab_set(1, 0)
            fns[key] = run
    if out_dtype == torch.bfloat16:
        c_old = torch.zeros(M, N, device=dev, dtype=out_dtype)
        outs["old_persist"] = c_old
        fns["old_persist"] = lambda: ops.gemm_nt(a, b, out=c_old)
    c_lib = torch.zeros(M, N, device=dev, dtype=out_dtype)
    if out_dtype == torch.bfloat16:
        fns["library"] = lambda: torch.mm(a, b.t(), out=c_lib)
    else:
        fns["library"] = lambda: torch.mm(a, b.t(), out_dtype=torch.float32, out=c_lib)
    t = timed(fns, rounds)
    for k in fns:
        err = ""
        if k in outs:
            e = float((outs[k].float() - ref).abs().max()) / float(ref.abs().max())
            err = f"  relerr {e:.2e}" + ("  WRONG" if e > 1e-2 and not (k[2:4].isdigit()) else "")
        print(f"{name:10s} {M}x{N}x{K} {k:14s} median {t[k][0]:7.1f} us  min {t[k][1]:7.1f} us  {fl / t[k][0] / 1e6:7.1f} TF/s{err}", flush=True)


def lab_tn(name, Mi, Nj, K, rounds, S, variants=(0,)):
    x = padded(K, Mi)
    gu = padded(K, Nj, 0.05, pad=0)
    ref = x.float().t() @ gu.float()
    fl = 2.0 * Mi * Nj * K
    fns, outs = {}, {}
    for v in variants:
        P = torch.zeros(S, Mi, Nj, device=dev)
        outs[f"rs{v}"] = P
        fns[f"rs{v}"] = (lambda v=v, P=P: check(H.sp_lab_gemm_tn(ptr(x), ptr(gu), ptr(P), Mi, Nj, K, x.stride(0), gu.stride(0), S, v,
                                                                 stream_ptr()), "lab_tn"))
    Pg = torch.zeros(S, Mi, Nj, device=dev)
    outs["general"] = Pg
    fns["general"] = lambda: ops.gemm(x.t(), gu.t(), out=Pg, out_dtype=torch.float32, split_k=S)
    Kc = K // 4 * 4
    xa, ga = x[:Kc].view(4, Kc // 4, Mi).transpose(1, 2), gu[:Kc].view(4, Kc // 4, Nj)
    fns["library_bmm4"] = lambda: torch.bmm(xa, ga, out_dtype=torch.float32)
    t = timed(fns, rounds)
    for k in fns:
        err = ""
        if k in outs:
            e = float((outs[k].sum(0) - ref).abs().max()) / float(ref.abs().max())
            err = f"  relerr {e:.2e}" + ("  WRONG" if e > 1e-3 else "")
        print(f"{name:10s} {Mi}x{Nj}x{K} {k:12s} median {t[k][0]:7.1f} us  min {t[k][1]:7.1f} us  {fl / t[k][0] / 1e6:7.1f} TF/s{err}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    M = 24032
    if what in ("nt", "all"):
        abl = tuple(int(v) for v in os.environ.get("LAB_VARIANTS", "0,1,2,3").split(","))
        lab_nt("fwd", M, 3072, 1024, rounds, variants=abl)
        lab_nt("dX", M, 1024, 3072, rounds, variants=abl)
        if os.environ.get("LAB_SHORT"):
            sys.exit(0)
        lab_nt("L0fwd", M, 4096, 1280, rounds)
        lab_nt("L0dX", M, 1280, 4096, rounds, variants=(0, 1, 3))
        lab_nt("logits", M, 1024, 1024, rounds, torch.float32)
        lab_nt("ragged", 1000, 768, 192, 2)
    if what in ("tn", "all") and hasattr(H, "sp_lab_gemm_tn"):
        lab_tn("dW", 1024, 3072, M, rounds, 5)
        lab_tn("L0dW", 1280, 4096, M, rounds, 3)
        lab_tn("cls_dW", 1024, 1024, M, rounds, 16)
