"""GEMM A/B lab: two builds of libspeech_hip.so in ONE process, the same operands, interleaved launches, every
launch between two HBM-bound spacer launches (as tools/gemm_lab.py); checks each against an f32 reference.
    python tools/gemm_ab.py <libA.so> <libB.so> [rounds] [nt|tn|all]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib          # noqa: E402

dev = torch.device("cuda")
spacer_buf = torch.empty(64 << 20, dtype=torch.float32, device=dev)


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    for name in ("sp_gemm", "sp_gemm_tn_splits", "sp_last_error"):
        res, args = _lib.SIGNATURES[name]
        getattr(L, name).restype, getattr(L, name).argtypes = res, args
    return L


def gemm(L, a, b, out, split_k=1):
    M, K = a.shape
    N = b.shape[0]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ldc = out.stride(-2)
    css = out.stride(0) if split_k > 1 else 0
    r = L.sp_gemm(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), a.stride(1), b.stride(0), b.stride(1),
                  ldc, _lib.dtype_code(a), _lib.dtype_code(out), 0, split_k, css, st)
    if r != 0:
        raise RuntimeError(L.sp_last_error().decode())


def padded(rows, cols, scale=1.0, pad=64, dtype=torch.bfloat16):
    buf = torch.empty(rows, cols + pad, device=dev, dtype=dtype)
    buf[:, :cols] = (torch.randn(rows, cols, device=dev) * scale).to(dtype)
    return buf[:, :cols]


def timed(fns, rounds, reps=4):
    res = {k: [] for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            ev = []
            for _ in range(reps):
                spacer_buf.mul_(1.0001)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            res[k] += [a.elapsed_time(b) * 1e3 for a, b in ev]
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def lab_nt(libs, name, M, N, K, rounds, out_dtype=torch.bfloat16, in_dtype=torch.bfloat16):
    a, b = padded(M, K, dtype=in_dtype), padded(N, K, 0.05, dtype=in_dtype)
    ref = a.float() @ b.float().t()
    outs = {k: torch.zeros(M, N, device=dev, dtype=out_dtype) for k in libs}
    fns = {k: (lambda k=k: gemm(libs[k], a, b, outs[k])) for k in libs}
    t = timed(fns, rounds)
    for k in libs:
        err = float((outs[k].float() - ref).abs().max()) / float(ref.abs().max())
        same = all(torch.equal(outs[k], outs[j]) for j in libs)
        med, mn = t[k]
        print(f"{name:8s} {M}x{N}x{K} {str(in_dtype)[6:]:>8s} {k:4s} median {med:7.1f} us  min {mn:7.1f} us  {2.0 * M * N * K / med / 1e6:7.1f} TF/s"
              f"  relerr {err:.2e}{'  WRONG' if err > 1e-2 else ''}  {'bit-identical' if same else 'differs from the other build'}", flush=True)


def lab_tn(libs, name, Mi, Nj, K, rounds):
    x, gu = padded(K, Mi), padded(K, Nj, 0.05, pad=0)
    ref = x.float().t() @ gu.float()
    S = libs["A"].sp_gemm_tn_splits(Mi, Nj, K, _lib.SP_BF16)
    outs = {k: torch.zeros(S, Mi, Nj, device=dev) for k in libs}
    fns = {k: (lambda k=k: gemm(libs[k], x.t(), gu.t(), outs[k], split_k=S)) for k in libs}
    t = timed(fns, rounds)
    for k in libs:
        err = float((outs[k].sum(0) - ref).abs().max()) / float(ref.abs().max())
        med, mn = t[k]
        print(f"{name:8s} {Mi}x{Nj}x{K}/S{S} {k:4s} median {med:7.1f} us  min {mn:7.1f} us  {2.0 * Mi * Nj * K / med / 1e6:7.1f} TF/s"
              f"  relerr {err:.2e}{'  WRONG' if err > 1e-2 else ''}", flush=True)


if __name__ == "__main__":
    libs = {"A": load(sys.argv[1]), "B": load(sys.argv[2])}
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    what = sys.argv[4] if len(sys.argv) > 4 else "nt"
    M = 24032
    if what in ("nt", "all"):
        lab_nt(libs, "tiny", 1000, 256 * 16, 128, 1)            # ragged M, two K steps per tile
        lab_nt(libs, "fwd", M, 3072, 1024, rounds)
        lab_nt(libs, "dX", M, 1024, 3072, rounds)
        lab_nt(libs, "L0fwd", M, 4096, 1280, rounds)
        lab_nt(libs, "L0dX", M, 1280, 4096, rounds)
        lab_nt(libs, "logits", M, 1024, 1024, rounds, torch.float32)
        lab_nt(libs, "fwd f16", M, 3072, 1024, max(rounds // 2, 1), torch.float16, torch.float16)
    if what in ("tn", "all"):
        lab_tn(libs, "dW", 1024, 3072, M, rounds)
        lab_tn(libs, "L0dW", 1280, 4096, M, rounds)
        lab_tn(libs, "cls_dW", 1024, 1024, M, rounds)
