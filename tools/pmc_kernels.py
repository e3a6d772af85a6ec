"""Runs each hand-written north-star kernel a few times at the bench's sizes (C2: 32 x 15 s,
V = 1000) so that `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, see
MI355X_MICROARCH.md "HBM") can attribute memory-side traffic to them.  Usage on the GPU box:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r -o f -- python3 tools/pmc_kernels.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, N, V, Lmax, d = 32, 240000, 1000, 60, 512
g = torch.Generator().manual_seed(7)
T = 1 + N // 160
Tp = (T - 1) // 2 + 1
wav = (0.1 * torch.randn(B * N, generator=g)).to(dev)
offs = (torch.arange(B + 1, dtype=torch.int64) * N).to(dev)
masks = torch.tensor([[10, 10, 100, 30]] * B, dtype=torch.int32, device=dev)
logits = (torch.randn(B, Tp, V, generator=g) * 3).to(dev)
labels = torch.randint(1, V, (B, Lmax), generator=g).to(dev)
lens = (torch.rand(B, generator=g) * 0.5 + 0.5).mul(Lmax).long().clamp(min=1).to(torch.int32).to(dev)
il = torch.full((B,), Tp, dtype=torch.int32, device=dev)
U = torch.randn(Tp, B, 3, 2, d, device=dev).bfloat16()
xr = torch.randn(Tp, B, 2 * d, device=dev).bfloat16()
wc = torch.randn(4 * d, device=dev) * 0.5
bias = torch.zeros(4 * d, device=dev)
h, c = ops.sru_fwd(U, xr, wc, bias, None, Tp, B, d, 3)
gh = torch.randn_like(h)
for _ in range(3):
    ops.melspec(wav, offs, masks=masks)
    ops.ctc_fwd_bwd(logits, labels, il, lens, max_label_len=Lmax)
    ops.greedy_decode_ids(logits)
    ops.sru_fwd(U, xr, wc, bias, None, Tp, B, d, 3)
    ops.sru_bwd(U, xr, wc, bias, None, c, gh, Tp, B, d, 3)
torch.cuda.synchronize()
