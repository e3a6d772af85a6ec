"""GPU parity tests: the HIP kernels (called through the C ABI) against the CPU oracle, on the
committed golden vectors, on seeded inputs, and at full size through size-independent properties.

Tolerances (written where used):
  decode  bit-exact ids.
  mel     |hip - oracle| <= 1e-4 * max|oracle| per utterance (north_star: 1e-4 relative in fp32; a
          per-element relative test is meaningless in near-zero-energy bins, so the floor is the
          utterance's peak).
  ctc     loss/nll 1e-4 relative.  Gradient: |hip - ref64| <= 1e-4 * max|ref64| against the
          reference's CTC evaluated in float64 (torch CPU, dtype=float64).  Against the reference's
          own fp32 run the bound is 1e-4 * max + 2 * |ref32 - ref64|: alpha/beta reach thousands of
          nats where one fp32 ulp is ~2e-4, so torch's fp32 CTC itself is off by a few 1e-4 relative
          at T >= 100 (measured in the test), which no implementation can track element by element.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import decode as odec
from oracle import mel as omel
from oracle import sru as osru

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _ids_list(ids, lengths):
    ids, lengths = ids.cpu().numpy(), lengths.cpu().numpy()
    return [ids[b, : lengths[b]].tolist() for b in range(ids.shape[0])]


# =============================================================================== decode
def test_decode_golden(cuda, golden_dir):
    from speech_amd import ops
    arrs = _load(golden_dir, "decode_golden.npz")
    meta = json.load(open(os.path.join(golden_dir, "decode_golden.json")))
    for name, m in meta.items():
        scores = torch.from_numpy(arrs[name + "__scores"]).to(cuda)
        ids, lengths = ops.greedy_decode_ids(scores)
        assert _ids_list(ids, lengths) == m["ids"], name
        full = ids.cpu().numpy()
        for b, L in enumerate(lengths.cpu().tolist()):
            assert (full[b, L:] == -1).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(7, 751, 29), (3, 257, 29), (2, 100, 1000), (5, 1, 64), (2, 300, 61)])
def test_decode_random(cuda, dtype, shape):
    from speech_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    scores = (torch.randn(*shape, generator=g) * 2).to(dtype)
    # make decodes non-trivial: peaky rows with frequent blanks and repeats
    path = torch.randint(0, min(shape[2], 6), shape[:2], generator=g)
    scores.scatter_(2, path.unsqueeze(-1), 6.0)
    ids, lengths = ops.greedy_decode_ids(scores.to(cuda))
    ref = odec.greedy_decode_ids(scores.float())
    assert _ids_list(ids, lengths) == ref


def test_decode_frame_lengths_and_empty(cuda):
    from speech_amd import ops
    g = torch.Generator().manual_seed(5)
    scores = torch.randn(4, 90, 29, generator=g)
    fl = torch.tensor([90, 40, 1, 0], dtype=torch.int32)
    ids, lengths = ops.greedy_decode_ids(scores.to(cuda), frame_lengths=fl)
    got = _ids_list(ids, lengths)
    for b in range(4):
        assert got[b] == odec.greedy_decode_ids(scores[b: b + 1, : int(fl[b])])[0]
    ids, lengths = ops.greedy_decode_ids(torch.empty(0, 10, 29, device=cuda))
    assert ids.shape == (0, 10)


def test_decode_full_size_properties(cuda):
    """C5 size (1000 x 751 x 29): idempotence-style properties instead of an oracle pass."""
    from speech_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    scores = torch.randn(1000, 751, 29, generator=g, device=cuda) * 3
    ids, lengths = ops.greedy_decode_ids(scores)
    am = scores.argmax(2)
    keep = (am != 0) & torch.cat([torch.ones_like(am[:, :1], dtype=torch.bool), am[:, 1:] != am[:, :-1]], 1)
    assert torch.equal(lengths.long(), keep.sum(1))
    assert torch.equal(ids[keep.sum(1, keepdim=True) > torch.arange(751, device=cuda)].long(), am[keep])
    # no blank, no adjacent duplicate produced from adjacent frames, tail is -1
    sub = odec.greedy_decode_ids(scores[:8].cpu())
    assert _ids_list(ids[:8], lengths[:8]) == sub


# =============================================================================== mel
def _mel_case(cuda, wavs, masks=None, t_max=None):
    from speech_amd import ops
    flat = torch.cat([w.reshape(-1) for w in wavs]).float()
    offs = torch.tensor(np.cumsum([0] + [w.numel() for w in wavs]), dtype=torch.int64)
    m = None if masks is None else torch.tensor(masks, dtype=torch.int32)
    out = ops.melspec(flat.to(cuda), offs, t_max=t_max, masks=m)
    ref = omel.collate([w.reshape(1, -1) for w in wavs], masks)
    return out.cpu(), ref


def _assert_mel_close(out, ref):
    assert out.shape[:3] == ref.shape[:3]
    Tr = ref.shape[-1]
    for b in range(ref.shape[0]):
        tol = 1e-4 * float(ref[b].abs().max()) + 1e-12   # 1e-4 of the utterance's peak
        assert float((out[b, ..., :Tr] - ref[b]).abs().max()) <= tol
    if out.shape[-1] > Tr:
        assert float(out[..., Tr:].abs().max()) == 0.0


def test_mel_golden(cuda, golden_dir):
    g = _load(golden_dir, "mel_golden.npz")
    for name in ["noise", "sine_1k", "impulse", "dc", "short_201", "ragged_1777"]:
        wav = torch.from_numpy(g[name + "__wav"])
        out, _ = _mel_case(cuda, [wav])
        ref = torch.from_numpy(g[name + "__mel"])
        assert out.shape == (1, 1, 80, ref.shape[-1])
        err = float((out[0, 0] - ref).abs().max())
        assert err <= 1e-4 * float(ref.abs().max()), (name, err)
        f64 = g[name + "__mel_f64"]
        assert np.abs(out[0, 0].numpy() - f64).max() <= 2e-6 * np.abs(f64).max(), name


def test_mel_ragged_batch_masks_padding(cuda):
    g = torch.Generator().manual_seed(7)
    lens = [16000, 15999, 8123, 4000, 1601, 201]
    wavs = []
    for n in lens:
        t = torch.arange(n) / 16000.0
        f = float(torch.randint(100, 4000, (1,), generator=g))
        wavs.append((0.1 * torch.randn(n, generator=g) + 0.3 * torch.sin(2 * np.pi * f * t)).clamp(-1, 1))
    out, ref = _mel_case(cuda, wavs)
    _assert_mel_close(out, ref)
    masks = [(10, 14, 5, 34), (0, 0, 0, 0), (70, 10, 40, 20), (79, 1, 0, 1), (3, 5, 8, 2), (0, 15, 1, 1)]
    out, ref = _mel_case(cuda, wavs, masks, t_max=120)
    _assert_mel_close(out, ref)
    assert out.shape[-1] == 120


def test_mel_full_size_properties(cuda):
    """C2 size (32 x 15 s): linearity, zero padding, and agreement with the oracle on a slice."""
    from speech_amd import ops
    B, N = 32, 240000
    g = torch.Generator(device="cuda").manual_seed(7)
    wav = (0.1 * torch.randn(B, N, generator=g, device=cuda)).clamp(-1, 1)
    offs = torch.arange(B + 1, dtype=torch.int64) * N
    a = ops.melspec(wav.reshape(-1), offs)
    assert a.shape == (B, 1, 80, 1501)
    b2 = ops.melspec((2 * wav).reshape(-1), offs)
    assert float((b2 - 2 * a).abs().max()) <= 1e-5 * float(a.abs().max())      # homogeneous of degree 1
    a2 = ops.melspec(wav.reshape(-1), offs)
    assert torch.equal(a, a2)                                                     # deterministic
    ref = omel.mel_spectrogram(wav[3].cpu())
    assert float((a[3, 0].cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


# =============================================================================== ctc
def _ctc_hip(cuda, logits_tbv, labels, il, ll, time_major=True, max_label_len=None):
    from speech_amd import ops
    z = torch.as_tensor(logits_tbv).float().to(cuda)
    if not time_major:
        z = z.transpose(0, 1).contiguous()
    loss, nll, grad = ops.ctc_fwd_bwd(z, torch.as_tensor(labels).to(cuda), torch.as_tensor(il).to(cuda),
                                      torch.as_tensor(ll).to(cuda), time_major=time_major,
                                      max_label_len=max_label_len)
    grad = grad.cpu()
    if not time_major:
        grad = grad.transpose(0, 1)
    return float(loss.cpu()), nll.cpu().numpy(), grad.numpy()


CTC_GOLDEN = ["char_small", "char_ragged_T", "char_mid", "char_s200", "char_s400", "sp_vocab", "t1_l1", "l0"]


@pytest.mark.parametrize("name", CTC_GOLDEN)
@pytest.mark.parametrize("time_major", [True, False])
def test_ctc_golden(cuda, golden_dir, name, time_major):
    g = _load(golden_dir, "ctc_golden.npz")
    loss, nll, grad = _ctc_hip(cuda, g[name + "__logits"], g[name + "__labels"],
                               g[name + "__input_lengths"], g[name + "__label_lengths"], time_major)
    assert loss == pytest.approx(float(g[name + "__loss"]), rel=1e-4)          # 1e-4 relative
    assert np.allclose(nll, g[name + "__nll"], rtol=1e-4, atol=1e-5)
    gref = g[name + "__grad"]
    _, _, g64 = octc.ctc_torch(g[name + "__logits"], g[name + "__labels"], g[name + "__input_lengths"],
                               g[name + "__label_lengths"], dtype=torch.float64)
    g64 = g64.numpy()
    gmax = np.abs(g64).max()
    assert np.abs(grad - g64).max() <= 1e-4 * gmax + 1e-12
    ref32_err = np.abs(gref - g64).max()
    assert np.abs(grad - gref).max() <= 1e-4 * gmax + 2 * ref32_err + 1e-12
    # frames past input_length carry exactly zero gradient
    il = g[name + "__input_lengths"]
    for b, tb in enumerate(il):
        assert np.all(grad[tb:, b] == 0)


def test_ctc_infeasible_is_inf(cuda, golden_dir):
    g = _load(golden_dir, "ctc_golden.npz")
    name = "infeasible_short"
    loss, nll, _ = _ctc_hip(cuda, g[name + "__logits"], g[name + "__labels"],
                            g[name + "__input_lengths"], g[name + "__label_lengths"])
    assert np.isinf(nll).all() and np.isinf(loss)


@pytest.mark.parametrize("cfg", [(751, 8, 29, 225), (300, 4, 1000, 60), (1501, 2, 29, 450), (700, 3, 29, 600)])
def test_ctc_seeded_vs_oracles(cuda, cfg):
    """Sizes the CPU finishes in seconds: against torch CPU fp32 (the reference) and float64."""
    T, B, V, L = cfg
    g = torch.Generator().manual_seed(T + V)
    logits = torch.randn(T, B, V, generator=g) * 3
    labels = torch.randint(1, V, (B, L), generator=g)
    labels[:, 1::7] = labels[:, 0::7][:, : labels[:, 1::7].shape[1]]           # some adjacent repeats
    ll = torch.randint(L // 2, L + 1, (B,), generator=g).int()
    ll[0] = L
    il = torch.full((B,), T, dtype=torch.int32)
    loss, nll, grad = _ctc_hip(cuda, logits, labels, il, ll)
    rl, rnll, rgrad = octc.ctc_torch(logits, labels, il, ll)
    assert loss == pytest.approx(float(rl), rel=1e-4)
    assert np.allclose(nll, rnll.numpy(), rtol=1e-4)
    feasible = np.isfinite(rnll.numpy())
    assert feasible.any()
    _, _, g64 = octc.ctc_torch(logits, labels, il, ll, dtype=torch.float64)
    g64 = g64.numpy()[:, feasible]
    gmax = np.abs(g64).max()
    assert np.abs(grad[:, feasible] - g64).max() <= 1e-4 * gmax + 1e-12
    ref32_err = np.abs(rgrad.numpy()[:, feasible] - g64).max()
    assert np.abs(grad[:, feasible] - rgrad.numpy()[:, feasible]).max() <= 1e-4 * gmax + 2 * ref32_err + 1e-12


def test_ctc_autograd_wrapper(cuda):
    from speech_amd import ops
    g = torch.Generator().manual_seed(3)
    B, T, V, L = 4, 50, 29, 10
    logits = torch.randn(B, T, V, generator=g)
    labels = torch.randint(1, V, (B, L), generator=g)
    ll = torch.tensor([10, 8, 3, 1], dtype=torch.int32)
    il = torch.full((B,), T, dtype=torch.int32)
    z = logits.to(cuda).requires_grad_(True)
    loss = ops.ctc_loss(z, labels.to(cuda), il.to(cuda), ll.to(cuda))
    (3.0 * loss).backward()
    zr = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(zr, 2).transpose(0, 1)
    ref = torch.nn.CTCLoss(blank=0)(lp, labels, il, ll)
    (3.0 * ref).backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)
    assert float((z.grad.cpu() - zr.grad).abs().max()) <= 1e-4 * float(zr.grad.abs().max())


def test_ctc_full_size_properties(cuda):
    """C4 size (T'=1501, B=64): posterior mass per frame is 1 => every gradient row sums to 0;
    nll matches torch CPU on a 2-utterance slice."""
    from speech_amd import ops
    T, B, V, L = 1501, 64, 29, 450
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(T, B, V, generator=g) * 3
    labels = torch.randint(1, V, (B, L), generator=g)
    ll = torch.randint(L // 2, L + 1, (B,), generator=g).int()
    il = torch.full((B,), T, dtype=torch.int32)
    loss, nll, grad = ops.ctc_fwd_bwd(logits.to(cuda), labels.to(cuda), il.to(cuda), ll.to(cuda), time_major=True)
    assert torch.isfinite(nll).all()
    rows = grad.sum(2).abs().max()
    assert float(rows) <= 1e-5
    _, rn, _ = octc.ctc_torch(logits[:, :2], labels[:2], il[:2], ll[:2])
    assert np.allclose(nll[:2].cpu().numpy(), rn.numpy(), rtol=1e-5)


# =============================================================================== sru recurrence
def _sru_inputs(T, B, d, k, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    U = torch.randn(T, B, k, 2, d, generator=g)
    x = torch.randn(T, B, 2 * d, generator=g) if k == 3 else None
    wc = torch.randn(4 * d, generator=g) * 0.5
    bias = torch.randn(4 * d, generator=g) * 0.1
    mask = (torch.rand(B, 2 * d, generator=g) > 0.1).float() / 0.9
    return U.to(dtype), (x.to(dtype) if x is not None else None), wc, bias, mask


def _sru_ref(U, x, wc, bias, mask, d):
    """Recurrence only (the projection is outside the kernel): oracle.sru's equations on a
    pre-computed U in the kernel layout [T,B,k,2,d]."""
    T, B, k = U.shape[:3]
    U = U.double().requires_grad_(True)
    x = x.double().requires_grad_(True) if x is not None else None
    wc = wc.double().requires_grad_(True)
    bias = bias.double().requires_grad_(True)
    D2 = 2 * d
    vf, vr = wc[:D2].view(2, d), wc[D2:].view(2, d)
    bf, br = bias[:D2].view(2, d), bias[D2:].view(2, d)
    m = mask.double().view(B, 2, d)
    hs = [[None] * T, [None] * T]
    for direction in range(2):
        c = torch.zeros(B, d, dtype=torch.float64)
        order = range(T) if direction == 0 else range(T - 1, -1, -1)
        for t in order:
            u0, u1, u2 = U[t, :, 0, direction], U[t, :, 1, direction], U[t, :, 2, direction]
            f = torch.sigmoid(u1 + vf[direction] * c + bf[direction])
            r = torch.sigmoid(u2 + vr[direction] * c + br[direction])
            c = u0 + (c - u0) * f
            xp = x[t].view(B, 2, d)[:, direction] if k == 3 else U[t, :, 3, direction]
            hs[direction][t] = (c * m[:, direction] - xp) * r + xp
    h = torch.stack([torch.cat([hs[0][t], hs[1][t]], -1) for t in range(T)])
    return h, (U, x, wc, bias)


@pytest.mark.parametrize("k", [3, 4])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_sru_recurrence_fwd_bwd(cuda, k, dtype, tol):
    from speech_amd import ops
    T, B, d = 37, 3, 64
    U, x, wc, bias, mask = _sru_inputs(T, B, d, k, 11 + k, dtype)
    Ud = U.to(cuda).requires_grad_(True)
    xd = x.to(cuda).requires_grad_(True) if x is not None else None
    wcd = wc.to(cuda).requires_grad_(True)
    bd = bias.to(cuda).requires_grad_(True)
    h = ops.sru_recurrence(Ud, xd, wcd, bd, mask.to(cuda), d)
    g = torch.Generator().manual_seed(99)
    gh = torch.randn(T, B, 2 * d, generator=g).to(dtype)
    h.backward(gh.to(cuda))
    href, (Ur, xr, wcr, br) = _sru_ref(U.float(), x.float() if x is not None else None, wc, bias, mask, d)
    href.backward(gh.double())
    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach().double()
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), what
    close(h, href, "h")
    close(Ud.grad, Ur.grad, "gU")
    if k == 3:
        close(xd.grad, xr.grad, "gx")
    close(wcd.grad, wcr.grad, "gwc")
    close(bd.grad, br.grad, "gbias")


def test_sru_matches_layer_oracle(cuda):
    """Whole-layer check against oracle.sru.sru_layer (reference weight layout, k innermost)."""
    from speech_amd import ops
    T, B, d, n_in = 21, 2, 64, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, n_in, generator=g)
    W = torch.randn(n_in, 2 * d * 3, generator=g) * 0.1
    wc = torch.randn(4 * d, generator=g) * 0.5
    bias = torch.randn(4 * d, generator=g) * 0.1
    lnw, lnb = torch.rand(n_in, generator=g) + 0.5, torch.randn(n_in, generator=g) * 0.1
    href, _ = osru.sru_layer(x, W, wc, bias, lnw, lnb, d)
    xn = torch.nn.functional.layer_norm(x, (n_in,), lnw, lnb)
    U = (xn.reshape(T * B, n_in) @ W).view(T, B, 2, d, 3).permute(0, 1, 4, 2, 3).contiguous()  # -> [T,B,k,2,d]
    h = ops.sru_recurrence(U.to(cuda), x.to(cuda), wc.to(cuda), bias.to(cuda), None, d)
    assert float((h.cpu() - href).abs().max()) <= 2e-5 * max(1.0, float(href.abs().max()))
