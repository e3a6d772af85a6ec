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


def test_mel_frame_count_edges(cuda):
    """Frame counts around the kernel's units (8 frames per wave, 16 per wave pair, 32 per workgroup; 1520 staged
    samples per wave), padded outputs whose extra frames fall in a live wave's partner / in a pair of their own,
    utterances whose sample offsets are odd (the 16-byte staging loads are only 4-byte aligned), f32 and int16."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(11)
    lens = [201, 239, 320, 321, 399, 401, 1119, 1120, 1121, 1279, 1280, 1281, 2399, 2400, 2559, 2560, 2561, 4959, 5121, 7013]
    wavs = [(0.2 * torch.randn(n, generator=g)).clamp(-1, 1) for n in lens]
    for chunk in (wavs[:7], wavs[7:14], wavs[14:]):
        tb = max(1 + w.numel() // 160 for w in chunk)
        for extra in (0, 1, 7, 8, 9, 17, 40):
            out, ref = _mel_case(cuda, chunk, t_max=tb + extra)
            assert out.shape[-1] == tb + extra
            _assert_mel_close(out, ref)
    # int16 PCM of the same utterances (odd lengths make odd sample offsets): bit-identical to the f32 path on
    # the dequantised samples
    pcm = [(w * 32767).round().to(torch.int16) for w in wavs]
    offs = torch.tensor([0] + list(np.cumsum([w.numel() for w in wavs])), dtype=torch.int64)
    a = ops.melspec(torch.cat(pcm).to(cuda), offs.to(cuda))
    b = ops.melspec((torch.cat(pcm).float() / 32768.0).to(cuda), offs.to(cuda))
    assert torch.equal(a, b)


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


def test_mel_full_length_against_f64_dft(cuda):
    """The sharp bound at production length: two 15 s utterances (noise + tone, a quiet one) against the float64
    DFT restatement -- 2e-6 of the utterance peak on every bin, and element-relative 2e-4 on every bin that holds
    at least 1e-3 of the peak (f32 FFT round-off is relative to the frame's energy, not to the bin)."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(21)
    N = 240000
    t = torch.arange(N, dtype=torch.float32) / 16000.0
    wavs = [(0.1 * torch.randn(N, generator=g) + 0.3 * torch.sin(2 * torch.pi * 440.0 * t)).clamp(-1, 1),
            1e-3 * torch.randn(N - 777, generator=g)]
    flat = torch.cat(wavs)
    offs = torch.tensor([0, N, 2 * N - 777], dtype=torch.int64)
    out = ops.melspec(flat.to(cuda), offs.to(cuda)).cpu().numpy()
    for b, w in enumerate(wavs):
        ref = np.asarray(omel.mel_spectrogram_f64(w.numpy()))
        got = out[b, 0, :, : ref.shape[-1]].astype(np.float64)
        peak = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 2e-6 * peak, b
        big = np.abs(ref) >= 1e-3 * peak
        assert (np.abs(got - ref)[big] / np.abs(ref)[big]).max() <= 2e-4, b


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


@pytest.mark.parametrize("cfg", [(203, 5, 1000, 60), (203, 5, 1000, 120), (150, 4, 600, 250), (203, 5, 29, 60),
                                 (97, 3, 31, 7), (2, 2, 1000, 1), (1, 3, 50, 1), (40, 70, 1000, 6), (300, 66, 29, 130), (60, 2, 4000, 50),
                                 (120, 2, 2048, 100)])
def test_ctc_split_gradient_repeats_and_ragged(cuda, cfg):
    """Single-wave lattices form the gradient in two halves (softmax over the vocabulary beside the lattice, then the
    label classes' occupancies): labels that repeat a class many times (adjacent and apart), ragged input lengths
    that cut the frame groups of the sparse half anywhere, every states-per-lane variant, class- and position-indexed
    rows, and batches above 64 (single-wave lattice two workgroups to a CU + the fused gradient kernel, also on the
    eight-states-per-lane kernel's permuted rows); against torch
    CPU float64, and zero rows beyond each length."""
    T, B, V, L = cfg
    g = torch.Generator().manual_seed(T * 7 + L)
    logits = torch.randn(T, B, V, generator=g) * 3
    labels = torch.randint(1, V, (B, L), generator=g)
    labels[0, : max(L // 2, 1)] = 5                       # one class over and over (mandatory blanks between)
    if L >= 6:
        labels[1, ::3] = 7                                 # the same class apart
        labels[1, 1::3] = 9
    ll = torch.randint(max(L // 2, 1), L + 1, (B,), generator=g).int()
    ll[0] = max(L // 2, 1)
    il = torch.full((B,), T, dtype=torch.int32)
    if T > 8:
        il[1] = T - 1
        il[min(2, B - 1)] = T - 6
        il[-1] = T - 3
    for b in range(B):                                     # keep every utterance feasible
        need = int(ll[b]) + int((labels[b, 1:int(ll[b])] == labels[b, :int(ll[b]) - 1]).sum())
        while need > int(il[b]):
            ll[b] -= 1
            need = int(ll[b]) + int((labels[b, 1:int(ll[b])] == labels[b, :int(ll[b]) - 1]).sum())
    ll.clamp_(min=1)
    loss, nll, grad = _ctc_hip(cuda, logits, labels, il, ll, max_label_len=L)
    rl, rnll, g64 = octc.ctc_torch(logits, labels, il, ll, dtype=torch.float64)
    feasible = np.isfinite(rnll.numpy())
    assert feasible.any()
    assert np.allclose(nll[feasible], rnll.numpy()[feasible], rtol=1e-4, atol=1e-5)
    g64 = g64.numpy()
    gmax = np.abs(g64[:, feasible]).max()
    assert np.abs(grad[:, feasible] - g64[:, feasible]).max() <= 1e-4 * gmax + 1e-12
    for b in range(B):
        assert np.all(grad[int(il[b]):, b] == 0)


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
    pre-computed U in the kernel layout [T,B,k,2,d].  The time steps are unbound once (one stack in the
    backward instead of a full-size zero fill per step), so T = 1501 stays a matter of seconds."""
    T, B, k = U.shape[:3]
    U = U.double().requires_grad_(True)
    x = x.double().requires_grad_(True) if x is not None else None
    wc = wc.double().requires_grad_(True)
    bias = bias.double().requires_grad_(True)
    D2 = 2 * d
    vf, vr = wc[:D2].view(2, d), wc[D2:].view(2, d)
    bf, br = bias[:D2].view(2, d), bias[D2:].view(2, d)
    m = mask.double().view(B, 2, d)
    Ut = U.unbind(0)
    xt = x.view(T, B, 2, d).unbind(0) if x is not None else None
    hs = [[None] * T, [None] * T]
    for direction in range(2):
        c = torch.zeros(B, d, dtype=torch.float64)
        order = range(T) if direction == 0 else range(T - 1, -1, -1)
        for t in order:
            u = Ut[t][:, :, direction].unbind(1)
            f = torch.sigmoid(u[1] + vf[direction] * c + bf[direction])
            r = torch.sigmoid(u[2] + vr[direction] * c + br[direction])
            c = u[0] + (c - u[0]) * f
            xp = xt[t][:, direction] if k == 3 else u[3]
            hs[direction][t] = (c * m[:, direction] - xp) * r + xp
    h = torch.cat([torch.stack(hs[0]), torch.stack(hs[1])], -1)
    return h, (U, x, wc, bias)


@pytest.mark.parametrize("k", [3, 4])
@pytest.mark.parametrize("dtype,tol,T,d", [(torch.float32, 2e-5, 37, 64), (torch.bfloat16, 2e-2, 37, 64),
                                           (torch.float16, 3e-3, 37, 64), (torch.float16, 3e-3, 131, 128),   # fp16: the reference's eval type
                                           (torch.bfloat16, 2e-2, 131, 128),   # loader-ring kernels
                                           (torch.bfloat16, 2e-2, 7, 256), (torch.bfloat16, 2e-2, 64, 128),
                                           (torch.bfloat16, 2e-2, 1, 64), (torch.bfloat16, 2e-2, 2, 64), (torch.bfloat16, 2e-2, 6, 64)])
def test_sru_recurrence_fwd_bwd(cuda, k, dtype, tol, T, d):
    from speech_amd import ops
    B = 3
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


# =============================================================================== conv stack
def _conv_ref_and_hip(cuda, B, Fin, Tin, seed, train=True):
    """torch fp32 statement of net.py:56-59 + :6-42 (the reference's own ops) vs the HIP stack."""
    from speech_amd import convstack, net
    torch.manual_seed(seed)
    cnn = torch.nn.Conv2d(1, 32, 3, stride=2, padding=1).to(cuda)
    blocks = [net.ResidualBlock(32, 32, 5).to(cuda) for _ in range(2)]
    for blk in blocks:
        for bn in (blk.bn1, blk.bn2):
            bn.weight.data.uniform_(0.5, 1.5)
            bn.bias.data.uniform_(-0.3, 0.3)
    spec = torch.rand(B, 1, Fin, Tin, device=cuda) * 2
    import copy
    cnn_r, blocks_r = copy.deepcopy(cnn), copy.deepcopy(blocks)
    for m in [cnn_r] + blocks_r + [cnn] + blocks:
        m.train(train)
    y = cnn_r(spec)
    for blk in blocks_r:
        y = blk(y)
    ref = y.permute(3, 0, 2, 1).reshape(y.shape[3], B, -1)            # [T', B, f*32 + c]
    out = convstack.conv_stack(spec, cnn, blocks, train)
    return spec, (cnn, blocks, out), (cnn_r, blocks_r, ref)


@pytest.mark.parametrize("shape", [(2, 80, 301), (3, 16, 70), (1, 9, 131)])
def test_conv_stack_forward(cuda, shape):
    B, Fin, Tin = shape
    _, (_, blocks, out), (_, blocks_r, ref) = _conv_ref_and_hip(cuda, B, Fin, Tin, 3)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    err = float((out.float() - ref).abs().max())
    assert err <= 3e-2 * float(ref.abs().max()), err            # bf16 activations through 5 convs
    # running statistics follow torch's update rule
    for a, b in zip(blocks, blocks_r):
        assert torch.allclose(a.bn2.running_var, b.bn2.running_var, rtol=3e-2, atol=1e-3)
        assert torch.allclose(a.bn1.running_mean, b.bn1.running_mean, rtol=3e-2, atol=2e-3)


def test_conv_stack_eval_mode(cuda):
    _, (_, _, out), (_, _, ref) = _conv_ref_and_hip(cuda, 2, 40, 90, 5, train=False)
    assert float((out.float() - ref).abs().max()) <= 3e-2 * float(ref.abs().max())


def _bf(x):
    return x.to(torch.bfloat16)


def _nhwc(x):   # [B,C,F,T] f32 -> [B,F,T,C] bf16
    return _bf(x.permute(0, 2, 3, 1).contiguous())


def _nchw(x):   # [B,F,T,C] bf16 -> [B,C,F,T] f32
    return x.float().permute(0, 3, 1, 2).contiguous()


def _relmax(a, b):
    return float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-9)


# (70, 8, 832): 1820 weight-gradient work items for at most 512 x 3 persistent groups -- two items per group
@pytest.mark.parametrize("shape", [(2, 7, 100), (1, 40, 751), (3, 4, 64), (70, 8, 832)])
def test_conv_kernels_individually(cuda, shape):
    """Every backward-path kernel against torch on IDENTICAL bf16-representable operands, so the
    comparison is not blurred by ReLU-mask flips: tolerance = bf16 output rounding (1e-2 of the
    peak) for bf16 outputs, 2e-3 for f32 accumulations."""
    from speech_amd import convstack as cs
    from speech_amd._lib import check, lib, ptr, stream_ptr
    B, Fq, T = shape
    g_ = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    rnd = lambda *sz: torch.randn(*sz, generator=g_, device=cuda)   # noqa: E731
    w = rnd(32, 32, 3, 5) * 0.1
    wq = _bf(w).float()
    gten, yten, aten = _bf(rnd(B, Fq, T, 32)), _bf(rnd(B, Fq, T, 32)), _bf(rnd(B, Fq, T, 32))
    coef3 = torch.cat([rnd(32) * 0.5 + 1, rnd(32) * 0.2, rnd(32) * 0.1]).contiguous()
    coef2 = torch.cat([rnd(32) * 0.5 + 1, rnd(32) * 0.3]).contiguous()
    gy = _bf(coef3[:32] * gten.float() + coef3[32:64] * yten.float() + coef3[64:]).float()   # staged value
    gy_nchw = gy.permute(0, 3, 1, 2).contiguous()
    # --- backward-data, mask epilogue
    ref_gin = torch.nn.grad.conv2d_input((B, 32, Fq, T), wq, gy_nchw, padding=(1, 2))
    keep = (coef2[:32] * aten.float() + coef2[32:]) > 0
    out = cs._conv(gten, yten, coef3, cs.IN_BNBWD, cs._pack(w, 1), None, aten, coef2, cs.EPI_MASK, B, Fq, T)
    ref = ref_gin.permute(0, 2, 3, 1) * keep
    assert _relmax(out, ref) <= 1e-2
    # --- backward-data, residual-add epilogue
    out = cs._conv(gten, yten, coef3, cs.IN_BNBWD, cs._pack(w, 1), None, aten, None, cs.EPI_ADD, B, Fq, T)
    assert _relmax(out, ref_gin.permute(0, 2, 3, 1) + aten.float()) <= 1e-2
    # --- weight gradient, activation = relu(BN(a_src)) and identity
    from speech_amd._lib import CONV_WGRAD_WS_WORDS
    ws = torch.full((CONV_WGRAD_WS_WORDS,), float("nan"), device=cuda)     # nothing relies on its contents
    for mode in (cs.IN_BNRELU, cs.IN_ID):
        act = _bf(torch.relu(coef2[:32] * aten.float() + coef2[32:])).float() if mode == cs.IN_BNRELU else aten.float()
        ref_gw = torch.nn.grad.conv2d_weight(act.permute(0, 3, 1, 2).contiguous(), (32, 32, 3, 5), gy_nchw, padding=(1, 2))
        gw = torch.zeros(32, 32, 3, 5, device=cuda)
        check(lib().sp_conv3x5_wgrad(ptr(aten), ptr(coef2), mode, ptr(gten), ptr(yten), ptr(coef3), B, Fq, T,
                                     ptr(ws), ptr(gw), stream_ptr()), "wgrad")
        assert _relmax(gw, ref_gw) <= 2e-3, mode
        gw2 = torch.zeros(32, 32, 3, 5, device=cuda)        # no float atomics: the same bits every time
        check(lib().sp_conv3x5_wgrad(ptr(aten), ptr(coef2), mode, ptr(gten), ptr(yten), ptr(coef3), B, Fq, T,
                                     ptr(ws), ptr(gw2), stream_ptr()), "wgrad")
        assert torch.equal(gw, gw2)
    # --- BN backward reduce / block backward entry
    from speech_amd._lib import BN_SUMS_WORDS
    sums_buf = torch.empty(BN_SUMS_WORDS, device=cuda)
    sums = sums_buf          # per-workgroup partials + their count; sp_bn_finalize_* adds them up
    red = _bn_total
    check(lib().sp_bn_bwd_reduce(ptr(gten), ptr(yten), B * Fq * T, ptr(sums), stream_ptr()), "reduce")
    ref_s = torch.cat([gten.float().sum((0, 1, 2)), (gten.float() * yten.float()).sum((0, 1, 2))])
    assert float((red(sums) - ref_s).abs().max()) <= 2e-3 * float(ref_s.abs().max()) + 1e-2
    zten = _bf(rnd(B, Fq, T, 32))
    gpre = torch.empty_like(gten)
    check(lib().sp_block_bwd_in(ptr(gten), ptr(zten), Fq * T * 32, T * 32, 32, ptr(yten), B, Fq, T, ptr(gpre),
                                ptr(sums), stream_ptr()), "bwd_in")
    ref_g = gten.float() * (zten.float() > 0)
    assert torch.equal(gpre.float(), ref_g)
    ref_s = torch.cat([ref_g.sum((0, 1, 2)), (ref_g * yten.float()).sum((0, 1, 2))])
    assert float((red(sums) - ref_s).abs().max()) <= 2e-3 * float(ref_s.abs().max()) + 1e-2
    # --- BN finalize backward: coefficients reproduce torch's batch_norm backward
    n = B * Fq * T
    yv = yten.float().reshape(-1, 32).clone().requires_grad_(True)
    gamma = rnd(32) * 0.3 + 1
    mean, var = yv.mean(0), yv.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    (gamma * (yv - mean) * invstd).backward(gten.float().reshape(-1, 32))
    save = torch.cat([mean.detach(), invstd.detach()]).contiguous()
    c3 = torch.empty(96, device=cuda)
    gg, gb = torch.zeros(32, device=cuda), torch.zeros(32, device=cuda)
    check(lib().sp_bn_bwd_reduce(ptr(gten), ptr(yten), n, ptr(sums), stream_ptr()), "reduce")
    check(lib().sp_bn_finalize_bwd(ptr(sums), ptr(save), ptr(gamma), n, ptr(c3), ptr(gg), ptr(gb), stream_ptr()), "fin")
    dy = c3[:32] * gten.float().reshape(-1, 32) + c3[32:64] * yten.float().reshape(-1, 32) + c3[64:]
    assert _relmax(dy, yv.grad) <= 2e-3
    # --- stem backward
    Fin, Tin = 2 * Fq - 1, 2 * T
    spec = rnd(B, 1, Fin, Tin)
    assert (Fin - 1) // 2 + 1 == Fq and (Tin - 1) // 2 + 1 == T
    gsw, gsb = torch.zeros(288, device=cuda), torch.zeros(32, device=cuda)
    check(lib().sp_conv_stem_bwd(ptr(spec), ptr(gten), B, Fin, Tin, ptr(gsw), ptr(gsb), stream_ptr()), "stem")
    gn = gten.float().permute(0, 3, 1, 2).contiguous()
    ref_w = torch.nn.grad.conv2d_weight(spec, (32, 1, 3, 3), gn, stride=2, padding=1)
    assert _relmax(gsw.view(32, 1, 3, 3), ref_w) <= 2e-3
    assert _relmax(gsb, gn.sum((0, 2, 3))) <= 2e-3


@pytest.mark.parametrize("shape", [(2, 80, 301), (3, 16, 70), (1, 9, 131), (2, 7, 2), (1, 1, 1), (33, 11, 129)])
def test_conv_stem_forward_backward(cuda, shape):
    """sp_conv_stem_fwd / sp_conv_stem_bwd alone (persistent work items, quad-shared taps) against torch
    fp32 on odd and even extents, fewer positions than one work item, more utterances than row groups."""
    from speech_amd._lib import check, lib, ptr, stream_ptr
    B, Fin, Tin = shape
    F, T = (Fin - 1) // 2 + 1, (Tin - 1) // 2 + 1
    g = torch.Generator().manual_seed(B * 1000 + Fin * 10 + Tin)
    spec = torch.randn(B, 1, Fin, Tin, generator=g).to(cuda)
    w = torch.randn(32, 1, 3, 3, generator=g).to(cuda)
    bias = torch.randn(32, generator=g).to(cuda)
    out = torch.full((B, F, T, 32), float("nan"), dtype=torch.bfloat16, device=cuda)
    check(lib().sp_conv_stem_fwd(ptr(spec), ptr(w), ptr(bias), B, Fin, Tin, ptr(out), stream_ptr()), "stem_fwd")
    ref = torch.nn.functional.conv2d(spec, w, bias, stride=2, padding=1).permute(0, 2, 3, 1)
    assert out.shape == ref.shape
    assert float((out.float() - ref).abs().max()) <= 8e-3 * float(ref.abs().max()) + 1e-6      # bf16 rounding of the output
    gout = torch.randn(B, F, T, 32, generator=g).to(cuda).bfloat16()
    gw, gb = torch.zeros(288, device=cuda), torch.zeros(32, device=cuda)
    check(lib().sp_conv_stem_bwd(ptr(spec), ptr(gout), B, Fin, Tin, ptr(gw), ptr(gb), stream_ptr()), "stem_bwd")
    gn = gout.float().permute(0, 3, 1, 2).contiguous()
    ref_w = torch.nn.grad.conv2d_weight(spec, (32, 1, 3, 3), gn, stride=2, padding=1)
    assert float((gw.view(32, 1, 3, 3) - ref_w).abs().max()) <= 2e-3 * float(ref_w.abs().max()) + 1e-5
    assert float((gb - gn.sum((0, 2, 3))).abs().max()) <= 2e-3 * float(gn.sum((0, 2, 3)).abs().max()) + 1e-5


def test_conv_stack_backward(cuda):
    """End to end against torch fp32 with loss = sum(z^2)/2.  bf16 activations flip the ReLU mask
    at ~2 % of the positions (near-zero pre-activations); with a random upstream gradient each
    flip moves a parameter-gradient sum by a whole |g| (measured 5-20 % of the peak), with this
    loss the upstream gradient vanishes where the last mask flips, leaving 0.2-5 %.  Bound: 8 %
    of the peak.  The kernels themselves are pinned tightly, on identical operands, by
    test_conv_kernels_individually."""

    B, Fin, Tin = 2, 24, 150
    _, (cnn, blocks, out), (cnn_r, blocks_r, ref) = _conv_ref_and_hip(cuda, B, Fin, Tin, 9)
    (0.5 * ref.float() ** 2).sum().backward()
    (0.5 * out.float() ** 2).sum().backward()
    def rel(a, b):
        return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)
    tol = 8e-2
    assert rel(cnn.weight.grad, cnn_r.weight.grad) <= tol
    assert rel(cnn.bias.grad, cnn_r.bias.grad) <= tol
    for a, b in zip(blocks, blocks_r):
        assert rel(a.conv1.weight.grad, b.conv1.weight.grad) <= tol
        assert rel(a.conv2.weight.grad, b.conv2.weight.grad) <= tol
        assert rel(a.bn1.weight.grad, b.bn1.weight.grad) <= tol
        assert rel(a.bn1.bias.grad, b.bn1.bias.grad) <= tol
        assert rel(a.bn2.weight.grad, b.bn2.weight.grad) <= tol
        assert rel(a.bn2.bias.grad, b.bn2.bias.grad) <= tol
        # conv bias before batch-statistics BN: zero gradient (torch returns rounding noise)
        assert float(a.conv1.bias.grad.abs().max()) == 0.0
        assert float(b.conv1.bias.grad.abs().max()) <= 1e-3 * float(b.conv1.weight.grad.abs().max()) + 1e-4


# =============================================================================== layer norm
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("shape", [(37, 1024), (5, 1280), (130, 64), (3, 2048)])
def test_layernorm_fwd_bwd(cuda, dtype, tol, shape):
    from speech_amd import ops
    M, N = shape
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, N, generator=g) * 2 + 0.5).to(dtype)
    w, b = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    gy = torch.randn(M, N, generator=g).to(dtype)
    xd = x.to(cuda).requires_grad_(True)
    wd, bd = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    y = ops.layer_norm(xd, wd, bd)
    y.backward(gy.to(cuda))
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (N,), wr, br)
    yr.backward(gy.double())
    def close(a, b_, what):
        a, b_ = a.detach().cpu().double(), b_.detach()
        assert float((a - b_).abs().max()) <= tol * max(1.0, float(b_.abs().max())), what
    close(y, yr, "y"); close(xd.grad, xr.grad, "gx"); close(wd.grad, wr.grad, "dgamma"); close(bd.grad, br.grad, "dbeta")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n_in,d,k,C", [(128, 64, 3, 1), (320, 64, 4, 32), (1280, 512, 4, 32)])
def test_sru_weight_pack_unpack(cuda, dtype, n_in, d, k, C):
    """sp_sru_pack_w / sp_sru_unpack_dw against the index permutation written with torch views."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(n_in + k)
    w = torch.randn(n_in, 2 * d * k, generator=g)
    ref = w.view(n_in, 2, d, k).permute(0, 3, 1, 2)                      # [n_in, k, 2, d]
    ref = ref.reshape(C, n_in // C, k, 2, d).transpose(0, 1).reshape(n_in, k * 2 * d)   # rows f*C + c
    out = ops.sru_pack_w(w.to(cuda), d, k, C, dtype)
    assert torch.equal(out.cpu(), ref.to(dtype))
    outT = ops.sru_pack_wt(w.to(cuda), d, k, C, dtype)                    # the same matrix, transposed
    assert outT.shape == (k * 2 * d, n_in) and torch.equal(outT.cpu(), ref.to(dtype).t())
    outT2, out2 = ops.sru_pack_wt(w.to(cuda), d, k, C, dtype, pad=64, both=True)   # both layouts, padded pitch
    assert outT2.stride(0) == n_in + 64 and torch.equal(outT2.cpu(), ref.to(dtype).t()) and torch.equal(out2, out)
    S = 3
    parts = torch.randn(S, n_in, k * 2 * d, generator=g).to(dtype)
    tot = parts.float().sum(0)
    back = tot.view(n_in // C, C, k, 2, d).transpose(0, 1).reshape(n_in, k, 2, d).permute(0, 2, 3, 1)
    got = ops.sru_unpack_dw(parts.to(cuda), n_in, d, k, C)
    assert torch.allclose(got.cpu(), back.reshape(n_in, -1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("time_major,V", [(False, 1000), (True, 29)])
def test_ctc_grad_in_bf16_is_the_rounded_f32_grad(cuda, time_major, V):
    """grad_dtype = bf16 (the fused training head's choice): bit-identical to rounding the f32 gradient,
    on the vectorised (V % 4 == 0) and the scalar store path, ragged input lengths included."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(V)
    B, T, L = 5, 77, 9
    logits = (torch.randn(B, T, V, generator=g) * 2).to(cuda)
    if time_major:
        logits = logits.transpose(0, 1).contiguous()
    labels = torch.randint(1, V, (B, L), generator=g).to(cuda)
    il = torch.tensor([77, 60, 77, 33, 70], dtype=torch.int32, device=cuda)
    ll = torch.tensor([9, 4, 7, 1, 9], dtype=torch.int32, device=cuda)
    l32, n32, g32 = ops.ctc_fwd_bwd(logits, labels, il, ll, max_label_len=L, time_major=time_major)
    l16, n16, g16 = ops.ctc_fwd_bwd(logits, labels, il, ll, max_label_len=L, time_major=time_major,
                                    grad_dtype=torch.bfloat16)
    assert g16.dtype == torch.bfloat16 and g16.stride() == g32.stride()
    assert torch.equal(l32, l16) and torch.equal(n32, n16)
    assert torch.equal(g16, g32.to(torch.bfloat16))


def test_mel_int16_pcm_equals_float_path(cuda):
    """int16 PCM input (flags bit 2): bit-identical to featurising pcm / 32768 as float32 -- the value
    torchaudio.load(normalize=True) gives the reference -- ragged batch, masks, padding included."""
    from speech_amd import data, ops
    g = torch.Generator().manual_seed(11)
    lens = [5000, 16000, 12345, 801]
    pcm = [torch.randint(-32768, 32768, (1, n), generator=g, dtype=torch.int32).to(torch.int16) for n in lens]
    offs = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64)
    masks = torch.tensor([[3, 10, 5, 20], [70, 10, 0, 0], [0, 0, 60, 17], [10, 5, 2, 3]], dtype=torch.int32)
    flat16 = torch.cat([w.reshape(-1) for w in pcm]).to(cuda)
    flat32 = flat16.to(torch.float32) / 32768.0
    a = ops.melspec(flat16, offs, masks=masks)
    b = ops.melspec(flat32, offs, masks=masks)
    assert a.shape == (4, 1, 80, 101) and torch.equal(a, b)
    # through the collate path: a list of int16 clips stays int16 up to the kernel
    c = data.featurize(pcm)
    d = data.featurize([w.float() / 32768.0 for w in pcm])
    assert torch.equal(c, d)
    # a mixed list falls back to float
    e = data.featurize([pcm[0], pcm[1].float() / 32768.0])
    assert torch.equal(e, data.featurize([pcm[0].float() / 32768.0, pcm[1].float() / 32768.0]))


def _bn_total(t):
    """The reduction a producer left in a `sums` buffer: its per-workgroup partials (count in the last word)."""
    n = int(t[512 + 1024 * 64: 512 + 1024 * 64 + 1].view(torch.int32))
    assert 1 <= n <= 1024
    return t[512: 512 + 64 * n].view(n, 64).double().sum(0).float()


def test_conv_fused_bn_sums_are_reproducible(cuda):
    """The statistics fused into the convolution epilogue equal sp_bn_stats of its output, and two
    launches give the same BITS (per-workgroup partials summed in a fixed order, no float atomics)."""
    from speech_amd import convstack as cs
    from speech_amd._lib import BN_SUMS_WORDS, check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    B, F, T = 3, 40, 333
    x = torch.randn(B, F, T, 32, generator=g).bfloat16().to(cuda)
    w = (torch.randn(32, 32, 3, 5, generator=g) * 0.1).to(cuda)
    bias = torch.randn(32, generator=g).to(cuda)
    s1 = torch.empty(BN_SUMS_WORDS, device=cuda)
    s2 = torch.full((BN_SUMS_WORDS,), float("nan"), device=cuda)     # nothing relies on a zeroed buffer
    y1 = cs._conv(x, None, None, cs.IN_ID, cs._pack(w, 0), bias, None, None, cs.EPI_NONE, B, F, T, s1)
    y2 = cs._conv(x, None, None, cs.IN_ID, cs._pack(w, 0), bias, None, None, cs.EPI_NONE, B, F, T, s2)
    n = int(s1[512 + 1024 * 64: 512 + 1024 * 64 + 1].view(torch.int32))
    assert torch.equal(y1, y2) and torch.equal(s1[512: 512 + 64 * n], s2[512: 512 + 64 * n])
    assert torch.equal(s1[512 + 1024 * 64: 513 + 1024 * 64].view(torch.int32), s2[512 + 1024 * 64: 513 + 1024 * 64].view(torch.int32))
    s3 = torch.empty(BN_SUMS_WORDS, device=cuda)
    check(lib().sp_bn_stats(ptr(y1), B * F * T, ptr(s3), stream_ptr()), "sp_bn_stats")
    ref = torch.cat([y1.float().reshape(-1, 32).sum(0), y1.float().reshape(-1, 32).pow(2).sum(0)])
    assert torch.allclose(_bn_total(s1), ref, rtol=1e-4, atol=1e-2)
    assert torch.allclose(_bn_total(s3), ref, rtol=1e-4, atol=1e-2)
    # the finaliser adds the partials into the eight rows in a fixed order: same bits on both buffers, and the
    # coefficients are those of torch's batch statistics
    outs = []
    for sb in (s1, s2):
        gamma, beta = torch.ones(32, device=cuda), torch.zeros(32, device=cuda)
        rm, rv = torch.zeros(32, device=cuda), torch.ones(32, device=cuda)
        coef, save = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
        check(lib().sp_bn_finalize_fwd(ptr(sb), B * F * T, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, 1,
                                       ptr(coef), ptr(save), stream_ptr()), "sp_bn_finalize_fwd")
        outs.append((coef, save, sb[:512].clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert torch.allclose(outs[0][2].view(8, 64).sum(0), ref, rtol=1e-4, atol=1e-2)
    yf = y1.float().reshape(-1, 32)
    assert torch.allclose(outs[0][1][:32], yf.mean(0), rtol=1e-3, atol=1e-3)
    assert torch.allclose(outs[0][1][32:], (yf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=2e-3)


# =============================================================================== hand-written GEMM
@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (300, 256, 96), (1000, 384, 1024), (24032, 1024, 3072),
                                   (24032, 3072, 1024), (12345, 2048, 192), (49155, 256, 64),
                                   (100, 65536, 64), (200001, 256, 128), (24032, 4096, 1280)])
def test_gemm_nt_bf16(cuda, M, N, K):
    """sp_gemm on K-contiguous bf16 operands (the persistent role-split kernel from 64 tiles up, the general
    kernel below) against an f32 matmul of the same operands (ragged M, strided rows)."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    b = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(cuda)
    out = ops.gemm(a, b)
    ref = a.float() @ b.float().t()
    err = float((out.float() - ref).abs().max())
    assert err <= 1e-2 * float(ref.abs().max()) + 1e-3          # bf16 output rounding
    if M >= 300:   # operands that are row slices of wider buffers
        aw = torch.randn(M, K + 64, generator=g).bfloat16().to(cuda)
        out2 = ops.gemm(aw[:, 32:32 + K].contiguous(), b)
        out3 = torch.empty(M, N + 8, dtype=torch.bfloat16, device=cuda)
        ops.gemm(aw[:, 32:32 + K].contiguous(), b, out=out3[:, :N])
        assert torch.equal(out2, out3[:, :N])


@pytest.mark.parametrize("form", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("dtype,M,N,K", [
    (torch.bfloat16, 393, 768, 256),       # T'=131 x B=3 rows: ragged in M
    (torch.bfloat16, 300, 29, 1024),       # char-vocabulary classifier: N = 29
    (torch.bfloat16, 257, 1000, 72),       # K not a multiple of the K step
    (torch.float16, 130, 136, 200),
    (torch.float32, 393, 384, 128),
    (torch.float32, 65, 29, 100),
    (torch.bfloat16, 1, 8, 8),
])
def test_gemm_general(cuda, form, dtype, M, N, K):
    """sp_gemm (the general MFMA kernel) in all four operand forms against an f64 matmul of the same
    inputs: f32 accumulation, so the error is the output rounding (f32 out: accumulation order only)."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(dtype).to(cuda)
    b = torch.randn(N, K, generator=g).to(dtype).to(cuda)
    av = a if form[0] == "n" else a.t().contiguous().t()          # same values, m-contiguous storage
    bv = b if form[1] == "t" else b.t().contiguous().t()          # "n": B stored [K][N]
    ref = a.double() @ b.double().t()
    out32 = ops.gemm(av, bv, out_dtype=torch.float32)
    tol = 1e-5 if dtype == torch.float32 else 2e-6 * K ** 0.5 + 1e-6
    assert float((out32.double() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    if dtype != torch.float32:
        out = ops.gemm(av, bv)
        assert out.dtype == dtype
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        assert float((out.double() - ref).abs().max()) <= eps * float(ref.abs().max()) + 1e-6
    # split-K slabs sum to the product; accumulation adds to what is there
    S = 3
    slabs = ops.gemm(av, bv, out_dtype=torch.float32, split_k=S)
    assert tuple(slabs.shape) == (S, M, N)
    assert float((slabs.double().sum(0) - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    base = torch.randn(M, N, generator=g).to(cuda)
    acc = base.clone()
    ops.gemm(av, bv, out=acc, accumulate=True)
    assert float((acc.double() - base.double() - ref).abs().max()) <= 2 * tol * float(ref.abs().max()) + 1e-5
    if (M * N) % 4 == 0:
        tgt = base.clone()
        sc = torch.tensor([0.5], device=cuda)
        ops.sum_slabs(slabs, tgt, True, sc)
        assert float((tgt.double() - base.double() - 0.5 * ref).abs().max()) <= 2 * tol * float(ref.abs().max()) + 1e-5


def test_gemm_general_unaligned_and_strided(cuda):
    """Row pitches that are not multiples of a 16-byte vector fall back to element-wise loads; padded
    pitches and an offset output view are honoured."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 77, 45, 29
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)            # pitch 29 elements
    b = torch.randn(N, K, generator=g).bfloat16().to(cuda)
    ref = a.double() @ b.double().t()
    out = ops.gemm(a, b, out_dtype=torch.float32)
    assert float((out.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    wide = torch.zeros(M, 64, device=cuda)
    ops.gemm(a, b, out=wide[:, 8:8 + N])
    assert float((wide[:, 8:8 + N].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float(wide[:, :8].abs().max()) == 0 and float(wide[:, 8 + N:].abs().max()) == 0
    ap = torch.zeros(M, 40, device=cuda).bfloat16()
    ap[:, :K] = a
    assert torch.equal(ops.gemm(ap[:, :K], b, out_dtype=torch.float32), out)


@pytest.mark.parametrize("Mi,Nj,K", [(1024, 3072, 24032), (1280, 4096, 24032), (1024, 1024, 24032), (256, 1024, 24032),
                                     (512, 768, 4100), (256, 256, 8191)])
def test_gemm_tn_splitk_bf16(cuda, Mi, Nj, K):
    """The weight-gradient form X^T G (both operands row-contiguous, K = T'*B rows, f32 split-K slabs) on the
    persistent TN kernel: transposed LDS reads, a ragged last K step (rows past K read as zeros), padded
    operand pitches; against an f64 product of the same bf16 inputs."""
    from speech_amd import _lib, ops
    g = torch.Generator().manual_seed(Mi + Nj + K)
    xb = torch.zeros(K, Mi + 64, dtype=torch.bfloat16, device=cuda)
    xb[:, :Mi] = torch.randn(K, Mi, generator=g).bfloat16().to(cuda)
    x = xb[:, :Mi]
    gu = (torch.randn(K, Nj, generator=g) * 0.1).bfloat16().to(cuda)
    S = _lib.lib().sp_gemm_tn_splits(Mi, Nj, K, _lib.SP_BF16)
    assert 1 <= S <= 64
    P = ops.gemm(x.t(), gu.t(), out_dtype=torch.float32, split_k=S)
    P = P if S > 1 else P.unsqueeze(0)
    ref = x.double().t() @ gu.double()
    assert float((P.double().sum(0) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # slabs into a wider buffer: pitch and slab stride are honoured
    wide = torch.full((S, Mi, Nj + 16), 7.0, device=cuda)
    ops.gemm(x.t(), gu.t(), out=wide[:, :, :Nj] if S > 1 else wide[0, :, :Nj], out_dtype=torch.float32, split_k=S)
    assert torch.equal(wide[:, :, :Nj], P) and float((wide[:, :, Nj:] - 7.0).abs().max()) == 0


# =============================================================================== full-size parity (configs[1] / [3])
@pytest.mark.parametrize("k,T,B", [(3, 751, 32), (4, 751, 32), (3, 1501, 64)])
def test_sru_recurrence_production_shape(cuda, k, T, B):
    """The recurrence at the production shapes (d = 512, bf16; 512 or 1024 workgroups, the LDS ring wrapping
    hundreds of times) against the f64 restatement.  Columns are independent, so the CPU reference runs on a
    32-channel slice of both directions over ALL utterances: outputs, gU, gx and the slices of the v / b
    gradients (which sum over the batch) are compared."""
    from speech_amd import ops
    d, nsel = 512, 32
    U, x, wc, bias, mask = _sru_inputs(T, B, d, k, 100 + k + T, torch.bfloat16)
    gen = torch.Generator().manual_seed(5)
    idx = torch.randperm(d, generator=gen)[:nsel].sort().values
    gh = torch.randn(T, B, 2 * d, generator=gen).bfloat16()
    Ud = U.to(cuda).requires_grad_(True)
    xd = x.to(cuda).requires_grad_(True) if x is not None else None
    wcd, bd = wc.to(cuda).requires_grad_(True), bias.to(cuda).requires_grad_(True)
    h = ops.sru_recurrence(Ud, xd, wcd, bd, mask.to(cuda), d)
    h.backward(gh.to(cuda))
    # the slice, as a d' = 64 problem
    Us = U.float()[:, :, :, :, idx]
    xs = x.float().view(T, B, 2, d)[:, :, :, idx].reshape(T, B, 2 * nsel) if x is not None else None
    sel4 = lambda v: v.view(4, d)[:, idx].reshape(-1)          # noqa: E731
    ms = mask.view(B, 2, d)[:, :, idx].reshape(B, 2 * nsel)
    href, (Ur, xr, wcr, br) = _sru_ref(Us, xs, sel4(wc), sel4(bias), ms, nsel)
    ghs = gh.double().view(T, B, 2, d)[:, :, :, idx].reshape(T, B, 2 * nsel)
    href.backward(ghs)

    def close(a, b, what, tol=2e-2):
        a, b = a.detach().cpu().double(), b.detach().double()
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), what
    close(h.view(T, B, 2, d)[:, :, :, idx].reshape(T, B, -1), href, "h")
    close(Ud.grad[:, :, :, :, idx], Ur.grad, "gU")
    if k == 3:
        close(xd.grad.view(T, B, 2, d)[:, :, :, idx].reshape(T, B, -1), xr.grad, "gx")
    # sums over T*B = 24 k .. 96 k bf16-rounded terms: relative to the sum's own size
    close(wcd.grad.view(4, d)[:, idx].reshape(-1), wcr.grad, "gwc", 1e-2)
    close(bd.grad.view(4, d)[:, idx].reshape(-1), br.grad, "gbias", 1e-2)


def test_ctc_production_shape_sentencepiece_vocab(cuda):
    """configs[1]'s loss shape -- T' = 751, B = 32, V = 1000, up to 60 pieces per utterance -- against torch's CPU
    CTC (f32 and f64) on a 4-utterance slice, and through the zero-row-sum property on all 32."""
    from speech_amd import ops
    T, B, V, L = 751, 32, 1000, 60
    g = torch.Generator().manual_seed(21)
    logits = torch.randn(T, B, V, generator=g) * 3
    labels = torch.randint(1, V, (B, L), generator=g)
    labels[:, 1::9] = labels[:, 0::9][:, : labels[:, 1::9].shape[1]]
    ll = torch.randint(L // 2, L + 1, (B,), generator=g).int()
    ll[0] = L
    il = torch.full((B,), T, dtype=torch.int32)
    loss, nll, grad = ops.ctc_fwd_bwd(logits.to(cuda), labels.to(cuda), il.to(cuda), ll.to(cuda), time_major=True,
                                      max_label_len=L)
    assert torch.isfinite(nll).all() and float(grad.sum(2).abs().max()) <= 1e-5
    sl = [0, 7, 19, 31]
    rl, rnll, rg32 = octc.ctc_torch(logits[:, sl], labels[sl], il[sl], ll[sl])
    _, _, rg64 = octc.ctc_torch(logits[:, sl], labels[sl], il[sl], ll[sl], dtype=torch.float64)
    assert np.allclose(nll[sl].cpu().numpy(), rnll.numpy(), rtol=1e-5)
    # the oracle's mean is over its 4 utterances, the kernel's over 32: compare d nll_b / d logits
    scale = np.array([B / len(sl)], dtype=np.float64)
    got = grad[:, sl].cpu().double().numpy() * scale
    g64 = rg64.numpy()
    gmax = np.abs(g64).max()
    assert np.abs(got - g64).max() <= 1e-4 * gmax
    ref32_err = np.abs(rg32.numpy() - g64).max()
    assert np.abs(got - rg32.numpy()).max() <= 1e-4 * gmax + 2 * ref32_err
    assert float(loss) == pytest.approx(float((nll.cpu().double() / ll.double()).mean()), rel=1e-6)


@pytest.mark.parametrize("cfg", [(1501, 8, 29, 450), (700, 3, 29, 300), (1100, 5, 40, 500), (1501, 3, 29, 600)])
def test_ctc_long_form_character_vocab(cuda, cfg):
    """configs[3] with the reference's characters (text.py:1-12): T' = 1501 frames, up to 450 labels -> 901 lattice
    states and the fused gradient kernel; further lattice sizes on either side of 1024 states (up to there one wave
    with sixteen states per lane, above four waves with eight), ragged input lengths.  nll AND gradient against torch's CPU CTC in float64
    (and float32) on a 2-utterance slice; zero row sums and zero rows beyond each length on the whole batch."""
    from speech_amd import ops
    T, B, V, L = cfg
    g = torch.Generator().manual_seed(T + L)
    logits = torch.randn(T, B, V, generator=g) * 3
    labels = torch.randint(1, V, (B, L), generator=g)
    labels[:, 1::9] = labels[:, 0::9][:, : labels[:, 1::9].shape[1]]           # adjacent repeats
    ll = torch.randint(L // 2, L + 1, (B,), generator=g).int()
    ll[0] = L
    il = torch.full((B,), T, dtype=torch.int32)
    il[-1] = T - 37                                                             # ragged
    loss, nll, grad = ops.ctc_fwd_bwd(logits.to(cuda), labels.to(cuda), il.to(cuda), ll.to(cuda), time_major=True,
                                      max_label_len=L)
    assert torch.isfinite(nll).all()
    assert float(grad.sum(2).abs().max()) <= 1e-5
    assert float(grad[T - 37:, -1].abs().max()) == 0.0
    sl = [0, B - 1]
    rl, rnll, rg32 = octc.ctc_torch(logits[:, sl], labels[sl], il[sl], ll[sl])
    _, rnll64, rg64 = octc.ctc_torch(logits[:, sl], labels[sl], il[sl], ll[sl], dtype=torch.float64)
    assert np.allclose(nll[sl].cpu().numpy(), rnll64.numpy(), rtol=2e-6)
    scale = np.array([B / len(sl)], dtype=np.float64)
    got = grad[:, sl].cpu().double().numpy() * scale
    g64 = rg64.numpy()
    gmax = np.abs(g64).max()
    assert np.abs(got - g64).max() <= 1e-4 * gmax
    ref32_err = np.abs(rg32.numpy() - g64).max()
    assert np.abs(got - rg32.numpy()).max() <= 1e-4 * gmax + 2 * ref32_err
    assert float(loss) == pytest.approx(float((nll.cpu().double() / ll.double()).mean()), rel=1e-6)


@pytest.mark.parametrize("cfg", [(300, 70, 29, 120), (64, 5, 29, 10), (200, 130, 40, 60), (520, 9, 64, 257)])
def test_ctc_character_vocab_route_batches_and_dtypes(cuda, cfg):
    """The whole-row gradient route of vocabularies of <= 64 classes (ctc_sparse_kernel<.., dense> up to 256 lattice
    states, ctc_char_grad_kernel above) at batches beyond 64 (no dense half exists on this route, so no batch limit),
    every lattice size class, a label that names the BLANK class, ragged lengths; f32 gradient against torch's f64 CPU
    run on a 2-utterance slice, bf16 / f16 gradients = the f32 gradient rounded once."""
    from speech_amd import ops
    T, B, V, L = cfg
    g = torch.Generator().manual_seed(T * 7 + B)
    logits = torch.randn(T, B, V, generator=g) * 2
    labels = torch.randint(1, V, (B, L), generator=g)
    if V <= L + 1:                                                              # the class-indexed route only (the
        labels[0, L // 2] = 0                                                   # position-indexed one never supported it):
    ll = torch.randint(max(L // 2, 1), L + 1, (B,), generator=g).int()         # a label equal to the blank index
    ll[0] = L
    il = torch.full((B,), T, dtype=torch.int32)
    il[-1] = T - 11
    dl, dlab, dil, dll = logits.to(cuda), labels.to(cuda), il.to(cuda), ll.to(cuda)
    loss, nll, grad = ops.ctc_fwd_bwd(dl, dlab, dil, dll, time_major=True, max_label_len=L)
    assert torch.isfinite(nll).all()
    assert float(grad[T - 11:, -1].abs().max()) == 0.0
    sl = [0, B - 1]
    _, rnll64, rg64 = octc.ctc_torch(logits[:, sl], labels[sl], il[sl], ll[sl], dtype=torch.float64)
    assert np.allclose(nll[sl].cpu().numpy(), rnll64.numpy(), rtol=2e-6)
    got = grad[:, sl].cpu().double().numpy() * (B / len(sl))
    assert np.abs(got - rg64.numpy()).max() <= 1e-4 * np.abs(rg64.numpy()).max()
    # 16-bit gradients are ONE rounding of (p - occupancy) * scale (the compiler fuses the scaling into the conversion),
    # so they are within half a 16-bit ulp of the f32 gradient and equal to its rounding except where that sits on a tie
    _, _, g32 = ops.ctc_fwd_bwd(dl, dlab, dil, dll, time_major=True, max_label_len=L, grad_scale=32.0)
    for dt, half_ulp, tiny in ((torch.bfloat16, 2.0 ** -8, 2.0 ** -134), (torch.float16, 2.0 ** -11, 2.0 ** -25)):
        _, _, g16 = ops.ctc_fwd_bwd(dl, dlab, dil, dll, time_major=True, max_label_len=L, grad_scale=32.0, grad_dtype=dt)
        assert g16.dtype == dt
        assert bool(((g16.float() - g32).abs() <= g32.abs() * half_ulp * 1.0001 + tiny).all())
        assert float((g16 != g32.to(dt)).float().mean()) < 1e-3


def test_ctc_random_shapes_against_torch_f64(cuda):
    """Differential sweep over the routes of sp_ctc_fwd_bwd: 80 seeded random shapes (T <= 90, B <= 7, V from 2 to 80,
    label lengths up to 300 where T allows: class-indexed and position-indexed rows, every lattice size class up to
    sixteen states per lane, ragged input lengths, empty labels, infeasible utterances); nll and gradient against
    torch's float64 CPU CTC (zero_infinity semantics checked on the feasible utterances only)."""
    from speech_amd import ops
    rng = np.random.RandomState(20251)
    ngrad = ninf = 0
    for case in range(80):
        B = int(rng.randint(1, 8))
        V = int(rng.choice([2, 3, 5, 17, 29, 33, 64, 65, 80]))
        big = case % 4 == 0
        T = int(rng.randint(200, 640)) if big else int(rng.randint(1, 90))
        L = int(rng.randint(1, min(300, T))) if big else int(rng.randint(0, max(1, T // 2) + 1))
        g = torch.Generator().manual_seed(1000 + case)
        logits = torch.randn(T, B, V, generator=g) * float(rng.choice([0.5, 2.0, 5.0]))
        labels = torch.randint(1, V, (B, max(L, 1)), generator=g)
        ll = torch.randint(0, L + 1, (B,), generator=g).int()
        ll[0] = L
        il = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g).int()
        il[0] = T
        loss, nll, grad = ops.ctc_fwd_bwd(logits.to(cuda), labels.to(cuda), il.to(cuda), ll.to(cuda), time_major=True,
                                          max_label_len=max(L, 1))
        _, rnll, rg = octc.ctc_torch(logits, labels, il, ll, dtype=torch.float64)
        rn = rnll.numpy()
        feas = np.isfinite(rn) & (rn < 1e29)
        got_n = nll.cpu().numpy()
        assert np.allclose(got_n[feas], rn[feas], rtol=2e-6, atol=1e-5), (case, T, B, V, L)
        assert np.all(~np.isfinite(got_n[~feas]) | (got_n[~feas] > 1e29)), (case, T, B, V, L)
        ninf += int((~feas).sum())
        if feas.all():
            ngrad += 1
            gmax = max(float(np.abs(rg.numpy()).max()), 1e-30)
            assert float(np.abs(grad.cpu().double().numpy() - rg.numpy()).max()) <= 1e-4 * gmax, (case, T, B, V, L)
    assert ngrad >= 20 and ninf >= 5, (ngrad, ninf)      # the sweep really holds both kinds


def test_mel_log_flag(cuda):
    """flags bit 0: natural log of (mel + 1e-6), the optional log stage north_star names (off by default: the
    reference feeds magnitudes).  Against log(oracle + 1e-6); padding frames hold log(1e-6)."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(4)
    lens = [4000, 2777, 1601]
    wavs = [torch.randn(1, n, generator=g) * 0.1 for n in lens]
    flat = torch.cat([w.reshape(-1) for w in wavs]).to(cuda)
    offs = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64)
    out = ops.melspec(flat, offs.to(cuda), flags=1).cpu()
    ref = omel.collate(wavs)
    T = ref.shape[-1]
    for b, n in enumerate(lens):
        tb = 1 + n // 160
        r = torch.log(ref[b, 0, :, :tb] + 1e-6)
        assert float((out[b, 0, :, :tb] - r).abs().max()) <= 1e-4 * float(r.abs().max())
        if tb < T:       # padding is written in the output's domain: log(0 + 1e-6)
            pad = out[b, 0, :, tb:]
            assert float((pad - np.log(1e-6)).abs().max()) <= 1e-4 or float(pad.abs().max()) == 0


@pytest.mark.parametrize("log", [0, 1])
def test_mel_per_utterance_normalisation(cuda, log):
    """flags bit 1 (with and without the log stage): every mel row of every utterance has zero mean and unit
    variance over its valid frames, SpecAugment bands are zeroed AFTER the statistics, padding stays 0; against
    the same arithmetic on the oracle's features."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(8)
    lens = [4800, 3333, 1777]
    wavs = [torch.randn(1, n, generator=g) * 0.1 + 0.05 * torch.sin(torch.arange(n) * 0.3) for n in lens]
    flat = torch.cat([w.reshape(-1) for w in wavs]).to(cuda)
    offs = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64)
    masks = torch.tensor([[5, 7, 3, 9], [0, 0, 0, 0], [70, 10, 8, 2]], dtype=torch.int32)
    out = ops.melspec(flat, offs.to(cuda), masks=masks, flags=2 | log).cpu()
    ref = omel.collate(wavs)
    for b, n in enumerate(lens):
        tb = 1 + n // 160
        x = ref[b, 0, :, :tb].double()
        if log:
            x = torch.log(x + 1e-6)
        y = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
        f0, fw, t0, tw = masks[b].tolist()
        y[f0:f0 + fw] = 0
        y[:, t0:t0 + tw] = 0
        assert float((out[b, 0, :, :tb].double() - y).abs().max()) <= 2e-4 * float(y.abs().max())
        assert float(out[b, 0, :, tb:].abs().max() if tb < out.shape[-1] else 0.0) == 0


@pytest.mark.parametrize("M,N,K,out_dtype", [(24032, 3072, 1024, torch.float16), (24032, 1024, 3072, torch.float16),
                                            (5000, 1024, 1024, torch.float32), (3000, 384, 192, torch.float16)])
def test_gemm_nt_fp16(cuda, M, N, K, out_dtype):
    """fp16 operands (the reference's evaluation type, model.half()) on the same role-split kernel
    (v_mfma_f32_16x16x32_f16), f16 or f32 out."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = torch.randn(M, K, generator=g).half().to(cuda)
    b = (torch.randn(N, K, generator=g) * 0.1).half().to(cuda)
    out = ops.gemm(a, b, out_dtype=out_dtype)
    ref = a.float() @ b.float().t()
    tol = 1e-3 if out_dtype == torch.float16 else 1e-5
    assert out.dtype == out_dtype
    assert float((out.float() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-4


# ------------------------------------------------------------------------------------------
# beam search (scripts/eval_dataset.py:20-27, :43-51): the GPU candidate stage and the whole decoder
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("V", [29, 1000])
def test_beam_candidates_match_the_oracle(cuda, dtype, V):
    from oracle import beam as obeam
    from speech_amd import ops
    g = torch.Generator().manual_seed(V)
    B, T = 3, 57
    x = (torch.randn(B, T, V, generator=g) * 4).to(dtype)
    fl = torch.tensor([T, 31, 0], dtype=torch.int32)
    cn, ci, cl = ops.beam_candidates(x.to(cuda), -5.0, fl)
    cn, ci, cl = cn.cpu().numpy(), ci.cpu().numpy(), cl.cpu().numpy()
    assert ci.shape[2] == min(V, 150)
    for b in range(B):
        want = obeam.candidates(x[b].float().numpy(), -5.0)
        for t in range(T):
            if t >= fl[b]:
                assert cn[b, t] == 0
                continue
            w = want[t]
            got_ids = ci[b, t, : cn[b, t]].tolist()
            lp = dict(zip(got_ids, cl[b, t, : cn[b, t]].tolist()))
            # a token within float rounding of the threshold may fall on either side
            sure = [i for i, p in w if abs(p + 5.0) > 1e-3]
            assert got_ids == sorted(got_ids)
            assert set(sure) <= set(got_ids) and all(abs(lp[i] - p) < 2e-4 for i, p in w if i in lp)
            assert all(i in dict(w) or abs(lp[i] + 5.0) < 1e-3 for i in got_ids)


def test_beam_decoder_end_to_end(cuda, tmp_path):
    """decoder.build_ctcdecoder(...).decode_batch on device logits = the oracle's search on the same logits;
    a peaked lattice decodes to its greedy text; the language model changes the winner where it should."""
    from oracle import beam as obeam
    from speech_amd import decoder, text
    labels = list(text.CHARS)
    arpa = tmp_path / "lm.arpa"
    arpa.write_text("\\data\\\nngram 1=6\nngram 2=2\n\n\\1-grams:\n-1.5\t<unk>\n-99\t<s>\t-0.3\n-1.0\t</s>\n"
                    "-0.5\tthe\t-0.2\n-0.8\tcat\t-0.2\n-2.5\ttha\n\n\\2-grams:\n-0.2\tthe cat\n-0.3\t<s> the\n\n\\end\\\n")
    dec0 = decoder.build_ctcdecoder(labels, None)
    dec1 = decoder.build_ctcdecoder(labels, str(arpa), alpha=0.3, beta=1.0)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 40, len(labels), generator=g) * 3
    got0 = dec0.decode_batch(None, x.to(cuda), beam_width=16)
    got1 = dec1.decode_batch(None, x.half().to(cuda), beam_width=16)
    lm = obeam.ArpaLM(str(arpa), 0.3, 1.0)
    for b in range(5):
        assert got0[b] == obeam.prefix_beam(x[b].numpy(), dec0.labels, dec0.blank, None, 16)[0][0]
        assert got1[b] == obeam.prefix_beam(x[b].half().float().numpy(), dec0.labels, dec0.blank, lm, 16)[0][0]
    # a peaked lattice spelling "the cat": beam = greedy; numpy input and a list of ragged arrays work too
    ids = [labels.index(c) for c in "the cat"]
    frames = []
    for i in ids:
        frames += [i, 0]
    peaked = torch.full((len(frames), len(labels)), -8.0)
    peaked[torch.arange(len(frames)), torch.tensor(frames)] = 8.0
    assert dec0.decode(peaked.numpy()) == "the cat" and dec1.decode(peaked.to(cuda)) == "the cat"
    assert dec0.decode_batch(None, [peaked.numpy(), peaked[:6].numpy()]) == ["the cat", "the"]
    # ambiguous last vowel of the first word: acoustics prefer "tha", the language model "the"
    amb = peaked.clone()
    pos = frames.index(labels.index("e"))
    amb[pos, labels.index("e")] = 2.0
    amb[pos, labels.index("a")] = 2.6
    assert dec0.decode(amb.to(cuda)) == "tha cat" and dec1.decode(amb.to(cuda)) == "the cat"
    beams = dec1.decode_beams(amb.to(cuda), n_best=3)
    assert beams[0][0] == "the cat" and beams[0][2] >= beams[1][2]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_pack(cuda, dtype):
    """sp_linear_pack: padded 16-bit weight and scaled transpose against torch, ragged V and n."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(5)
    for V, n, Vp in [(1000, 1024, 1024), (29, 1024, 256), (70, 96, 70), (129, 200, 192)]:
        w = torch.randn(V, n, generator=g).to(cuda)
        wb = ops._linear_pack(w, Vp, dtype)
        assert wb.shape == (Vp, n)
        assert torch.equal(wb[:V], w.to(dtype))
        assert float(wb[V:].float().abs().max()) == 0.0 if Vp > V else True
        sc = torch.tensor(0.37, device=cuda)
        wt = ops._linear_pack(w, Vp, dtype, sc, transposed=True)
        assert wt.shape == (n, Vp)
        # scale and rounding are one instruction on the device (single rounding): torch's product is rounded to
        # f32 first, so an occasional element differs by one unit in the last place
        ref = (w * sc).t()
        ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
        assert float(((wt[:, :V].float() - ref).abs() / (ref.abs() + 1e-4)).max()) <= ulp
        assert float((wt[:, :V] != ref.to(dtype)).float().mean()) < 1e-2
        assert float(wt[:, V:].float().abs().max()) == 0.0 if Vp > V else True
        assert torch.equal(ops._linear_pack(w, Vp, dtype, None, transposed=True)[:, :V], w.to(dtype).t())


def test_ctc_sharp_logits_range(cuda):
    """Logits scaled until the mean loss is ~13, ~30 and ~45 bits PER FRAME (random labels against sharp random
    scores: every frame shrinks the lattice by that much): the scaled-probability recursion with its 8-frame
    rescale blocks must still agree with torch's float64 log-space CTC, loss and gradient."""
    from speech_amd import ops
    B, T, V, L = 4, 300, 1000, 40
    for scale in (3.0, 8.0, 12.0):
        g = torch.Generator().manual_seed(3)
        logits = torch.randn(T, B, V, generator=g) * scale
        labels = torch.randint(1, V, (B, L), generator=g)
        ll = torch.tensor([40, 33, 20, 40], dtype=torch.int32)
        il = torch.full((B,), T, dtype=torch.int32)
        z = logits.double().requires_grad_(True)
        ref = torch.nn.functional.ctc_loss(torch.log_softmax(z, -1), labels, il.long(), ll.long(), blank=0, reduction="none")
        (ref / ll.double().clamp(min=1)).mean().backward()
        _, nll, grad = ops.ctc_fwd_bwd(logits.to(cuda).transpose(0, 1), labels.to(cuda), il.to(cuda), ll.to(cuda), max_label_len=L)
        assert float(((nll.cpu().double() - ref.detach()).abs() / ref.detach()).max()) <= 1e-6, scale
        gr = z.grad.transpose(0, 1)
        assert bool(torch.isfinite(grad).all())
        assert float((grad.cpu().double() - gr).abs().max()) <= 1e-4 * float(gr.abs().max()), scale
