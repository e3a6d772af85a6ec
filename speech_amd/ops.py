"""Tensor-level wrappers over the C ABI (libspeech_hip.so).  torch supplies device memory, the
current HIP stream and the autograd graph; the arithmetic of the hot path -- frontend, convolutions,
normalisations, every matrix product, recurrence, CTC, decode -- runs in the hand-written gfx950
kernels.  What is left to torch are dtype casts and scalings of parameter-sized tensors and the
dropout masks' random draws."""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

HOP = 160
N_FFT = 400
N_MELS = 80
N_FREQS = 201


def _need_cuda(t, name):
    if not t.is_cuda:
        raise _lib.SpeechHipError(f"{name} must be a CUDA/HIP tensor (no CPU path exists)")


# ------------------------------------------------------------------------------------------
# Frontend (data.py:15-24, :174-175)
# ------------------------------------------------------------------------------------------
def melscale_fbanks_htk(n_freqs=N_FREQS, f_min=0.0, f_max=8000.0, n_mels=N_MELS, sample_rate=16000):
    """HTK triangular filterbank [n_freqs, n_mels], norm=None -- the matrix the reference's
    MelSpectrogram(sample_rate=16000, n_fft=400, n_mels=80) builds (data.py:16-18)."""
    import math
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class MelPlan:
    """Device-resident constants of the frontend kernel (window, twiddles, filterbank tiles)."""

    def __init__(self, device, fbank=None):
        fb = melscale_fbanks_htk() if fbank is None else fbank
        fb = fb.detach().to("cpu", torch.float32).contiguous()
        if tuple(fb.shape) != (N_FREQS, N_MELS):
            raise ValueError(f"filterbank must be [{N_FREQS}, {N_MELS}], got {tuple(fb.shape)}")
        self.device = torch.device(device)
        nbytes = lib().sp_melspec_plan_bytes()
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().sp_melspec_plan(ctypes.c_void_p(fb.data_ptr()), ptr(self.buf), stream_ptr()),
                  "sp_melspec_plan")


_plans = {}


def default_plan(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _plans:
        _plans[key] = MelPlan(device)
    return _plans[key]


def num_frames(n_samples):
    return 1 + n_samples // HOP


def melspec(wav, offsets, t_max=None, masks=None, flags=0, plan=None, out=None, offsets_host=None):
    """wav: flat f32 (or int16 PCM, read as sample / 32768) device tensor of concatenated utterances;
    offsets: int64 [B+1] (device or host) -> mel [B, 1, 80, T_max] f32.  masks: int32 [B,4] (f0, fw, t0, tw) or None.
    offsets_host: the same offsets on the host, when the caller has them (a loader does): with device
    `offsets` alone the shapes have to be fetched back, which drains the stream once per call."""
    _need_cuda(wav, "wav")
    if wav.dtype not in (torch.float32, torch.int16) or not wav.is_contiguous():
        raise ValueError("wav must be a contiguous float32 (or int16 PCM) tensor")
    if wav.dtype == torch.int16:
        flags |= 4           # samples are read as pcm / 32768
    if offsets_host is not None:
        offs_host = torch.as_tensor(offsets_host, dtype=torch.int64)
        if offs_host.is_cuda or offs_host.numel() != offsets.numel():
            raise ValueError("offsets_host must be a host copy of offsets")
    else:
        offs_host = offsets.detach().cpu().to(torch.int64) if offsets.is_cuda else offsets.to(torch.int64)
    B = offs_host.numel() - 1
    lens = offs_host[1:] - offs_host[:-1]
    if B > 0:
        if int(lens.min()) <= N_FFT // 2:
            raise ValueError("every utterance needs more than 200 samples (reflect padding)")
        if int(offs_host[-1]) > wav.numel() or int(offs_host[0]) < 0:
            raise ValueError("offsets exceed the waveform buffer")
    tb_max = int(lens.max()) // HOP + 1 if B > 0 else 0
    if t_max is None:
        t_max = tb_max
    if t_max < tb_max:
        raise ValueError(f"t_max={t_max} smaller than the longest utterance ({tb_max} frames)")
    offs_dev = offsets if offsets.is_cuda and offsets.dtype == torch.int64 else offs_host.to(wav.device)
    if masks is not None:
        if not masks.is_cuda:     # pinned staging + async copy: a pageable copy would wait for the stream
            masks = masks.to(torch.int32).contiguous().pin_memory()
        masks = masks.to(device=wav.device, dtype=torch.int32, non_blocking=True).contiguous()
        if tuple(masks.shape) != (B, 4):
            raise ValueError("masks must be [B, 4]")
    if plan is None:
        plan = default_plan(wav.device)
    if out is None:
        out = torch.empty((B, 1, N_MELS, t_max), dtype=torch.float32, device=wav.device)
    with torch.cuda.device(wav.device):
        check(lib().sp_melspec_fwd(ptr(wav), ptr(offs_dev), B, t_max, ptr(plan.buf), ptr(masks), flags,
                                   ptr(out), stream_ptr()), "sp_melspec_fwd")
    return out


# ------------------------------------------------------------------------------------------
# CTC (train.py:69-77)
# ------------------------------------------------------------------------------------------
_ctc_ws = {}


def _workspace(key, nbytes, device):
    buf = _ctc_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ctc_ws[key] = buf
    return buf


def ctc_fwd_bwd(logits, labels, input_lengths, label_lengths, blank=0, max_label_len=None,
                time_major=False, grad_scale=1.0, want_grad=True, grad_dtype=torch.float32, grad_out=None):
    """logits [B,T,V] (or [T,B,V] when time_major) f32 -> (loss[1], nll[B], grad like logits).

    loss = mean_b nll_b / max(L_b, 1) of log_softmax(logits); grad = grad_scale * dloss/dlogits, stored as
    grad_dtype (float32, or bfloat16 for a bf16 consumer: rounded once, in the kernel's store).
    """
    if grad_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("grad_dtype must be float32, bfloat16 or float16")
    _need_cuda(logits, "logits")
    if logits.dtype != torch.float32:
        raise ValueError("logits must be float32 (the reference computes CTC in fp32)")
    if logits.stride(2) != 1:
        logits = logits.contiguous()
    if time_major:
        T, B, V = logits.shape
        st, sb = logits.stride(0), logits.stride(1)
    else:
        B, T, V = logits.shape
        sb, st = logits.stride(0), logits.stride(1)
    dev = logits.device
    labels = labels.to(device=dev, dtype=torch.int64)
    if labels.dim() != 2 or labels.shape[0] != B:
        raise ValueError("labels must be [B, Lmax]")
    if labels.stride(1) != 1 and labels.shape[1] > 1:
        labels = labels.contiguous()
    if labels.numel() == 0:
        labels = torch.zeros((B, 1), dtype=torch.int64, device=dev)
    if max_label_len is None:
        max_label_len = int(labels.shape[1])
        if max_label_len > 1023:
            raise ValueError(f"labels are padded to {max_label_len} > 1023 columns: pass max_label_len (<= 1023), "
                             "a true bound on label_lengths")
    # lengths that are still on the host can be checked without a device round trip (device lengths are clamped by
    # the kernel, as the header says)
    if not label_lengths.is_cuda and label_lengths.numel() and int(label_lengths.max()) > max_label_len:
        raise ValueError(f"label_lengths.max() = {int(label_lengths.max())} exceeds max_label_len = {max_label_len}")
    input_lengths = input_lengths.to(device=dev, dtype=torch.int32).contiguous()
    label_lengths = label_lengths.to(device=dev, dtype=torch.int32).contiguous()
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    if grad_out is not None:      # caller-owned gradient buffer, same strides (in elements) as the logits
        if grad_out.stride() != logits.stride() or grad_out.shape != logits.shape or grad_out.dtype != grad_dtype:
            raise ValueError("grad_out must match the logits' shape and strides and have grad_dtype")
        grad = grad_out
    else:
        grad = torch.empty_strided(logits.shape, logits.stride(), dtype=grad_dtype, device=dev) if want_grad else None
    nbytes = lib().sp_ctc_workspace_bytes(T, B, V, max_label_len)
    ws = _workspace((dev.index, "ctc"), nbytes, dev)
    with torch.cuda.device(dev):
        check(lib().sp_ctc_fwd_bwd(ptr(logits), st, sb, ptr(labels), labels.stride(0), ptr(input_lengths),
                                   ptr(label_lengths), T, B, V, blank, max_label_len, float(grad_scale),
                                   ptr(nll), ptr(loss), ptr(grad), _lib.dtype_code(grad) if grad is not None else 0,
                                   ptr(ws), ws.numel(), stream_ptr()), "sp_ctc_fwd_bwd")
    return loss, nll, grad


class _CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, input_lengths, label_lengths, blank, time_major, max_label_len):
        loss, nll, grad = ctc_fwd_bwd(logits, labels, input_lengths, label_lengths, blank,
                                      max_label_len, time_major, 1.0, logits.requires_grad)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None


def ctc_loss(logits, labels, input_lengths, label_lengths, blank=0, time_major=False,
             max_label_len=None):
    """Differentiable fused log_softmax + CTC (mean reduction), same value as
    ``CTCLoss(blank)(log_softmax(logits), ...)`` in the reference (train.py:69-77)."""
    return _CTCLossFn.apply(logits, labels, input_lengths, label_lengths, blank, time_major,
                            max_label_len)


# ------------------------------------------------------------------------------------------
# Greedy decode (decoder.py:28-57)
# ------------------------------------------------------------------------------------------
def greedy_decode_ids(scores, blank=0, frame_lengths=None):
    """scores [B,T,V] (f32/f16/bf16, device) -> (ids int32 [B,T] padded with -1, lengths int32 [B])."""
    _need_cuda(scores, "scores")
    if scores.dim() != 3:
        raise ValueError("scores must be [B, T, V]")
    scores = scores.contiguous()
    B, T, V = scores.shape
    ids = torch.empty((B, T), dtype=torch.int32, device=scores.device)
    lengths = torch.empty(B, dtype=torch.int32, device=scores.device)
    fl = None
    if frame_lengths is not None:
        fl = frame_lengths.to(device=scores.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(scores.device):
        check(lib().sp_greedy_decode(ptr(scores), _lib.dtype_code(scores), B, T, V, blank, ptr(fl),
                                     ptr(ids), ptr(lengths), stream_ptr()), "sp_greedy_decode")
    return ids, lengths


def edit_distance(a, b):
    """Unit-cost Levenshtein distance between two int sequences (host, native)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.int32))
    r = lib().sp_edit_distance(ctypes.c_void_p(a.ctypes.data), a.size, ctypes.c_void_p(b.ctypes.data), b.size)
    if r < 0:
        raise _lib.SpeechHipError("sp_edit_distance failed")
    return r


# ------------------------------------------------------------------------------------------
# Beam search (scripts/eval_dataset.py:20-27, :43-51)
# ------------------------------------------------------------------------------------------
def beam_kmax(V, token_min_logp):
    """Candidate-list width that cannot overflow: at most floor(exp(-min)) tokens reach the threshold."""
    if token_min_logp <= -34.0:
        return int(V)
    return int(min(V, math.floor(math.exp(-token_min_logp)) + 2))


def beam_candidates(scores, token_min_logp=-5.0, frame_lengths=None, kmax=None):
    """scores [B,T,V] logits on the device -> (cand_n int32 [B,T], cand_id int32 [B,T,kmax],
    cand_lp f32 [B,T,kmax]) on the device: per frame the tokens a beam search expands."""
    _need_cuda(scores, "scores")
    if scores.dim() != 3:
        raise ValueError("scores must be [B, T, V]")
    scores = scores.contiguous()
    B, T, V = scores.shape
    kmax = beam_kmax(V, token_min_logp) if kmax is None else int(kmax)
    cand_n = torch.empty((B, T), dtype=torch.int32, device=scores.device)
    cand_id = torch.empty((B, T, kmax), dtype=torch.int32, device=scores.device)
    cand_lp = torch.empty((B, T, kmax), dtype=torch.float32, device=scores.device)
    fl = None
    if frame_lengths is not None:
        fl = torch.as_tensor(frame_lengths).to(device=scores.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(scores.device):
        check(lib().sp_beam_candidates(ptr(scores), _lib.dtype_code(scores), B, T, V, float(token_min_logp), ptr(fl),
                                       kmax, ptr(cand_n), ptr(cand_id), ptr(cand_lp), stream_ptr()), "sp_beam_candidates")
    return cand_n, cand_id, cand_lp


def ctc_beam_decode(cand_n, cand_id, cand_lp, codepoints, blank, space, lm_handle=None, beam_width=100,
                    beam_prune_logp=-10.0, frame_lengths=None, n_threads=16, n_best=1, max_out=None):
    """Host prefix beam search over candidate lists (numpy arrays, shapes as beam_candidates returns).
    -> (ids int32 [B,n_best,max_out], lengths int32 [B,n_best], scores f32 [B,n_best,2])."""
    cand_n = np.ascontiguousarray(cand_n, dtype=np.int32)
    cand_id = np.ascontiguousarray(cand_id, dtype=np.int32)
    cand_lp = np.ascontiguousarray(cand_lp, dtype=np.float32)
    B, T = cand_n.shape
    kmax = cand_id.shape[2]
    if cand_id.shape != (B, T, kmax) or cand_lp.shape != (B, T, kmax):
        raise ValueError("candidate arrays disagree in shape")
    cps = np.ascontiguousarray(codepoints, dtype=np.int32)
    max_out = T if max_out is None else int(max_out)
    fl = None if frame_lengths is None else np.ascontiguousarray(frame_lengths, dtype=np.int32)
    out_ids = np.empty((B, n_best, max(max_out, 1)), dtype=np.int32)
    out_len = np.empty((B, n_best), dtype=np.int32)
    out_sc = np.empty((B, n_best, 2), dtype=np.float32)
    vp = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
    check(lib().sp_ctc_beam_decode(vp(cand_n), vp(cand_id), vp(cand_lp), B, T, kmax, vp(fl), vp(cps), cps.size,
                                   int(blank), int(space), lm_handle, int(beam_width), float(beam_prune_logp),
                                   int(n_threads), int(n_best), max(max_out, 1), vp(out_ids), vp(out_len), vp(out_sc)),
          "sp_ctc_beam_decode")
    return out_ids, out_len, out_sc


# ------------------------------------------------------------------------------------------
# SRU recurrence (net.py:61-72)
# ------------------------------------------------------------------------------------------
def sru_fwd(U, xres, wc, bias, mask, T, B, d, k, keep_c=True):
    """-> (h, c); keep_c=False (inference) skips the cell-state store and returns c = None."""
    _need_cuda(U, "U")
    h = torch.empty((T, B, 2 * d), dtype=U.dtype, device=U.device)
    # cell states kept for the backward: bf16 on the bf16 kernels (as the sru package does), f32 otherwise
    cdt = torch.bfloat16 if U.dtype == torch.bfloat16 else torch.float32
    c = torch.empty((T, B, 2 * d), dtype=cdt, device=U.device) if keep_c else None
    with torch.cuda.device(U.device):
        check(lib().sp_sru_fwd(ptr(U), ptr(xres), ptr(wc), ptr(bias), ptr(mask), T, B, d, k,
                               _lib.dtype_code(U), ptr(h), ptr(c), stream_ptr()), "sp_sru_fwd")
    return h, c


def sru_bwd(U, xres, wc, bias, mask, c, gh, T, B, d, k, gwc=None, gbias=None):
    """gwc / gbias: optional f32 [4d] buffers the kernel ACCUMULATES into (default: fresh zeros)."""
    gU = torch.empty_like(U)
    gx = torch.empty((T, B, 2 * d), dtype=U.dtype, device=U.device) if k == 3 else None
    if gwc is None:
        gwc = torch.zeros(4 * d, dtype=torch.float32, device=U.device)
    if gbias is None:
        gbias = torch.zeros(4 * d, dtype=torch.float32, device=U.device)
    with torch.cuda.device(U.device):
        check(lib().sp_sru_bwd(ptr(U), ptr(xres), ptr(wc), ptr(bias), ptr(mask), ptr(c), ptr(gh), T, B,
                               d, k, _lib.dtype_code(U), ptr(gU), ptr(gx), ptr(gwc), ptr(gbias),
                               stream_ptr()), "sp_sru_bwd")
    return gU, gx, gwc, gbias


class _SRURecurrenceFn(torch.autograd.Function):
    """U [T,B,k,2,d], xres [T,B,2d] or None (k == 4), wc/bias [4d] f32, mask [B,2d] f32 or None."""

    @staticmethod
    def forward(ctx, U, xres, wc, bias, mask, d):
        T, B = U.shape[0], U.shape[1]
        k = U.shape[2]
        U = U.contiguous()
        xr = xres.contiguous() if xres is not None else None
        wc32, b32 = wc.float().contiguous(), bias.float().contiguous()
        h, c = sru_fwd(U, xr, wc32, b32, mask, T, B, d, k)
        ctx.save_for_backward(U, xr, wc32, b32, mask, c)
        ctx.dims = (T, B, d, k)
        ctx.param_dtypes = (wc.dtype, bias.dtype)
        return h

    @staticmethod
    def backward(ctx, gh):
        U, xr, wc32, b32, mask, c = ctx.saved_tensors
        T, B, d, k = ctx.dims
        gU, gx, gwc, gbias = sru_bwd(U, xr, wc32, b32, mask, c, gh.contiguous(), T, B, d, k)
        return gU, gx, gwc.to(ctx.param_dtypes[0]), gbias.to(ctx.param_dtypes[1]), None, None


def sru_recurrence(U, xres, wc, bias, mask, d):
    return _SRURecurrenceFn.apply(U, xres, wc, bias, mask, d)


# ------------------------------------------------------------------------------------------
# LayerNorm and the fused SRU layer (net.py:61-76)
# ------------------------------------------------------------------------------------------
def layernorm_fwd(x2d, gamma, beta, eps, pad=0):
    """pad: extra elements per output row (multiple of 8): y is then an [M, N] view of an [M, N + pad]
    buffer (see sp_layernorm_fwd's ldy)."""
    M, N = x2d.shape
    y = torch.empty((M, N + pad), dtype=x2d.dtype, device=x2d.device)[:, :N] if pad else torch.empty_like(x2d)
    mean = torch.empty(M, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        check(lib().sp_layernorm_fwd(ptr(x2d), ptr(gamma), ptr(beta), M, N, float(eps), _lib.dtype_code(x2d),
                                     ptr(y), y.stride(0), ptr(mean), ptr(rstd), stream_ptr()), "sp_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(gy2d, x2d, mean, rstd, gamma, gres2d=None, dgamma=None, dbeta=None):
    """dgamma / dbeta: optional f32 [N] buffers the kernel ACCUMULATES into (default: fresh zeros)."""
    M, N = x2d.shape
    gx = torch.empty_like(x2d)
    if dgamma is None:
        dgamma = torch.zeros(N, dtype=torch.float32, device=x2d.device)
    if dbeta is None:
        dbeta = torch.zeros(N, dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        check(lib().sp_layernorm_bwd(ptr(gy2d), ptr(x2d), ptr(mean), ptr(rstd), ptr(gamma), ptr(gres2d), M, N,
                                     _lib.dtype_code(x2d), ptr(gx), ptr(dgamma), ptr(dbeta), stream_ptr()),
              "sp_layernorm_bwd")
    return gx, dgamma, dbeta


def gemm(a, b, out=None, out_dtype=None, accumulate=False, split_k=1):
    """a [M, K] times b[N, K]^T -> [M, N] (sp_gemm, f32 accumulation).  a and b are VIEWS: each must be
    contiguous along k or along its row index, so ``gemm(x, w.t())`` is x @ w and ``gemm(x.t(), g.t())``
    is x^T g without any copy.  out_dtype: float32 / bfloat16 / float16 (default: a's dtype).
    accumulate (f32 out): out += product.  split_k = S > 1: returns S partial products [S, M, N]."""
    _need_cuda(a, "a")
    M, K = a.shape
    N = b.shape[0]
    if b.shape[1] != K or a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError(f"gemm: operands [M,K] and [N,K] of one float dtype expected, got {tuple(a.shape)} "
                         f"{a.dtype} and {tuple(b.shape)} {b.dtype}")
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else a.dtype

    def strides(t):      # a dimension of extent 1 may carry any stride
        s0, s1 = t.stride()
        if t.shape[1] == 1 and s0 != 1:
            s1 = 1
        if t.shape[0] == 1 and s1 != 1:
            s0 = 1
        return s0, s1
    sam, sak = strides(a)
    sbn, sbk = strides(b)
    if 1 not in (sam, sak):
        a = a.contiguous()
        sam, sak = a.stride()
    if 1 not in (sbn, sbk):
        b = b.contiguous()
        sbn, sbk = b.stride()
    if out is None:
        shape = (M, N) if split_k == 1 else (split_k, M, N)
        out = torch.empty(shape, dtype=out_dtype, device=a.device)
    elif out.dtype != out_dtype or out.stride(-1) != 1:
        raise ValueError("gemm: out must have out_dtype and unit column stride")
    ldc = out.stride(-2) if M > 1 else max(N, 1)
    css = out.stride(0) if split_k > 1 else 0
    with torch.cuda.device(a.device):
        check(lib().sp_gemm(ptr(a), ptr(b), ptr(out), M, N, K, sam, sak, sbn, sbk, ldc, _lib.dtype_code(a),
                            _lib.dtype_code(out), int(accumulate), split_k, css, stream_ptr()), "sp_gemm")
    return out


def sum_slabs(partials, out=None, accumulate=False, scale=None):
    """partials [S, ...] f32 -> sum over S (times the device scalar `scale`), written or added to `out`."""
    S = partials.shape[0]
    n = partials[0].numel()
    if out is None:
        out = torch.empty(partials.shape[1:], dtype=torch.float32, device=partials.device)
    if scale is not None:
        scale = scale.detach().reshape(-1)[:1].float().contiguous()
    with torch.cuda.device(partials.device):
        check(lib().sp_sum_slabs_f32(ptr(partials), S, n, partials.stride(0), ptr(out), int(accumulate), ptr(scale),
                                     stream_ptr()), "sp_sum_slabs_f32")
    return out


class GradSink:
    """Active inside train.train_step while every parameter's .grad is a pre-allocated, zeroed f32 view
    of the optimizer's flat gradient arena (train.FlatAdam).  The backward kernels that accumulate
    (float atomics, `+=` epilogues) then write straight into p.grad and the autograd functions return
    None for those inputs: no zero-fill of a temporary and no `p.grad += g` kernel per parameter.
    `dp` (train.BucketedAllReduce) is told about every parameter finished this way, since autograd's
    post-accumulate hooks do not fire for it."""

    def __init__(self, dp=None):
        self.dp = dp

    @staticmethod
    def ok(*params):
        return all(p is not None and p.grad is not None and p.grad.dtype == torch.float32 and
                   p.grad.is_contiguous() and p.dtype == torch.float32 for p in params)

    def ready(self, *params):
        if self.dp is not None:
            for p in params:
                self.dp.param_ready(p)


_sink = None


def set_grad_sink(sink):
    global _sink
    _sink = sink


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = x.contiguous().view(-1, x.shape[-1])
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y, mean, rstd = layernorm_fwd(x2, g32, b32, eps)
        ctx.save_for_backward(x2, mean, rstd, g32)
        ctx.pdt = (gamma.dtype, beta.dtype)
        ctx.params = (gamma, beta)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, mean, rstd, g32 = ctx.saved_tensors
        gyv = gy.contiguous().view(x2.shape).to(x2.dtype)
        sink = _sink
        if sink is not None and sink.ok(*ctx.params):
            gamma, beta = ctx.params
            gx, _, _ = layernorm_bwd(gyv, x2, mean, rstd, g32, None, gamma.grad, beta.grad)
            sink.ready(gamma, beta)
            return gx.view(gy.shape), None, None, None
        gx, dg, db = layernorm_bwd(gyv, x2, mean, rstd, g32)
        return gx.view(gy.shape), dg.to(ctx.pdt[0]), db.to(ctx.pdt[1]), None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNormFn.apply(x, gamma, beta, eps)


def sru_pack_w(weight, d, k, C, dtype):
    """state_dict weight f32 [n_in, 2*d*k] (dir, d, k) -> Wk [n_in, k*2*d] in dtype, rows f*C + c."""
    n_in = weight.shape[0]
    out = torch.empty((n_in, k * 2 * d), dtype=dtype, device=weight.device)
    with torch.cuda.device(weight.device):
        check(lib().sp_sru_pack_w(ptr(weight), n_in, d, k, C, _lib.dtype_code(out), ptr(out), stream_ptr()),
              "sp_sru_pack_w")
    return out


def sru_pack_wt(weight, d, k, C, dtype, pad=0, both=False):
    """state_dict weight f32 [n_in, 2*d*k] -> WkT [k*2*d, n_in] in dtype (columns f*C + c): sru_pack_w's
    result transposed, the K-contiguous B operand of the projection GEMM; pad: extra elements per row of the buffer.
    both=True: returns (WkT, Wk), the second being sru_pack_w's layout written in the same pass."""
    n_in = weight.shape[0]
    out = torch.empty((k * 2 * d, n_in + pad), dtype=dtype, device=weight.device)[:, :n_in]
    out_w = torch.empty((n_in, k * 2 * d), dtype=dtype, device=weight.device) if both else None
    with torch.cuda.device(weight.device):
        check(lib().sp_sru_pack_wt(ptr(weight), n_in, d, k, C, _lib.dtype_code(out), ptr(out), out.stride(0),
                                   ptr(out_w), stream_ptr()), "sp_sru_pack_wt")
    return (out, out_w) if both else out


def sru_pack_wt_many(weights, d, k, C, dtype, pad=0, both=False):
    """sru_pack_wt for several layers of one shape in ONE launch -> list of (WkT, Wk or None)."""
    L = len(weights)
    n_in, dev = weights[0].shape[0], weights[0].device
    res = []
    for lo in range(0, L, 16):
        ws = [w.detach().float().contiguous() for w in weights[lo:lo + 16]]
        n = len(ws)
        out = torch.empty((n, k * 2 * d, n_in + pad), dtype=dtype, device=dev)
        out_w = torch.empty((n, n_in, k * 2 * d), dtype=dtype, device=dev) if both else None
        arr = ctypes.c_void_p * n
        pw = arr(*[w.data_ptr() for w in ws])
        pt = arr(*[out[i].data_ptr() for i in range(n)])
        po = arr(*[out_w[i].data_ptr() for i in range(n)]) if both else None
        with torch.cuda.device(dev):
            check(lib().sp_sru_pack_wt_batched(pw, n, n_in, d, k, C, _lib.dtype_code(out), pt, out.stride(1), po,
                                               stream_ptr()), "sp_sru_pack_wt_batched")
        res += [(out[i][:, :n_in], out_w[i] if both else None) for i in range(n)]
    return res


def sru_prepack(cells, T, B, dtype, train):
    """The packed projection weights of every SRU layer that runs the NT projection, one launch per group of
    same-shaped layers instead of one per layer.  cells: [(weight, d, k, C)] -> {index: (WkT, Wk or None)}."""
    groups = {}
    for i, (w, d, k, C) in enumerate(cells):
        n_in = w.shape[0]
        if _nt_projection(T * B, n_in, 2 * d * k, d, dtype):
            groups.setdefault((n_in, d, k, C), []).append(i)
    out = {}
    for (n_in, d, k, C), idx in groups.items():
        pad = 64 if (n_in * torch.empty((), dtype=dtype).element_size()) % 1024 == 0 else 0
        packs = sru_pack_wt_many([cells[i][0] for i in idx], d, k, C, dtype, pad, both=train)
        out.update(zip(idx, packs))
    return out


def sru_unpack_dw(partials, n_in, d, k, C, out=None):
    """split-K partials [S, n_in, k*2*d] -> f32 dweight [n_in, 2*d*k] in the state_dict layout; with
    `out` (the parameter's gradient buffer) the sum is ADDED to it."""
    S = partials.shape[0]
    acc = out is not None
    if out is None:
        out = torch.empty((n_in, 2 * d * k), dtype=torch.float32, device=partials.device)
    with torch.cuda.device(partials.device):
        check(lib().sp_sru_unpack_dw(ptr(partials), S, n_in, d, k, C, _lib.dtype_code(partials), int(acc),
                                     ptr(out), stream_ptr()), "sp_sru_unpack_dw")
    return out


def _splitk_dw(xn, gU2):
    """f32 partials [S, n_in, N] of xn^T gU: the output has only n_in/256 * N/256 = 48 tiles for 256
    CUs at the reference's sizes, so K (= T*B) is split S ways into slabs (no atomics; the slabs are
    summed by sp_sru_unpack_dw / sp_sum_slabs_f32).  S is the library's choice for the shape."""
    S = lib().sp_gemm_tn_splits(xn.shape[1], gU2.shape[1], xn.shape[0], _lib.dtype_code(xn))
    out = gemm(xn.t(), gU2.t(), out_dtype=torch.float32, split_k=S)
    return out if S > 1 else out.unsqueeze(0)


_VPAD = 256     # the fast GEMM kernels work on whole 256-column tiles


def _linear_pack(weight, Vp, dtype, scale=None, transposed=False):
    """sp_linear_pack: the f32 weight [V, n] as a 16-bit [Vp, n] (rows V.. zero) or, transposed=True, as
    scale * W^T [n, Vp] -- one launch either way."""
    V, n = weight.shape
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    out = torch.empty((n, Vp) if transposed else (Vp, n), dtype=dtype, device=w.device)
    if scale is not None:
        scale = scale.detach().reshape(-1)[:1].float().contiguous()
    with torch.cuda.device(w.device):
        check(lib().sp_linear_pack(ptr(w), V, n, Vp, ptr(scale), _lib.dtype_code(out), None if transposed else ptr(out),
                                   ptr(out) if transposed else None, stream_ptr()), "sp_linear_pack")
    return out


def _classifier_pack(weight, dtype, M):
    """The classifier weight [V, n] in the activation dtype, rows zero-padded to whole 256-column tiles when
    the problem is big enough for the persistent kernels (logits and their gradient are then [M, Vp] buffers of
    which the CTC kernels see the first V columns)."""
    V, n = weight.shape
    if dtype not in (torch.bfloat16, torch.float16) or M < 4096 or n % 256 or V % _VPAD == 0:
        return weight.detach().to(dtype), V
    Vp = (V + _VPAD - 1) // _VPAD * _VPAD
    return _linear_pack(weight, Vp, dtype), Vp


def _classifier_logits(x2d, weight):
    """(f32 logits [M, V] straight from the GEMM's f32 accumulators -- a view of an [M, Vp] buffer when the weight
    was padded --, the (padded) weight in the activation dtype)."""
    wb, Vp = _classifier_pack(weight, x2d.dtype, x2d.shape[0])
    return gemm(x2d, wb, out_dtype=torch.float32)[:, :weight.shape[0]], wb


def _classifier_dx(gb, wb, scale=None, weight=None):
    """gb [M, Vp] (pad columns zero) times wb [Vp, n]; with a padded weight as an NT product on scale * W^T, which
    sp_linear_pack takes straight from the f32 parameter (`weight`, unchanged since the forward)."""
    fast = wb.shape[0] % _VPAD == 0 and gb.shape[0] >= 4096 and wb.dtype in (torch.bfloat16, torch.float16)
    if fast and weight is not None:
        return gemm(gb, _linear_pack(weight, wb.shape[0], wb.dtype, scale, transposed=True))
    w = wb if scale is None else (wb.float() * scale).to(wb.dtype)
    if fast:
        return gemm(gb, w.t().contiguous())
    return gemm(gb, w.t())


def _classifier_dw(gb, x2d, weight, scale=None):
    """Weight gradient g^T x of the classifier -- a [V, n] output with K = T'*B, split over the rows into
    slabs that are summed (times the incoming loss gradient) straight into the parameter's gradient
    buffer when the step owns one (GradSink); returns None then, else the gradient tensor.  gb may carry
    zero pad columns (rows of the product beyond V, which are dropped)."""
    V = weight.shape[0]
    P = _splitk_dw(gb, x2d)[:, :V]
    sink = _sink
    vec = P[0].numel() % 4 == 0
    if vec and sink is not None and sink.ok(weight):
        sum_slabs(P, weight.grad, True, scale)
        sink.ready(weight)
        return None
    if vec:
        return sum_slabs(P, None, False, scale).to(weight.dtype)
    gw = P.sum(0)
    return (gw * scale if scale is not None else gw).to(weight.dtype)


class _ClassifierFn(torch.autograd.Function):
    """logits = x W^T (net.py:73-76, Linear without bias): the logits leave the GEMM in f32 (the CTC
    kernels read f32; no bf16 round trip and no conversion pass over [B*T', V]), and the weight gradient
    g^T x is a split-K GEMM."""

    @staticmethod
    def forward(ctx, x2d, weight):
        out, wb = _classifier_logits(x2d, weight)
        ctx.save_for_backward(x2d, wb, weight)
        ctx.weight = weight
        return out

    @staticmethod
    def backward(ctx, g):
        x2d, wb, w32 = ctx.saved_tensors
        V, Vp = ctx.weight.shape[0], wb.shape[0]
        if Vp != V:         # pad columns zero: they meet the padded weight's zero rows
            gb = torch.empty((g.shape[0], Vp), dtype=x2d.dtype, device=g.device)
            gb[:, V:].zero_()
            gb[:, :V].copy_(g)
        else:
            gb = g.to(x2d.dtype).contiguous()
        return _classifier_dx(gb, wb, None, w32), _classifier_dw(gb, x2d, ctx.weight)


_len_cache = {}
_gbuf_pool = {}      # key -> [buffer, in_use]


def _full_lengths(B, T, device):
    """int32 [B] filled with T, kept per shape (one fill kernel per step otherwise)."""
    key = (B, T, str(device))
    t = _len_cache.get(key)
    if t is None:
        if len(_len_cache) > 64:
            _len_cache.clear()
        t = _len_cache[key] = torch.full((B,), T, dtype=torch.int32, device=device)
    return t


def _grad_buffer(M, V, Vp, dtype, device):
    """[M, Vp] buffer for the CTC gradient whose pad columns V.. are zero.  The CTC kernel writes columns < V
    only, so a buffer that went through one step keeps its zero pad: one buffer per shape is handed out again
    once its backward has released it (a second forward before that gets a fresh one)."""
    if Vp == V:
        return torch.empty((M, Vp), dtype=dtype, device=device), None
    key = (M, V, Vp, dtype, str(device))
    ent = _gbuf_pool.get(key)
    if ent is not None and not ent[1]:
        ent[1] = True
        return ent[0], key
    buf = torch.empty((M, Vp), dtype=dtype, device=device)
    buf[:, V:].zero_()
    if len(_gbuf_pool) > 8:
        _gbuf_pool.clear()
    _gbuf_pool[key] = [buf, True]
    return buf, key


def _release_grad_buffer(key, buf):
    ent = _gbuf_pool.get(key) if key is not None else None
    if ent is not None and ent[0].data_ptr() == buf.data_ptr():
        ent[1] = False


class _HeadCTCFn(torch.autograd.Function):
    """Classifier GEMM + CTC as ONE differentiable node for the training step (train.py:69-77 on
    net.py:73-76's output): the logits stay time-major as the SRU stack leaves them (the CTC kernels take
    strides, so the [T',B] -> [B,T'] copy and its backward disappear), the CTC gradient is taken once and
    kept in the activation dtype, and the incoming scalar gradient multiplies the SMALL operands (W and
    dW) instead of a pass over [T'*B, V]."""

    @staticmethod
    def forward(ctx, x_tm, weight, labels, label_lengths, blank, max_label_len):
        T, B, n = x_tm.shape
        V = weight.shape[0]
        x2d = x_tm.reshape(T * B, n)
        logits, wb = _classifier_logits(x2d, weight)            # [T*B, V] view, row pitch Vp
        Vp = wb.shape[0]
        ctx.weight = weight
        il = _full_lengths(B, T, x_tm.device)                                  # train.py:72-74: all T' frames
        gbuf, ctx.gkey = _grad_buffer(T * B, V, Vp, x2d.dtype, x2d.device)     # the gradient leaves the kernel in
                                                                               # the GEMMs' dtype, pad columns zero
        loss, _, _ = ctc_fwd_bwd(logits.as_strided((T, B, V), (B * Vp, Vp, 1)), labels, il, label_lengths, blank,
                                 max_label_len, True, 1.0, True, x2d.dtype,
                                 grad_out=gbuf.as_strided((T, B, V), (B * Vp, Vp, 1)))
        ctx.save_for_backward(x2d, wb, gbuf, weight)
        ctx.shape = (T, B, n)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x2d, wb, gb, w32 = ctx.saved_tensors
        gx = _classifier_dx(gb, wb, g, w32).view(ctx.shape)
        gw = _classifier_dw(gb, x2d, ctx.weight, g)
        _release_grad_buffer(ctx.gkey, gb)
        return gx, gw, None, None, None, None


def classifier_ctc_loss(x_tm, weight, labels, label_lengths, blank=0, max_label_len=None):
    """x_tm [T', B, n] (time-major, already layer-normed), weight [V, n] -> scalar CTC loss (mean
    reduction over the batch of nll / label length, every utterance scored over all T' frames)."""
    return _HeadCTCFn.apply(x_tm, weight, labels, label_lengths, blank, max_label_len)


def classifier(x, weight):
    """x [..., n] (compute dtype) , weight [V, n] -> f32 logits [..., V]."""
    lead = x.shape[:-1]
    return _ClassifierFn.apply(x.reshape(-1, x.shape[-1]), weight).view(*lead, weight.shape[0])


def _nt_projection(M, n_in, N, d, dtype):
    """The big forward projections run as an NT GEMM on W^T with padded row pitches (sp_sru_pack_wt):
    bf16, output-heavy (N >= 2 n_in), enough rows for one 192-row tile per CU and column.  That layout
    is worth 15 % on this GEMM (profiles/r01_gemm.md): K-contiguous rows of both operands are what the
    LDS-DMA kernels stream, and the pitch padding keeps 256 CUs off a 2 KB row stride."""
    return (dtype in (torch.bfloat16, torch.float16) and N % 256 == 0 and n_in % 64 == 0 and d % 64 == 0 and N >= 2 * n_in
            and (M // 192) * (N // 256) >= 256 and M * max(n_in, N) < (1 << 30))


class _SRULayerFn(torch.autograd.Function):
    """One SRU layer (net.py:61-72): LayerNorm -> projection GEMM -> recurrence, all HIP kernels, with
    a hand-ordered backward: recurrence backward, the two GEMMs (dW split-K), LayerNorm backward
    fused with the highway's residual gradient.  x [T,B,n_in] with feature order f*C + c; ln_w, ln_b,
    weight are the state_dict parameters (f32, reference order); packing to the kernels' layout and the
    inverse for the gradients are HIP kernels (sp_sru_pack_w / sp_sru_unpack_dw)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, weight, wc, bias, mask, d, k, C, eps, pre=None):
        T, B, n_in = x.shape
        x2 = x.contiguous().view(T * B, n_in)
        g32, b32 = ln_w.detach().float(), ln_b.detach().float()
        if C > 1:
            g32, b32 = g32.view(C, n_in // C).t().reshape(-1), b32.view(C, n_in // C).t().reshape(-1)
        g32, b32 = g32.contiguous(), b32.contiguous()
        own = _nt_projection(T * B, n_in, 2 * d * k, d, x.dtype)
        # rows whose pitch is a multiple of 1 KB land on few L2 channels when 256 CUs walk them in step
        pad = 64 if own and (n_in * x.element_size()) % 1024 == 0 else 0
        xn, mean, rstd = layernorm_fwd(x2, g32, b32, eps, pad)
        w32 = weight.detach().float().contiguous()
        train = any(ctx.needs_input_grad)      # False under no_grad: nothing is kept for a backward
        if own and pre is not None and (pre[1] is not None or not train):   # packed by sru_prepack with its siblings
            wkt, wk = pre
            U = gemm(xn, wkt)
            del wkt
        elif own and train:    # W^T [N][n_in] for the forward (NT), W [n_in][N] for the data gradient
            wkt, wk = sru_pack_wt(w32, d, k, C, x.dtype, pad, both=True)
            U = gemm(xn, wkt)
            del wkt
        elif own:
            wk = None
            U = gemm(xn, sru_pack_wt(w32, d, k, C, x.dtype, pad))
        else:        # wk holds W [n_in][N]
            wk = sru_pack_w(w32, d, k, C, x.dtype)
            U = gemm(xn, wk.t())
        wc32, bs32 = wc.detach().float().contiguous(), bias.detach().float().contiguous()
        xr = x2 if k == 3 else None
        h, c = sru_fwd(U, xr, wc32, bs32, mask, T, B, d, k, keep_c=train)
        if train:
            ctx.save_for_backward(x2, xn, mean, rstd, g32, wk, U, wc32, bs32, mask, c)
        ctx.dims = (T, B, d, k, n_in, C)
        ctx.pdt = (ln_w.dtype, ln_b.dtype, weight.dtype, wc.dtype, bias.dtype)
        ctx.params = (ln_w, ln_b, weight, wc, bias)
        return h

    @staticmethod
    def backward(ctx, gh):
        x2, xn, mean, rstd, g32, wk, U, wc32, bs32, mask, c = ctx.saved_tensors
        T, B, d, k, n_in, C = ctx.dims
        gh = gh.contiguous().to(x2.dtype)
        ln_w, ln_b, weight, wc, bias = ctx.params
        sink = _sink
        direct = sink is not None and sink.ok(ln_w, ln_b, weight, wc, bias)
        gU, gxres, gwc, gbias = sru_bwd(U, x2 if k == 3 else None, wc32, bs32, mask, c, gh, T, B, d, k,
                                        wc.grad.view(-1) if direct else None, bias.grad.view(-1) if direct else None)
        gU2 = gU.view(T * B, -1)
        M = T * B
        gw = sru_unpack_dw(_splitk_dw(xn, gU2), n_in, d, k, C, weight.grad if direct else None)
        gxn = gemm(gU2, wk)
        gres = gxres.view(T * B, n_in) if k == 3 else None
        ln_direct = direct and C == 1   # layer 0's vectors are stored in the reference's c*F + f order
        gx, dg, db = layernorm_bwd(gxn, x2, mean, rstd, g32, gres, ln_w.grad if ln_direct else None,
                                   ln_b.grad if ln_direct else None)
        if C > 1:
            dg, db = dg.view(n_in // C, C).t().reshape(-1), db.view(n_in // C, C).t().reshape(-1)
        p = ctx.pdt
        if direct:
            sink.ready(weight, wc, bias)
            if ln_direct:
                sink.ready(ln_w, ln_b)
                return (gx.view(T, B, n_in), None, None, None, None, None, None, None, None, None, None, None)
            return (gx.view(T, B, n_in), dg.to(p[0]), db.to(p[1]), None, None, None, None, None, None, None, None, None)
        return (gx.view(T, B, n_in), dg.to(p[0]), db.to(p[1]), gw.to(p[2]), gwc.to(p[3]), gbias.to(p[4]),
                None, None, None, None, None, None)


class _MatmulFn(torch.autograd.Function):
    """y = a @ w for activations a [M, K] and an f32 parameter(-derived) matrix w [K, N]: the product runs in a's
    dtype on sp_gemm (f32 accumulation), the weight gradient a^T g comes back in f32."""

    @staticmethod
    def forward(ctx, a, w):
        wb = w.detach().to(a.dtype).contiguous()
        ctx.save_for_backward(a, wb)
        ctx.wdtype = w.dtype
        return gemm(a, wb.t())

    @staticmethod
    def backward(ctx, g):
        a, wb = ctx.saved_tensors
        g = g.contiguous()
        ga = gemm(g, wb) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = _splitk_dw(a, g).sum(0).to(ctx.wdtype)
        return ga, gw


def matmul(a, w):
    """a [M, K] (activation dtype) @ w [K, N] (parameter, any float dtype) -> [M, N] in a's dtype, differentiable."""
    return _MatmulFn.apply(a, w)


def sru_layer(x, ln_w, ln_b, weight, wc, bias, mask, d, k, C=1, eps=1e-5, pre=None):
    """pre: this layer's entry of sru_prepack (its weights already packed), or None."""
    return _SRULayerFn.apply(x, ln_w, ln_b, weight, wc, bias, mask, d, k, C, eps, pre)
