"""Oracle for log_softmax + CTC loss/gradient (reference train.py:69-77, :114-122, :266).

Test infrastructure only.  PINNED: the reference's CTC *is* ``torch.nn.CTCLoss(blank=0)`` and
torch's CPU implementation is installed here, so ``ctc_torch`` runs the reference arithmetic
itself.  ``ctc_numpy`` is an independent float64 alpha/beta restatement (log space), checked
against ``ctc_torch`` in tests/test_oracle.py; it also documents the gradient the HIP kernel
must produce.
"""
import numpy as np
import torch
import torch.nn.functional as F


def ctc_torch(logits_tbv, labels, input_lengths, label_lengths, blank=0, dtype=torch.float32):
    """Exactly the reference call chain (train.py:69-77) on [T, B, V] *logits*:
    log_softmax(dim=2) -> CTCLoss(blank=0, reduction='mean', zero_infinity=False).

    Returns (loss scalar, nll[B], grad wrt logits [T, B, V]).
    """
    z = torch.as_tensor(logits_tbv).detach().to(dtype).clone().requires_grad_(True)
    lp = F.log_softmax(z, dim=2)
    labels = torch.as_tensor(labels).long()
    il = torch.as_tensor(input_lengths).to(torch.int32)
    ll = torch.as_tensor(label_lengths).to(torch.int32)
    loss = F.ctc_loss(lp, labels, il, ll, blank=blank, reduction="mean", zero_infinity=False)
    nll = F.ctc_loss(lp.detach(), labels, il, ll, blank=blank, reduction="none")
    (grad,) = torch.autograd.grad(loss, z)
    return loss.detach(), nll, grad


def _logsumexp2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), -np.inf, r)


def ctc_numpy(logits_tbv, labels, input_lengths, label_lengths, blank=0):
    """Float64 alpha/beta recursion.  Returns (loss, nll[B], grad wrt logits [T,B,V]).

    nll_b = -logsumexp(alpha[T_b-1, S-1], alpha[T_b-1, S-2]);
    loss = mean_b nll_b / max(L_b, 1);
    grad[t,b,c] = (softmax[t,b,c] - exp(logsum_{s: l'_s = c}(alpha_t(s)+beta_t(s)) + nll_b - lp[t,b,c]))
                  / (B * max(L_b,1)) for t < T_b, 0 for t >= T_b.        
    """
    z = np.asarray(logits_tbv, dtype=np.float64)
    T, B, V = z.shape
    zmax = z.max(axis=2, keepdims=True)
    lp = z - zmax - np.log(np.exp(z - zmax).sum(axis=2, keepdims=True))
    labels = np.asarray(labels)
    nll = np.zeros(B)
    grad = np.zeros_like(z)
    for b in range(B):
        Tb, L = int(input_lengths[b]), int(label_lengths[b])
        S = 2 * L + 1
        ext = np.full(S, blank, dtype=np.int64)
        ext[1::2] = labels[b, :L]
        can_skip = np.zeros(S, dtype=bool)
        can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        alpha = np.full((Tb, S), -np.inf)
        beta = np.full((Tb, S), -np.inf)
        if Tb == 0:
            nll[b] = 0.0 if L == 0 else np.inf
            continue
        alpha[0, 0] = lp[0, b, blank]
        if S > 1:
            alpha[0, 1] = lp[0, b, ext[1]]
        for t in range(1, Tb):
            a = alpha[t - 1]
            acc = a.copy()
            acc[1:] = _logsumexp2(acc[1:], a[:-1])
            sk = np.full(S, -np.inf)
            sk[2:] = np.where(can_skip[2:], a[:-2], -np.inf)
            acc = _logsumexp2(acc, sk)
            alpha[t] = acc + lp[t, b, ext]
        end = alpha[Tb - 1, S - 1]
        if S > 1:
            end = _logsumexp2(end, alpha[Tb - 1, S - 2])
        nll[b] = -end
        beta[Tb - 1, S - 1] = lp[Tb - 1, b, blank]
        if S > 1:
            beta[Tb - 1, S - 2] = lp[Tb - 1, b, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            bt = beta[t + 1]
            acc = bt.copy()
            acc[:-1] = _logsumexp2(acc[:-1], bt[1:])
            sk = np.full(S, -np.inf)
            sk[:-2] = np.where(can_skip[2:], bt[2:], -np.inf)
            acc = _logsumexp2(acc, sk)
            beta[t] = acc + lp[t, b, ext]
        scale = 1.0 / (B * max(L, 1))
        ab = alpha + beta  # [Tb, S]
        for t in range(Tb):
            post = np.full(V, -np.inf)
            for s in range(S):
                post[ext[s]] = _logsumexp2(post[ext[s]], ab[t, s])
            with np.errstate(over="ignore", invalid="ignore"):
                occ = np.exp(post + nll[b] - lp[t, b])
            grad[t, b] = (np.exp(lp[t, b]) - occ) * scale
    denom = np.maximum(np.asarray(label_lengths, dtype=np.float64), 1.0)
    loss = float(np.mean(nll / denom))
    return loss, nll, grad
