"""Oracle restatement of the bidirectional SRU stack (reference net.py:61-72).

Test infrastructure only.  PARITY UNPINNED: the ``sru`` package (asappresearch/sru, unpinned
in requirements.txt) is neither installed nor vendored.  This follows the SRU paper (Lei et
al. 2018, eqs. 1-4 with the v2 "peephole" form) and the package's documented layout as
recorded in SURVEY.md section 8 row N4, for the constructor arguments the reference uses:
``layer_norm=True, bidirectional=True, rescale=False, highway_bias=0.0, use_tanh=False,
has_skip_term=True, projection_size=0`` and variational dropout on the cell state inside the
highway term (one mask per (batch, channel), constant over time; 0 on the last layer).

Parameter layout per layer (state_dict keys ``sru_layers.rnn_lst.{i}.*``):
  weight      [n_in, 2*d*k]   columns ordered (dir, j, k) with k innermost; k = 4 when
                              n_in != 2*d (extra highway projection) else 3
  weight_c    [2 * 2*d]       (v_f | v_r), each ordered (dir, j)
  bias        [2 * 2*d]       (b_f | b_r)
  layer_norm.{weight,bias} [n_in]   applied to the cell input before the projection

Per column (dir, j), c_{-1} = 0, forward direction t = 0..T-1, backward direction t = T-1..0:
  f_t = sigmoid(U1_t + v_f * c_{t-1} + b_f)
  r_t = sigmoid(U2_t + v_r * c_{t-1} + b_r)
  c_t = U0_t + (c_{t-1} - U0_t) * f_t
  h_t = (c_t * mask - x'_t) * r_t + x'_t        x' = U3_t (k = 4) or the un-normalised input (k = 3)
"""
import torch
import torch.nn.functional as F


def sru_layer(x, weight, weight_c, bias, ln_w, ln_b, d, mask_c=None):
    """x [T, B, n_in] -> (h [T, B, 2d], c_last [B, 2d]).  Differentiable (torch autograd)."""
    T, B, n_in = x.shape
    D2 = 2 * d
    k = weight.shape[1] // D2
    xn = F.layer_norm(x, (n_in,), ln_w, ln_b)
    U = (xn.reshape(T * B, n_in) @ weight).view(T, B, 2, d, k)
    vf, vr = weight_c[:D2].view(2, d), weight_c[D2:].view(2, d)
    bf, br = bias[:D2].view(2, d), bias[D2:].view(2, d)
    xres = x.view(T, B, 2, d) if k == 3 else U[..., 3]
    mask = None if mask_c is None else mask_c.view(B, 2, d)
    hs = [[None] * T, [None] * T]
    c_last = []
    for direction in range(2):
        c = x.new_zeros(B, d)
        order = range(T) if direction == 0 else range(T - 1, -1, -1)
        for t in order:
            u0, u1, u2 = U[t, :, direction, :, 0], U[t, :, direction, :, 1], U[t, :, direction, :, 2]
            f = torch.sigmoid(u1 + vf[direction] * c + bf[direction])
            r = torch.sigmoid(u2 + vr[direction] * c + br[direction])
            c = u0 + (c - u0) * f
            g = c if mask is None else c * mask[:, direction]
            xp = xres[t, :, direction]
            hs[direction][t] = (g - xp) * r + xp
        c_last.append(c)
    h = torch.stack([torch.cat([hs[0][t], hs[1][t]], dim=-1) for t in range(T)])
    return h, torch.cat(c_last, dim=-1)


def sru_stack(x, layers, d, masks=None):
    """layers: list of dicts with keys weight, weight_c, bias, ln_w, ln_b."""
    for i, p in enumerate(layers):
        m = None if masks is None else masks[i]
        x, _ = sru_layer(x, p["weight"], p["weight_c"], p["bias"], p["ln_w"], p["ln_b"], d, m)
    return x
