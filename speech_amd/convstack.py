"""Conv / BatchNorm / ReLU residual stack of SRModel (net.py:56-59 and :6-42 of the reference) on
the hand-written channel-last bf16 kernels of csrc/conv.hip, forward and backward.

The stack takes the f32 spectrogram [B,1,n_feats,T] and returns the SRU input directly in
[T', B, F*32] layout with feature index f*32 + c (the reference's reshape at net.py:81-83 gives
c*F + f; SRModel permutes the first SRU layer's weight rows and layer-norm vectors to match, the
state_dict keeps the reference order).  BatchNorm is never materialised: the consumer convolution
applies scale/shift(+ReLU) while it stages its input, and the backward convolutions apply the
BatchNorm-backward affine map the same way.
"""
import torch

from ._lib import BN_SUMS_WORDS, CONV_WGRAD_WS_WORDS, check, lib, ptr, stream_ptr

IN_ID, IN_BNRELU, IN_BNBWD = 0, 1, 2
EPI_NONE, EPI_MASK, EPI_ADD = 0, 1, 2
C = 32
PACK_ELEMS = 15 * 2 * 64 * 8


def _f32(t):
    return t.detach().float().contiguous()


def _pack(w, bwd_data):
    out = torch.empty(PACK_ELEMS, dtype=torch.bfloat16, device=w.device)
    check(lib().sp_conv_pack_w(ptr(w), int(bwd_data), ptr(out), stream_ptr()), "sp_conv_pack_w")
    return out


def _pack_both(w):
    """(forward image, backward-data image) of one convolution weight, one launch."""
    out = torch.empty(2 * PACK_ELEMS, dtype=torch.bfloat16, device=w.device)
    check(lib().sp_conv_pack_w(ptr(w), 2, ptr(out), stream_ptr()), "sp_conv_pack_w")
    return out[:PACK_ELEMS], out[PACK_ELEMS:]


def _pack_many(ws, bwd_data):
    """Every weight of `ws` packed in one launch -> list of (forward image, backward-data image or None)."""
    import ctypes
    n = len(ws)
    per = (2 if bwd_data == 2 else 1) * PACK_ELEMS
    out = torch.empty((n, per), dtype=torch.bfloat16, device=ws[0].device)
    arr = ctypes.c_void_p * n
    check(lib().sp_conv_pack_w_batched(arr(*[w.data_ptr() for w in ws]), n, int(bwd_data),
                                       arr(*[out[i].data_ptr() for i in range(n)]), stream_ptr()), "sp_conv_pack_w_batched")
    if bwd_data == 2:
        return [(out[i, :PACK_ELEMS], out[i, PACK_ELEMS:]) for i in range(n)]
    return [(out[i], None) for i in range(n)]


_wgrad_ws = {}


def _wgrad_workspace(dev):
    """sp_conv3x5_wgrad's scratch (one slab per persistent workgroup; nothing in it needs to be zero)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.empty(CONV_WGRAD_WS_WORDS, dtype=torch.float32, device=dev)
    return _wgrad_ws[key]


def _conv(inp, in2, coef, in_mode, wpack, bias, epi_src, epi_coef, epi_mode, B, F, T, bn_sums=None):
    """bn_sums: optional f32 [BN_SUMS_WORDS] (result = the sum of its first eight rows of 64) receiving the BatchNorm reduction over the output (fused epilogue)."""
    out = torch.empty((B, F, T, C), dtype=torch.bfloat16, device=inp.device)
    check(lib().sp_conv3x5(ptr(inp), ptr(in2), ptr(coef), in_mode, ptr(wpack), ptr(bias), ptr(epi_src),
                           ptr(epi_coef), epi_mode, B, F, T, ptr(out), ptr(bn_sums), stream_ptr()), "sp_conv3x5")
    return out


def _bn_fwd(sums, npos, bn_w, bn_b, rmean, rvar, momentum, eps, training):
    """sums: the (sum y, sum y^2) the producing convolution's epilogue left (None in eval mode)."""
    dev = bn_w.device
    coef = torch.empty(64, dtype=torch.float32, device=dev)
    save = torch.empty(64, dtype=torch.float32, device=dev)
    check(lib().sp_bn_finalize_fwd(ptr(sums), npos, ptr(bn_w), ptr(bn_b), ptr(rmean), ptr(rvar), momentum,
                                   eps, int(training), ptr(coef), ptr(save), stream_ptr()), "sp_bn_finalize_fwd")
    return coef, save


class _ConvStackFn(torch.autograd.Function):
    """args: spec, training, momentum, eps, stem_w, stem_b, then per block
    (conv1.w, conv1.b, bn1.w, bn1.b, conv2.w, conv2.b, bn2.w, bn2.b), then per block the four
    running-stat buffers (bn1.mean, bn1.var, bn2.mean, bn2.var)."""

    @staticmethod
    def forward(ctx, spec, training, momentum, eps, stem_w, stem_b, *rest):
        nblk = len(rest) // 12
        params, bufs = rest[: 8 * nblk], rest[8 * nblk:]
        spec = spec.detach().float().contiguous()
        B, _, Fin, Tin = spec.shape
        F, T = (Fin - 1) // 2 + 1, (Tin - 1) // 2 + 1
        npos = B * F * T
        dev = spec.device
        with torch.cuda.device(dev):
            s = stream_ptr()
            sw, sb = _f32(stem_w), _f32(stem_b)
            x = torch.empty((B, F, T, C), dtype=torch.bfloat16, device=dev)
            check(lib().sp_conv_stem_fwd(ptr(spec), ptr(sw), ptr(sb), B, Fin, Tin, ptr(x), s), "sp_conv_stem_fwd")
            saved = []
            # all convolution weights packed in one launch (forward and, in training, backward-data images)
            cws = [_f32(params[8 * i + j]) for i in range(nblk) for j in (0, 4)]
            packs = _pack_many(cws, 2 if training else 0)
            for i in range(nblk):
                c1w, c1b, b1w, b1b, c2w, c2b, b2w, b2b = [_f32(p) for p in params[8 * i: 8 * i + 8]]
                rm1, rv1, rm2, rv2 = bufs[4 * i: 4 * i + 4]
                s1 = torch.empty(BN_SUMS_WORDS, dtype=torch.float32, device=dev) if training else None
                s2 = torch.empty(BN_SUMS_WORDS, dtype=torch.float32, device=dev) if training else None
                (p1f, p1b), (p2f, p2b) = packs[2 * i], packs[2 * i + 1]
                y1 = _conv(x, None, None, IN_ID, p1f, c1b, None, None, EPI_NONE, B, F, T, s1)
                coef1, save1 = _bn_fwd(s1, npos, b1w, b1b, rm1, rv1, momentum, eps, training)
                y2 = _conv(y1, None, coef1, IN_BNRELU, p2f, c2b, None, None, EPI_NONE, B, F, T, s2)
                coef2, save2 = _bn_fwd(s2, npos, b2w, b2b, rm2, rv2, momentum, eps, training)
                last = i == nblk - 1
                if last:   # [T][B][F*32]
                    z = torch.empty((T, B, F * C), dtype=torch.bfloat16, device=dev)
                    zs = (F * C, C, B * F * C)
                else:      # [B][F][T][32]
                    z = torch.empty((B, F, T, C), dtype=torch.bfloat16, device=dev)
                    zs = (F * T * C, T * C, C)
                check(lib().sp_block_out(ptr(y2), ptr(coef2), ptr(x), B, F, T, ptr(z), zs[0], zs[1], zs[2], s),
                      "sp_block_out")
                saved.append((x, y1, y2, z, zs, coef1, save1, save2, c1w, c2w, b1w, b2w, p1b, p2b))
                x = z
        ctx.saved = saved
        ctx.params = (stem_w, stem_b) + tuple(params)
        ctx.spec = spec
        ctx.dims = (B, Fin, Tin, F, T, nblk)
        ctx.training = training
        return x

    @staticmethod
    def backward(ctx, gout):
        if not ctx.training:
            raise RuntimeError("conv stack backward needs training mode (batch-statistics BatchNorm)")
        B, Fin, Tin, F, T, nblk = ctx.dims
        npos = B * F * T
        dev = gout.device
        gz = gout.contiguous()
        if gz.dtype != torch.bfloat16:
            gz = gz.to(torch.bfloat16)
        grads = []
        from . import ops
        sink = ops._sink
        P = ctx.params
        direct = sink is not None and sink.ok(*P)   # accumulate straight into the parameters' .grad
        with torch.cuda.device(dev):
            s = stream_ptr()
            ws = _wgrad_workspace(dev)
            for i in range(nblk - 1, -1, -1):
                x, y1, y2, z, zs, coef1, save1, save2, c1w, c2w, b1w, b2w, p1b, p2b = ctx.saved[i]
                if direct:
                    pc1w, pc1b, pb1w, pb1b, pc2w, pc2b, pb2w, pb2b = P[2 + 8 * i: 10 + 8 * i]
                    gw1, gg1, gb1, gw2, gg2, gb2 = pc1w.grad, pb1w.grad, pb1b.grad, pc2w.grad, pb2w.grad, pb2b.grad
                else:
                    small = torch.zeros(6, 32, dtype=torch.float32, device=dev)
                    gg1, gb1, gg2, gb2, zb1, zb2 = small.unbind(0)
                    gw1, gw2 = torch.zeros_like(c1w), torch.zeros_like(c2w)
                sums = torch.empty(BN_SUMS_WORDS, dtype=torch.float32, device=dev)
                c3_2 = torch.empty(96, dtype=torch.float32, device=dev)
                c3_1 = torch.empty(96, dtype=torch.float32, device=dev)
                g2pre = torch.empty((B, F, T, C), dtype=torch.bfloat16, device=dev)
                check(lib().sp_block_bwd_in(ptr(gz), ptr(z), zs[0], zs[1], zs[2], ptr(y2), B, F, T, ptr(g2pre),
                                            ptr(sums), s), "sp_block_bwd_in")
                check(lib().sp_bn_finalize_bwd(ptr(sums), ptr(save2), ptr(b2w), npos, ptr(c3_2), ptr(gg2),
                                               ptr(gb2), s), "sp_bn_finalize_bwd")
                # (sum g1pre, sum g1pre * y1) for BatchNorm-1's backward comes out of this epilogue
                g1pre = _conv(g2pre, y2, c3_2, IN_BNBWD, p2b, None, y1, coef1, EPI_MASK, B, F, T, sums)
                check(lib().sp_conv3x5_wgrad(ptr(y1), ptr(coef1), IN_BNRELU, ptr(g2pre), ptr(y2), ptr(c3_2),
                                             B, F, T, ptr(ws), ptr(gw2), s), "sp_conv3x5_wgrad")
                check(lib().sp_bn_finalize_bwd(ptr(sums), ptr(save1), ptr(b1w), npos, ptr(c3_1), ptr(gg1),
                                               ptr(gb1), s), "sp_bn_finalize_bwd")
                gx = _conv(g1pre, y1, c3_1, IN_BNBWD, p1b, None, g2pre, None, EPI_ADD, B, F, T)
                check(lib().sp_conv3x5_wgrad(ptr(x), None, IN_ID, ptr(g1pre), ptr(y1), ptr(c3_1), B, F, T,
                                             ptr(ws), ptr(gw1), s), "sp_conv3x5_wgrad")
                # a conv bias feeding batch-statistics BatchNorm has an identically zero gradient
                if direct:
                    grads = [None] * 8 + grads
                else:
                    grads = [gw1, zb1, gg1, gb1, gw2, zb2, gg2, gb2] + grads
                gz = gx
            if direct:
                gsw, gsb = P[0].grad, P[1].grad
            else:
                gsw = torch.zeros(32 * 9, dtype=torch.float32, device=dev)
                gsb = torch.zeros(32, dtype=torch.float32, device=dev)
            check(lib().sp_conv_stem_bwd(ptr(ctx.spec), ptr(gz), B, Fin, Tin, ptr(gsw), ptr(gsb), s),
                  "sp_conv_stem_bwd")
        ctx.saved = None
        if direct:
            sink.ready(*P)
            return (None, None, None, None, None, None, *grads, *([None] * (4 * nblk)))
        return (None, None, None, None, gsw.view(32, 1, 3, 3), gsb, *grads, *([None] * (4 * nblk)))


def _conv_f32(x, xs, cin, w, scale, shift, res, rs, relu, B, Fin, Tin, stride, out, os_):
    """One sp_conv2d_f32 launch; xs / rs / os_ = (batch, frequency, time) strides in elements."""
    KH, KW = int(w.shape[0]), int(w.shape[1])
    rs = rs or (0, 0, 0)
    check(lib().sp_conv2d_f32(ptr(x), xs[0], xs[1], xs[2], cin, ptr(w), ptr(scale), ptr(shift), ptr(res), rs[0], rs[1], rs[2],
                              int(relu), B, Fin, Tin, KH, KW, stride, ptr(out), os_[0], os_[1], os_[2], stream_ptr()),
          "sp_conv2d_f32")
    return out


@torch.no_grad()
def conv_stack_eval_f32(spec, cnn, blocks):
    """The conv stack of net.py:56-59 / :6-42 in fp32 with BatchNorm on its running statistics -- what the reference's
    evaluation computes outside autocast (train.py:105-135, :162-168).  Same output layout as conv_stack:
    [T', B, F*32], feature index f*32 + c, f32.  The convolution weights are permuted to [KH][KW][Cin][32] and the
    BatchNorm folded into (scale, shift) here, on parameter-sized tensors; everything activation-sized runs in
    sp_conv2d_f32."""
    spec = spec.detach().float().contiguous()
    B, _, Fin, Tin = spec.shape
    F, T = (Fin - 1) // 2 + 1, (Tin - 1) // 2 + 1
    dev = spec.device
    with torch.cuda.device(dev):
        ones = torch.ones(C, dtype=torch.float32, device=dev)
        chl = (F * T * C, T * C, C)                                   # [B][F][T][32]
        x = torch.empty((B, F, T, C), dtype=torch.float32, device=dev)
        _conv_f32(spec, (Fin * Tin, Tin, 1), 1, _f32(cnn.weight).permute(2, 3, 1, 0).contiguous(), ones, _f32(cnn.bias),
                  None, None, False, B, Fin, Tin, 2, x, chl)

        def fold(conv, bn):
            scale = _f32(bn.weight) / torch.sqrt(_f32(bn.running_var) + bn.eps)
            shift = (_f32(conv.bias) - _f32(bn.running_mean)) * scale + _f32(bn.bias)
            return _f32(conv.weight).permute(2, 3, 1, 0).contiguous(), scale.contiguous(), shift.contiguous()
        for i, blk in enumerate(blocks):
            w1, sc1, sh1 = fold(blk.conv1, blk.bn1)
            w2, sc2, sh2 = fold(blk.conv2, blk.bn2)
            y1 = torch.empty((B, F, T, C), dtype=torch.float32, device=dev)
            _conv_f32(x, chl, C, w1, sc1, sh1, None, None, True, B, F, T, 1, y1, chl)
            if i == len(blocks) - 1:                                  # [T'][B][F*32]
                z, zs = torch.empty((T, B, F * C), dtype=torch.float32, device=dev), (F * C, C, B * F * C)
            else:
                z, zs = torch.empty((B, F, T, C), dtype=torch.float32, device=dev), chl
            _conv_f32(y1, chl, C, w2, sc2, sh2, x, chl, True, B, F, T, 1, z, zs)
            x = z
    return x


def conv_stack(spec, cnn, blocks, training):
    """cnn: nn.Conv2d(1,32,3,stride 2); blocks: iterable of ResidualBlock modules."""
    params, bufs = [], []
    for blk in blocks:
        params += [blk.conv1.weight, blk.conv1.bias, blk.bn1.weight, blk.bn1.bias,
                   blk.conv2.weight, blk.conv2.bias, blk.bn2.weight, blk.bn2.bias]
        bufs += [blk.bn1.running_mean, blk.bn1.running_var, blk.bn2.running_mean, blk.bn2.running_var]
    if training:      # one launch for all counters instead of one each
        torch._foreach_add_([b.num_batches_tracked for blk in blocks for b in (blk.bn1, blk.bn2)], 1)
    bn = blocks[0].bn1
    return _ConvStackFn.apply(spec, bool(training), float(bn.momentum), float(bn.eps), cnn.weight, cnn.bias,
                              *params, *bufs)
