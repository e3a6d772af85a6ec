"""Acoustic encoder -- same module surface and state_dict layout as the reference's net.py:6-87:
``SRModel(n_rnn_layers, rnn_dim, n_vocab, n_feats, dropout, projection_size)``,
``forward(x[B,1,n_feats,T]) -> [B,T',n_vocab+1]``.

state_dict keys: cnn.{weight,bias}; resnet_layers.{0,1,2}.{conv1,conv2}.*,
.{bn1,bn2}.*; sru_layers.rnn_lst.{i}.{weight,weight_c,bias,layer_norm.weight,layer_norm.bias}
(+ scale_x buffer; + weight_proj when projection_size > 0); classifier.0.{weight,bias}; classifier.1.weight.

The conv/BatchNorm stack (speech_amd.convstack) and the SRU recurrence (ops.sru_recurrence) run
in the HIP kernels.  The training-path conv kernels keep bf16 activations; an fp32 evaluation (eval mode, no
grad, compute_dtype float32 -- what train.eval_dataset and scripts/predict run) takes the f32 convolution
kernel instead, so that it is fp32 end to end like the reference's.  ``compute_dtype`` selects the
activation type of the SRU stack and classifier: float32, bfloat16 (the training type here) or
float16 (the reference trains under fp16 autocast with an fp16 recurrence, net.py:71, and evaluates
with model.half(), scripts/eval_dataset.py).  ResidualBlock.forward below is the plain
torch statement of one block and is used only by tests as the reference of the kernels.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import convstack, ops


class ResidualBlock(nn.Module):
    """net.py:6-42: conv(3 x t) -> BN -> ReLU -> conv -> BN -> += residual -> ReLU."""

    def __init__(self, in_channels, out_channels, t_kernel_s):
        super().__init__()
        pad = (3 // 2, t_kernel_s // 2)
        self.conv1 = nn.Conv2d(in_channels, out_channels, (3, t_kernel_s), 1, padding=pad)
        self.conv2 = nn.Conv2d(out_channels, out_channels, (3, t_kernel_s), 1, padding=pad)
        self.relu = nn.ReLU(inplace=True)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        for conv in (self.conv1, self.conv2):
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
        nn.init.ones_(self.bn1.weight)
        nn.init.zeros_(self.bn1.bias)
        nn.init.zeros_(self.bn2.weight)  # blocks start as the identity (net.py:30)
        nn.init.zeros_(self.bn2.bias)

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y = y + x
        return self.relu(y)


class SRUCell(nn.Module):
    """One bidirectional SRU layer with input layer-norm (sru.SRUCell as net.py:61-72 configures it).

    weight [n_in, 2*d*k] with columns ordered (dir, channel, k) -- the package's layout, kept for
    checkpoint compatibility; the kernels want (k, dir, channel), so the projection uses a
    permuted view of it.
    """

    def __init__(self, input_size, hidden_size, dropout=0.0, highway_bias=0.0, in_layout=None, projection_size=0):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        # in_layout = (C, F): the input arrives with feature index f*C + c while `weight` rows and
        # the layer-norm vectors are stored in the reference's c*F + f order (net.py:82)
        self.in_layout = in_layout
        self.output_size = 2 * hidden_size
        self.dropout = dropout
        self.projection_size = projection_size
        self.num_matrices = 3 if input_size == self.output_size else 4
        # sru.SRUCell(projection_size = p > 0): U = (LN(x) weight_proj) weight, a rank-p factorisation
        if projection_size > 0:
            self.weight_proj = nn.Parameter(torch.empty(input_size, projection_size))
            self.weight = nn.Parameter(torch.empty(projection_size, self.output_size * self.num_matrices))
        else:
            self.weight = nn.Parameter(torch.empty(input_size, self.output_size * self.num_matrices))
        self.weight_c = nn.Parameter(torch.empty(2 * self.output_size))
        self.bias = nn.Parameter(torch.empty(2 * self.output_size))
        self.register_buffer("scale_x", torch.ones(1))
        self.layer_norm = nn.LayerNorm(input_size)
        self.highway_bias = highway_bias
        self.reset_parameters()

    def reset_parameters(self):
        """Package defaults as recalled (parity unpinned): W ~ U(+-sqrt(3/fan_in)) * 0.1 under layer
        norm, the u0 columns scaled by sqrt(1-dropout); v ~ U(+-sqrt(3)) * sqrt(0.5); b = 0 with the
        reset-gate half set to highway_bias; weight_proj ~ U(+-sqrt(3/n_in))."""
        with torch.no_grad():
            fan_in = self.projection_size or self.input_size
            r = (3.0 / fan_in) ** 0.5
            self.weight.uniform_(-r, r)
            w = self.weight.view(fan_in, 2, self.hidden_size, self.num_matrices)
            if self.dropout > 0:
                w[..., 0].mul_((1 - self.dropout) ** 0.5)
            self.weight.mul_(0.1)
            if self.projection_size > 0:
                rp = (3.0 / self.input_size) ** 0.5
                self.weight_proj.uniform_(-rp, rp)
            self.weight_c.uniform_(-(3.0 ** 0.5), 3.0 ** 0.5).mul_(0.5 ** 0.5)
            self.bias.zero_()
            self.bias[self.output_size:].fill_(self.highway_bias)

    def kernel_weight(self, dtype):
        """[n_in, k, 2, d] column order for the kernels, flattened, in the compute dtype."""
        k, d = self.num_matrices, self.hidden_size
        w = self.weight.view(self.input_size, 2, d, k).permute(0, 3, 1, 2)
        if self.in_layout is not None:
            C, Fq = self.in_layout
            w = w.reshape(C, Fq, k, 2, d).transpose(0, 1)
        return w.reshape(self.input_size, k * 2 * d).to(dtype)

    def _ln_vec(self, v, dtype):
        if self.in_layout is not None:
            C, Fq = self.in_layout
            v = v.view(C, Fq).t().reshape(-1)
        return v.to(dtype)

    def forward(self, x, mask=None, pre=None):
        """mask: optional pre-drawn variational dropout mask [B, 2d] (already / keep); SRU.forward draws
        the masks of all layers in one call.  pre: this layer's packed weights from ops.sru_prepack."""
        T, B, n_in = x.shape
        d = self.hidden_size
        ln = self.layer_norm
        if mask is None and self.training and self.dropout > 0:
            keep = 1.0 - self.dropout
            mask = torch.bernoulli(torch.full((B, 2 * d), keep, device=x.device)) / keep
        C = self.in_layout[0] if self.in_layout is not None else 1
        if self.projection_size > 0:
            # LayerNorm -> rank-p projection -> gate projection -> recurrence, each a HIP kernel; the (small)
            # weights are brought into the kernels' row / column order by differentiable torch views
            k, p = self.num_matrices, self.projection_size
            xn = ops.layer_norm(x, self._ln_vec(ln.weight, torch.float32), self._ln_vec(ln.bias, torch.float32), ln.eps)
            wp = self.weight_proj
            if self.in_layout is not None:
                wp = wp.view(C, n_in // C, p).transpose(0, 1).reshape(n_in, p)
            wk = self.weight.view(p, 2, d, k).permute(0, 3, 1, 2).reshape(p, k * 2 * d)
            U = ops.matmul(ops.matmul(xn.reshape(T * B, n_in), wp), wk).view(T, B, k, 2, d)
            return ops.sru_recurrence(U, x if k == 3 else None, self.weight_c, self.bias, mask, d)
        return ops.sru_layer(x, ln.weight, ln.bias, self.weight, self.weight_c, self.bias, mask, d,
                             self.num_matrices, C, ln.eps, pre)


class SRU(nn.Module):
    """sru.SRU(input_size, hidden_size, num_layers, dropout, bidirectional=True, layer_norm=True,
    rescale=False, highway_bias=0): dropout on every layer but the last."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.0, projection_size=0,
                 highway_bias=0.0, in_layout=None):
        super().__init__()
        self.rnn_lst = nn.ModuleList(
            SRUCell(input_size if i == 0 else 2 * hidden_size, hidden_size,
                    dropout=dropout if i + 1 != num_layers else 0.0, highway_bias=highway_bias,
                    in_layout=in_layout if i == 0 else None, projection_size=projection_size)
            for i in range(num_layers))

    def forward(self, x):
        masks = None
        drops = [c.dropout for c in self.rnn_lst]
        if self.training and any(p > 0 for p in drops):
            # every layer's variational mask at once instead of three launches per layer
            B, d2 = x.shape[1], self.rnn_lst[0].output_size
            pos = sorted({p for p in drops if p > 0})
            if len(pos) == 1:      # one rate (the reference's case): scalar-probability kernel, in place
                keep = 1.0 - pos[0]
                masks = torch.empty((len(drops), B, d2), device=x.device).bernoulli_(keep).mul_(1.0 / keep)
            else:
                keep = getattr(self, "_keep", None)
                if keep is None or keep.device != x.device or keep.shape[0] != len(drops):
                    keep = self._keep = torch.tensor([1.0 - p for p in drops], device=x.device).view(-1, 1, 1)
                masks = torch.bernoulli(keep.expand(len(drops), B, d2)) / keep
        # the packed projection weights of all same-shaped layers in one launch (nine of the ten here)
        cells = [(c.weight, c.hidden_size, c.num_matrices, c.in_layout[0] if c.in_layout is not None else 1)
                 for c in self.rnn_lst]
        packs = {}
        if x.is_cuda and all(c.projection_size == 0 for c in self.rnn_lst):
            packs = ops.sru_prepack(cells, x.shape[0], x.shape[1], x.dtype, torch.is_grad_enabled())
        for i, cell in enumerate(self.rnn_lst):
            m = masks[i] if masks is not None and drops[i] > 0 else None
            x = cell(x, m, packs.get(i))
        return x, None


class SRModel(nn.Module):
    def __init__(self, n_rnn_layers, rnn_dim, n_vocab, n_feats, dropout, projection_size=0,
                 compute_dtype=torch.float32):
        super().__init__()
        self.cnn = nn.Conv2d(1, 32, 3, stride=2, padding=3 // 2)
        self.resnet_layers = nn.Sequential(*[ResidualBlock(32, 32, t_kernel_s=5) for _ in range(3)])
        n_features = 32 * n_feats // 2
        self.sru_layers = SRU(n_features, rnn_dim, n_rnn_layers, dropout=dropout,
                              projection_size=projection_size, in_layout=(32, n_feats // 2))
        self.classifier = nn.Sequential(nn.LayerNorm(2 * rnn_dim),
                                        nn.Linear(2 * rnn_dim, n_vocab + 1, bias=False))
        self.compute_dtype = compute_dtype

    def encode(self, x):
        """[B, 1, F, T] f32 spectrogram -> layer-normed encoder output, time-major [T', B, 2*rnn_dim]."""
        cd = self.compute_dtype
        if cd == torch.float32 and not self.training and not torch.is_grad_enabled():
            # fp32 evaluation (train.py:162-168, outside autocast): the conv stack in f32 as well
            x = convstack.conv_stack_eval_f32(x, self.cnn, list(self.resnet_layers))
        else:
            # conv stack on the channel-last bf16 kernels -> [T', B, (F/2)*32], feature = f*32 + c
            x = convstack.conv_stack(x, self.cnn, list(self.resnet_layers), self.training)
        x, _ = self.sru_layers(x.to(cd))
        ln = self.classifier[0]
        return ops.layer_norm(x, ln.weight, ln.bias, ln.eps)

    def forward(self, x):  # [B, 1, F, T] f32 spectrogram -> f32 logits [B, T', V]
        x = self.encode(x).transpose(0, 1).contiguous()
        return ops.classifier(x, self.classifier[1].weight)

    def ctc_loss(self, x, labels, label_lengths, max_label_len=None):
        """The training step's loss, CTCLoss(blank=0)(log_softmax(self(x)), ...) of train.py:69-77, with the
        classifier and the loss as one node (ops.classifier_ctc_loss)."""
        return ops.classifier_ctc_loss(self.encode(x), self.classifier[1].weight, labels, label_lengths, 0,
                                       max_label_len)

    # keys whose presence depends on the version of the `sru` package / torch that wrote the checkpoint and that
    # carry no trained values: sru registers `scale_x` (1.0 unless rescale=True; net.py:66 passes rescale=False)
    # as a buffer in some releases only; BatchNorm's step counter is bookkeeping
    _OPTIONAL_KEYS = (".scale_x", ".num_batches_tracked")

    def load_checkpoint(self, obj):
        """Accept both checkpoint shapes of the reference: the dict train.py:171-178 saves
        ({"epoch","model_state_dict","optimizer_state_dict"}) and the bare state_dict that
        scripts/predict.py:35 and scripts/eval_dataset.py:119 load.  Keys in _OPTIONAL_KEYS may be missing or
        extra (reported in the result, never an error); any other mismatch raises, as load_state_dict does.
        The key layout of the sru layers is recalled from the package, not verified against a reference
        checkpoint (none ships with the reference): parity unpinned."""
        sd = obj["model_state_dict"] if isinstance(obj, dict) and "model_state_dict" in obj else obj
        own = self.state_dict()
        optional = lambda k: k.endswith(self._OPTIONAL_KEYS)      # noqa: E731
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        bad = [k for k in missing + unexpected if not optional(k)]
        if bad:
            raise RuntimeError("checkpoint does not match SRModel: missing " + str([k for k in missing if not optional(k)])
                               + ", unexpected " + str([k for k in unexpected if not optional(k)]))
        self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)


def out_frames(t):
    """Encoder frames for t mel frames: the stride-2 stem (net.py:56)."""
    return (t - 1) // 2 + 1
