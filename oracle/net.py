"""Oracle restatement of the acoustic model and of one training step on the CPU
(reference net.py:6-87, train.py:61-87).  Test infrastructure only.

Conv / BatchNorm / LayerNorm / Linear are torch's own CPU ops, exactly what the reference calls;
the SRU stack is ``oracle.sru`` (parity unpinned, see there).  Parameter names follow the
reference's state_dict so that weights can be exchanged with ``speech_amd.net.SRModel``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ctc as octc
from . import sru as osru


def forward(sd, x, n_rnn_layers, rnn_dim, training=False, masks=None, bn_stats=None, bf16_conv=False):
    """sd: state_dict-like mapping name -> tensor (float32/64, CPU).  x [B,1,F,T] -> logits [B,T',V].
    bf16_conv: round where the HIP conv stack rounds -- stem output, the 3x5 convolutions' weights, inputs and
    outputs, every block's output -- so that a comparison with it is left with accumulation order only (the
    reference itself runs these layers under fp16 autocast, train.py:66)."""
    def q(t):
        return t.bfloat16().to(t.dtype) if bf16_conv else t

    def bn(y, prefix):
        return F.batch_norm(y, sd[prefix + ".running_mean"].clone(), sd[prefix + ".running_var"].clone(),
                            sd[prefix + ".weight"], sd[prefix + ".bias"], training=training,
                            momentum=0.1, eps=1e-5)
    y = q(F.conv2d(x, sd["cnn.weight"], sd["cnn.bias"], stride=2, padding=1))
    for i in range(3):
        p = f"resnet_layers.{i}"
        r = y
        y = q(F.conv2d(y, q(sd[p + ".conv1.weight"]), sd[p + ".conv1.bias"], padding=(1, 2)))
        y = q(F.relu(bn(y, p + ".bn1")))
        y = q(F.conv2d(y, q(sd[p + ".conv2.weight"]), sd[p + ".conv2.bias"], padding=(1, 2)))
        y = q(F.relu(bn(y, p + ".bn2") + r))
    B, C, Fq, T = y.shape
    y = y.reshape(B, C * Fq, T).permute(2, 0, 1).contiguous()          # net.py:82-83
    layers = []
    for i in range(n_rnn_layers):
        p = f"sru_layers.rnn_lst.{i}"
        layers.append({"weight": sd[p + ".weight"], "weight_c": sd[p + ".weight_c"], "bias": sd[p + ".bias"],
                       "ln_w": sd[p + ".layer_norm.weight"], "ln_b": sd[p + ".layer_norm.bias"]})
    y = osru.sru_stack(y, layers, rnn_dim, masks)
    y = y.transpose(0, 1)
    y = F.layer_norm(y, (y.shape[-1],), sd["classifier.0.weight"], sd["classifier.0.bias"])
    return y @ sd["classifier.1.weight"].t()


def loss_and_grads(sd, x, labels, label_lengths, n_rnn_layers, rnn_dim, training=True, masks=None, bf16_conv=False):
    """train.py:66-79 on the CPU: forward, log_softmax, CTC (input_lengths all = T'), backward.
    Returns (loss, {name: grad}) for the floating-point parameters."""
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k and k != "scale_x"
             and not k.endswith("scale_x")]
    leaf = {k: (sd[k].detach().clone().requires_grad_(True) if k in names else sd[k]) for k in sd}
    logits = forward(leaf, x, n_rnn_layers, rnn_dim, training=training, masks=masks, bf16_conv=bf16_conv)
    lp = F.log_softmax(logits, dim=2).transpose(0, 1)
    il = torch.full((x.shape[0],), lp.shape[0], dtype=torch.int32)
    loss = F.ctc_loss(lp, labels, il, label_lengths.to(torch.int32), blank=0)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return loss.detach(), dict(zip(names, grads)), logits.detach()
