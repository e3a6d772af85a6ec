"""GPU tests of the host-side mirror of the reference interface (net / data / decoder / train /
scripts) running on the HIP kernels, against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import decode as odec
from oracle import mel as omel
from oracle import net as onet
from oracle import sru as osru
from oracle import text as otext

pytestmark = pytest.mark.gpu


def _small_model(cuda, dtype=torch.float32, layers=2, dim=64, vocab=28, seed=0):
    from speech_amd import net
    torch.manual_seed(seed)
    m = net.SRModel(layers, dim, vocab, 80, 0.0, 0, compute_dtype=dtype).to(cuda)
    with torch.no_grad():
        for blk in m.resnet_layers:
            blk.bn2.weight.uniform_(0.5, 1.0)      # the default (0) makes every block the identity
    return m


def test_state_dict_layout_matches_reference_names(cuda):
    """key names and shapes of net.py's state_dict."""
    from speech_amd import net
    m = net.SRModel(10, 512, 28, 80, 0.1, 0)
    sd = m.state_dict()
    assert tuple(sd["cnn.weight"].shape) == (32, 1, 3, 3)
    assert tuple(sd["resnet_layers.2.conv2.weight"].shape) == (32, 32, 3, 5)
    assert "resnet_layers.0.bn1.num_batches_tracked" in sd
    assert float(sd["resnet_layers.1.bn2.weight"].abs().max()) == 0.0           # net.py:30
    assert tuple(sd["sru_layers.rnn_lst.0.weight"].shape) == (1280, 4096)
    assert tuple(sd["sru_layers.rnn_lst.9.weight"].shape) == (1024, 3072)
    assert tuple(sd["sru_layers.rnn_lst.3.weight_c"].shape) == (2048,)
    assert tuple(sd["sru_layers.rnn_lst.3.layer_norm.weight"].shape) == (1024,)
    assert tuple(sd["classifier.1.weight"].shape) == (29, 1024)
    assert sum(p.numel() for p in m.parameters()) == 33741184                   # SURVEY "Model facts"
    # both checkpoint shapes load
    m.load_checkpoint({"epoch": 1, "model_state_dict": sd, "optimizer_state_dict": {}})
    m.load_checkpoint(sd)


@pytest.mark.parametrize("train", [False, True])
def test_model_forward_matches_cpu_oracle(cuda, train):
    """SRModel on the HIP kernels vs oracle.net on the CPU with the same state_dict.
    eval (the reference's fp32 evaluation, train.py:162-168): fp32 end to end -- f32 conv kernel, f32 SRU stack --
    within 2e-3 of the logit range of the plain f32 oracle.
    train (batch-statistics BatchNorm, bf16 conv stack, f32 SRU stack): 2.5 % of the logit range against the plain f32
    oracle (measured 1.2 %), 1 % against the oracle that rounds to bf16 where the conv stack does (measured 0.56 %:
    accumulation order and the batch statistics of unrounded sums)."""
    m = _small_model(cuda)
    m.train(train)
    g = torch.Generator().manual_seed(1)
    spec = torch.rand(2, 1, 80, 121, generator=g) * 3
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        out = m(spec.to(cuda)).float().cpu()
        ref = onet.forward(sd, spec, 2, 64, training=train)
        ref_q = onet.forward(sd, spec, 2, 64, training=train, bf16_conv=True)
    assert out.shape == ref.shape == (2, 61, 29)
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    err_q = float((out - ref_q).abs().max()) / float(ref_q.abs().max())
    print("encoder forward (train=%s): max error vs f32 oracle %.5f, vs bf16-rounding oracle %.5f of the logit range" % (train, err, err_q))
    if train:
        assert err <= 2.5e-2
        assert err_q <= 1e-2      # against the oracle that rounds where the conv stack rounds: accumulation order
    else:
        assert err <= 2e-3


def test_conv2d_f32_matches_torch(cuda):
    """sp_conv2d_f32 (the fp32 evaluation's conv kernel) against torch's CPU convolution in f32: the stem (1 -> 32,
    3 x 3, stride 2), a block convolution (32 -> 32, 3 x 5) with folded BatchNorm + ReLU, and one with the residual, in
    both output layouts; ragged sizes."""
    from speech_amd import convstack
    g = torch.Generator().manual_seed(4)
    for B, Fin, Tin in ((2, 80, 121), (1, 7, 9), (3, 16, 300)):
        spec = torch.randn(B, 1, Fin, Tin, generator=g)
        w = torch.randn(32, 1, 3, 3, generator=g) * 0.3
        bias = torch.randn(32, generator=g)
        ref = torch.nn.functional.conv2d(spec, w, bias, stride=2, padding=1)                  # [B, 32, F, T]
        F, T = ref.shape[2], ref.shape[3]
        out = torch.empty((B, F, T, 32), device=cuda)
        convstack._conv_f32(spec.to(cuda), (Fin * Tin, Tin, 1), 1, w.permute(2, 3, 1, 0).contiguous().to(cuda),
                            torch.ones(32, device=cuda), bias.to(cuda), None, None, False, B, Fin, Tin, 2, out,
                            (F * T * 32, T * 32, 32))
        assert float((out.cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        x = torch.randn(B, 32, F, T, generator=g)
        w2 = torch.randn(32, 32, 3, 5, generator=g) * 0.05
        sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
        res = torch.randn(B, 32, F, T, generator=g)
        conv = torch.nn.functional.conv2d(x, w2, None, padding=(1, 2))
        ref1 = torch.relu(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        ref2 = torch.relu(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
        xl, rl = x.permute(0, 2, 3, 1).contiguous().to(cuda), res.permute(0, 2, 3, 1).contiguous().to(cuda)
        chl = (F * T * 32, T * 32, 32)
        wl = w2.permute(2, 3, 1, 0).contiguous().to(cuda)
        o1 = convstack._conv_f32(xl, chl, 32, wl, sc.to(cuda), sh.to(cuda), None, None, True, B, F, T, 1,
                                 torch.empty((B, F, T, 32), device=cuda), chl)
        assert float((o1.cpu().permute(0, 3, 1, 2) - ref1).abs().max()) <= 2e-5 * float(ref1.abs().max())
        o2 = convstack._conv_f32(xl, chl, 32, wl, sc.to(cuda), sh.to(cuda), rl, chl, True, B, F, T, 1,
                                 torch.empty((T, B, F * 32), device=cuda), (F * 32, 32, B * F * 32))      # [T][B][F*32]
        got = o2.cpu().view(T, B, F, 32).permute(1, 3, 2, 0)
        assert float((got - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())


LOSS_TOL, GRAD_TOL = 5e-3, 2e-2   # measured: loss 4e-4, gradients behind the conv stack 3e-3 .. 5e-3


def test_model_gradients_match_cpu_oracle(cuda):
    from speech_amd import ops
    m = _small_model(cuda, layers=2)
    m.train()
    g = torch.Generator().manual_seed(2)
    spec = torch.rand(2, 1, 80, 101, generator=g) * 3
    labels = torch.randint(1, 29, (2, 6), generator=g)
    ll = torch.tensor([6, 4], dtype=torch.int32)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    logits = m(spec.to(cuda)).float()
    il = torch.full((2,), logits.shape[1], dtype=torch.int32, device=cuda)
    loss = ops.ctc_loss(logits, labels.to(cuda), il, ll.to(cuda))
    loss.backward()
    ref_loss, ref_grads, _ = onet.loss_and_grads(sd, spec, labels, ll, 2, 64, training=True)
    q_loss, q_grads, _ = onet.loss_and_grads(sd, spec, labels, ll, 2, 64, training=True, bf16_conv=True)
    print("loss rel err: f32 oracle %.4f, bf16-rounding oracle %.4f" % (abs(float(loss) / float(ref_loss) - 1), abs(float(loss) / float(q_loss) - 1)))
    assert float(loss) == pytest.approx(float(ref_loss), rel=LOSS_TOL)
    named = dict(m.named_parameters())
    # everything behind the conv stack: 2 % (measured 0.3-0.5 % against the f32 oracle, 0.1 % against the one that
    # rounds like the conv stack).  The conv stack's own parameters: their gradients pass through bf16 activation
    # gradients and batch-statistics BatchNorm backward, which subtracts the gradient's mean and its projection on
    # the normalised activation -- a cancellation that amplifies the 2^-9 rounding of the stored tensors:
    # measured 6-12 % of the largest entry against the f32 oracle, 4 % against the rounding oracle.
    for k in ["classifier.1.weight", "sru_layers.rnn_lst.1.weight", "sru_layers.rnn_lst.1.weight_c",
              "sru_layers.rnn_lst.0.weight", "sru_layers.rnn_lst.0.layer_norm.weight", "sru_layers.rnn_lst.0.bias"]:
        a, b, c = named[k].grad.cpu(), ref_grads[k], q_grads[k]
        print("%-45s f32 oracle %.4f  bf16-rounding oracle %.4f" % (k, float((a - b).abs().max()) / float(b.abs().max()),
                                                                     float((a - c).abs().max()) / float(c.abs().max())))
        assert float((a - b).abs().max()) <= GRAD_TOL * float(b.abs().max()) + 1e-7, k
    for k in ["resnet_layers.2.conv2.weight", "resnet_layers.0.conv1.weight", "resnet_layers.1.bn1.weight", "cnn.weight"]:
        a, b, c = named[k].grad.cpu(), ref_grads[k], q_grads[k]
        print("%-45s f32 oracle %.4f  bf16-rounding oracle %.4f" % (k, float((a - b).abs().max()) / float(b.abs().max()),
                                                                     float((a - c).abs().max()) / float(c.abs().max())))
        assert float((a - b).abs().max()) <= 1.5e-1 * float(b.abs().max()) + 1e-7, k     # measured <= 12.1 %
        assert float((a - c).abs().max()) <= 5e-2 * float(c.abs().max()) + 1e-7, k       # measured <= 4.4 %; the kernels
        # on this path add in fixed orders (float atomics only join f32 partial sums of the stem's weight gradient: last bits),
        # so the figures do not move from run to run
        cos = float((a.flatten() * c.flatten()).sum() / (a.norm() * c.norm()))
        print("    cosine with the rounding oracle %.5f" % cos)
        assert cos >= 0.995, k


def test_adam_clips_the_mean_gradient_under_data_parallel(cuda):
    """Data parallel: the arena holds the SUM of the ranks' gradients and grad_div = world.  The clip must act
    on the mean gradient's norm, so that a world-2 step equals a single-process step on the averaged gradient
    with the same max_norm (Adam's eps is raised so the update is not scale-invariant)."""
    from speech_amd.train import FlatAdam
    torch.manual_seed(5)
    n, world, max_norm = 4099, 2.0, 0.7
    p0 = torch.randn(n, device=cuda)
    g_mean = torch.randn(n, device=cuda) * 0.05
    assert float(g_mean.norm()) > 2 * max_norm                  # the clip is active
    outs = []
    for scale, div in ((world, world), (1.0, 1.0)):
        p = torch.nn.Parameter(p0.clone())
        opt = FlatAdam([p], lr=1e-2, eps=1e-2, weight_decay=0.0)
        for t in range(3):
            opt.zero_grad()
            p.grad.copy_(g_mean * scale * (1 + t))
            opt.step(max_norm=max_norm, grad_div=div)
        outs.append(p.detach().clone())
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-6
    # and against the arithmetic written out
    q = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([q], lr=1e-2, eps=1e-2)
    for t in range(3):
        q.grad = g_mean * (1 + t)
        torch.nn.utils.clip_grad_norm_([q], max_norm)
        ref.step()
    assert float((outs[0] - q.detach()).abs().max()) <= 1e-5


def test_flat_adam_matches_torch_adam(cuda):
    from speech_amd.train import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n, device=cuda)) for n in (1000, 37, 4096, 3)]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FlatAdam(ps, lr=1e-2, weight_decay=0.0)
    ref = torch.optim.Adam(qs, lr=1e-2)
    for step in range(5):
        gs = [torch.randn_like(p) * (10.0 if step == 2 else 0.1) for p in ps]
        opt.zero_grad()
        for p, q, gr in zip(ps, qs, gs):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(qs, 2.0)
        ref.step()
        opt.step(max_norm=2.0)
        for p, q in zip(ps, qs):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), step
    # decoupled weight decay: p *= 1 - lr*wd after the Adam update
    opt2 = FlatAdam([torch.nn.Parameter(torch.ones(8, device=cuda))], lr=0.1, weight_decay=0.5)
    opt2.zero_grad()
    opt2.step()
    assert torch.allclose(opt2.plist[0], torch.full((8,), 0.95, device=cuda))


def test_collate_contract_and_specaugment(cuda):
    """data.collate_fn returns the reference's batch contract; masks follow torchaudio's draw order."""
    from speech_amd import data, text
    g = torch.Generator().manual_seed(4)
    clips = [(torch.randn(1, n, generator=g) * 0.1, s) for n, s in
             [(8000, "hello world"), (12000, "it's a test"), (4000, "a")]]
    spec, labels, ll = data.collate_fn(clips, "valid")
    assert spec.shape == (3, 1, 80, 76) and spec.dtype == torch.float32 and spec.is_cuda
    assert labels.dtype == torch.int64 and labels.shape == (3, 11) and ll.dtype == torch.int32
    assert labels[2].tolist() == text.text_to_int("a") + [0] * 10 and ll.tolist() == [11, 11, 1]
    ref = omel.collate([w for w, _ in clips])
    assert float((spec.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    torch.manual_seed(123)
    spec_t, _, _ = data.collate_fn(clips, "train")
    torch.manual_seed(123)
    masks = []
    for w, _ in clips:   # oracle.mel.mask_params on the same uniform draws, same order
        tb = 1 + w.numel() // 160
        u = [float(torch.rand(1)) for _ in range(2)]
        f0, fw = omel.mask_params(u[0], u[1], 15, 80)
        u = [float(torch.rand(1)) for _ in range(2)]
        t0, tw = omel.mask_params(u[0], u[1], 35, tb)
        masks.append((f0, fw, t0, tw))
    ref_t = omel.collate([w for w, _ in clips], masks)
    assert float((spec_t.cpu() - ref_t).abs().max()) <= 1e-4 * float(ref.abs().max())
    # the transform objects mirror data.spectrogram_transform's shape contract
    one = data.spectrogram_transform(clips[0][0])
    assert one.shape == (1, 80, 51)


def test_wer_parity_on_1k_synthetic_utterances(cuda):
    """BASELINE configs[4]: greedy decode of 1000 synthetic utterances; decoded strings, CER and WER
    identical to the CPU reference path (oracle.decode = decoder.py semantics)."""
    from speech_amd import decoder, text
    g = torch.Generator().manual_seed(7)
    B, T, V = 1000, 200, 29
    # render a random target alignment + noise so that WER is neither 0 nor 1
    L = 30
    labels = torch.randint(1, V, (B, L), generator=g)
    labels[:, ::5] = 2                                              # spaces -> words
    ll = torch.randint(10, L + 1, (B,), generator=g).int()
    scores = torch.randn(B, T, V, generator=g) * 0.8
    for b in range(B):
        pos = torch.sort(torch.randperm(T // 2, generator=g)[: int(ll[b])]).values * 2
        scores[b, :, 0] += 3.2
        scores[b, pos, labels[b, : int(ll[b])]] += 6.0
    dec, tgt = decoder.greedy_decoder(scores.to(cuda), labels, ll)
    rdec, rtgt = odec.greedy_decoder(scores, labels, ll)
    assert dec == rdec and tgt == rtgt
    keep = [i for i in range(B) if tgt[i].replace(" ", "") and tgt[i].split()]
    cer = [decoder.cer(tgt[i], dec[i]) for i in keep]
    wer = [decoder.wer(tgt[i], dec[i]) for i in keep]
    rcer = [odec.cer(rtgt[i], rdec[i]) for i in keep]
    rwer = [odec.wer(rtgt[i], rdec[i]) for i in keep]
    assert cer == rcer and wer == rwer
    assert 0.0 < sum(wer) / len(wer) < 1.0
    assert text.CHARS == otext.CHARS


def test_train_step_reduces_loss_bf16(cuda):
    from speech_amd.train import FlatAdam, train_step
    m = _small_model(cuda, dtype=torch.bfloat16, layers=2, dim=64)
    m.train()
    opt = FlatAdam(m.parameters(), lr=2e-3, weight_decay=1e-3)
    g = torch.Generator().manual_seed(3)
    spec = (torch.rand(4, 1, 80, 161, generator=g) * 3).to(cuda)
    labels = torch.randint(1, 29, (4, 8), generator=g).to(cuda)
    ll = torch.tensor([8, 6, 5, 3], dtype=torch.int32, device=cuda)
    losses = [float(train_step(m, opt, spec, labels, ll, 2.0)) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses


def test_train_step_fp16(cuda):
    """--dtype fp16 (the reference's AMP type): the whole training step, incl. the fused classifier + CTC node whose
    gradient leaves the CTC kernels as float16, runs and learns."""
    from speech_amd.train import FlatAdam, train_step
    m = _small_model(cuda, dtype=torch.float16, layers=2, dim=64)
    m.train()
    opt = FlatAdam(m.parameters(), lr=2e-3, weight_decay=1e-3)
    g = torch.Generator().manual_seed(3)
    spec = (torch.rand(4, 1, 80, 161, generator=g) * 3).to(cuda)
    labels = torch.randint(1, 29, (4, 8), generator=g).to(cuda)
    ll = torch.tensor([8, 6, 5, 3], dtype=torch.int32, device=cuda)
    losses = [float(train_step(m, opt, spec, labels, ll, 2.0)) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses


def test_ctc_f16_gradient_is_the_rounded_f32_gradient(cuda):
    from speech_amd import ops
    g = torch.Generator().manual_seed(5)
    for T, B, V, L in ((40, 3, 29, 7), (70, 2, 1000, 12)):
        logits = (torch.randn(B, T, V, generator=g) * 2).to(cuda)
        labels = torch.randint(1, V, (B, L), generator=g).to(cuda)
        il = torch.full((B,), T, dtype=torch.int32, device=cuda)
        ll = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32).to(cuda)
        _, _, g32 = ops.ctc_fwd_bwd(logits, labels, il, ll, max_label_len=L, grad_scale=64.0)
        _, _, g16 = ops.ctc_fwd_bwd(logits, labels, il, ll, max_label_len=L, grad_scale=64.0, grad_dtype=torch.float16)
        assert g16.dtype == torch.float16
        assert torch.equal(g16, g32.to(torch.float16))


def test_ctc_host_label_lengths_beyond_the_bound_raise(cuda):
    """Host-side lengths are checked against max_label_len before anything is launched (device-side ones are clamped
    by the kernel, as the header says)."""
    from speech_amd import ops
    logits = torch.randn(2, 30, 29, device=cuda)
    labels = torch.randint(1, 29, (2, 9), device=cuda)
    il = torch.full((2,), 30, dtype=torch.int32, device=cuda)
    with pytest.raises(ValueError, match="exceeds max_label_len"):
        ops.ctc_fwd_bwd(logits, labels, il, torch.tensor([9, 4], dtype=torch.int32), max_label_len=6)
    ops.ctc_fwd_bwd(logits, labels, il, torch.tensor([6, 4], dtype=torch.int32), max_label_len=6)


def test_resume_restores_optimizer_and_schedule(cuda, tmp_path):
    """save_checkpoint -> resume_from -> the next step is the step the uninterrupted run takes (weights, Adam moments,
    step count); a bare state_dict restores the weights only."""
    from speech_amd import train
    g = torch.Generator().manual_seed(3)
    spec = (torch.rand(4, 1, 80, 161, generator=g) * 3).to(cuda)
    labels = torch.randint(1, 29, (4, 8), generator=g).to(cuda)
    ll = torch.tensor([8, 6, 5, 3], dtype=torch.int32, device=cuda)
    m = _small_model(cuda, dtype=torch.float32).train()
    opt = train.FlatAdam(m.parameters(), lr=2e-3, weight_decay=1e-3)
    for _ in range(3):
        train.train_step(m, opt, spec, labels, ll, 2.0)
    path = tmp_path / "ck.pth"
    train.save_checkpoint(str(path), 2, m, opt, iterations=3)
    train.train_step(m, opt, spec, labels, ll, 2.0)
    want = opt.flat.detach().clone()
    m2 = _small_model(cuda, dtype=torch.float32, seed=5).train()
    opt2 = train.FlatAdam(m2.parameters(), lr=2e-3, weight_decay=1e-3)
    epoch, iters = train.resume_from(str(path), m2, opt2)
    assert (epoch, iters, opt2.t) == (2, 3, 3)
    train.train_step(m2, opt2, spec, labels, ll, 2.0)
    # the forward is bit-reproducible; weight gradients add with f32 atomics (last-bit differences through Adam)
    assert float((opt2.flat - want).abs().max()) <= 2 * 2e-3
    assert float((opt2.exp_avg - opt.exp_avg).abs().max()) <= 1e-5 * float(opt.exp_avg.abs().max()) + 1e-9
    m3 = _small_model(cuda, dtype=torch.float32, seed=6)
    opt3 = train.FlatAdam(m3.parameters(), lr=2e-3)
    torch.save(m.state_dict(), tmp_path / "bare.pth")
    assert train.resume_from(str(tmp_path / "bare.pth"), m3, opt3) == (0, 0) and opt3.t == 0


def test_eval_dataset_cli_on_disk_corpora(cuda, tmp_path):
    """scripts/eval_dataset.py's command line as the reference wires it (:70-129): `libri` through SortedLibriSpeech on a
    LibriSpeech-layout stand-in, `ibm` through IBMDataset batched by the loader; fp16 activations by default
    (model.half()); greedy and beam decode."""
    import pickle
    import wave
    from speech_amd.scripts import eval_dataset as ev
    rs = np.random.RandomState(4)

    def wav(path, n):
        with wave.open(str(path), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((rs.randn(n) * 2500).clip(-32768, 32767).astype("<i2").tobytes())
    libri = tmp_path / "LibriSpeech" / "test-clean" / "7" / "11"
    libri.mkdir(parents=True)
    entries, lines = [], []
    for k, n in enumerate([5000, 5600, 6100, 6900]):
        fid = f"7-11-{k:04d}"
        wav(libri / (fid + ".wav"), n)
        lines.append(f"{fid} SOME WORDS HERE")
        entries.append((str(libri / (fid + ".wav")), n))
    (libri / "7-11.trans.txt").write_text("\n".join(lines) + "\n")
    mdir = tmp_path / "m"
    (mdir / "datasets" / "librispeech").mkdir(parents=True)
    (mdir / "datasets" / "librispeech" / "sorted_test_clean_librispeech.pkl").write_bytes(pickle.dumps(entries))
    ibm = tmp_path / "ibm"
    ibm.mkdir()
    for k in range(3):
        wav(ibm / f"i{k}.wav", 6000 + 300 * k)
        (ibm / f"i{k}.txt").write_text("ibm line")
    m = _small_model(cuda, layers=2, dim=64)
    ckpt = tmp_path / "m.pth"
    torch.save(m.state_dict(), ckpt)                         # a bare state_dict, as eval_dataset.py:119 loads it
    common = ["--n_rnn_layers", "2", "--rnn_dim", "64", "--batch_size", "2", "--num_workers", "0"]
    logs = []
    cer, wer = ev.main(["False", "libri", str(ckpt), "--manifest_root", str(mdir), "--root_map", "/nowhere=/x"] + common,
                       log=logs.append)
    assert np.isfinite(cer) and np.isfinite(wer) and any("Average CER" in str(l) for l in logs)
    cer32, _ = ev.main(["False", "libri", str(ckpt), "--manifest_root", str(mdir), "--root_map", "/nowhere=/x",
                        "--dtype", "fp32"] + common, log=lambda *a: None)
    assert abs(cer - cer32) <= 0.25                          # an untrained model: the two precisions decode alike
    cer_b, _ = ev.main(["True", "ibm", str(ckpt), "--ibm_root", str(ibm)] + common, log=lambda *a: None)
    assert np.isfinite(cer_b)


def test_scripts_predict_and_eval(cuda, tmp_path):
    import wave
    from speech_amd import data
    from speech_amd.scripts import eval_dataset, predict
    m = _small_model(cuda, layers=1, dim=64)
    ckpt = tmp_path / "m.pth"
    torch.save({"epoch": 1, "model_state_dict": m.state_dict(), "optimizer_state_dict": {}}, ckpt)
    m2 = _small_model(cuda, layers=1, dim=64, seed=5)
    m2.load_checkpoint(torch.load(ckpt, weights_only=True))
    m2.eval()
    pcm = (np.sin(np.arange(16000) * 0.05) * 8000).astype(np.int16)
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    out = predict.predict(path, m2)
    assert isinstance(out, list) and len(out) == 1 and isinstance(out[0], str)
    ds = data.SyntheticSorted(8, 4, max_s=2.0, seed=3)
    loader = [data.collate_fn(ds[i], "valid") for i in range(len(ds))]
    cer, wer = eval_dataset.test("synthetic", 4, m2, loader, False, log=lambda *a: None)
    assert cer >= 0 and wer >= 0
    # beam_decode=True (eval_dataset.py:20-27, :43-51), without and with a language model
    cer_b, wer_b = eval_dataset.test("synthetic", 4, m2, loader, True, log=lambda *a: None)
    assert cer_b >= 0 and wer_b >= 0
    arpa = tmp_path / "lm.arpa"
    arpa.write_text("\\data\\\nngram 1=5\n\n\\1-grams:\n-1.0\t<unk>\n-99\t<s>\n-1.0\t</s>\n-0.6\tthe\n-0.9\tcat\n\n\\end\\\n")
    cer_l, wer_l = eval_dataset.test("synthetic", 4, m2, loader, True, log=lambda *a: None, lm_path=str(arpa))
    assert cer_l >= 0 and wer_l >= 0


def test_grad_sink_matches_autograd_accumulation(cuda):
    """train_step's direct path (backward kernels accumulate straight into FlatAdam's gradient arena,
    ops.GradSink) against plain autograd accumulation on the same model and batch."""
    import copy
    from speech_amd import ops
    from speech_amd.train import FlatAdam
    m1 = _small_model(cuda, dtype=torch.bfloat16, layers=2, dim=64)
    m1.train()
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(11)
    spec = (torch.rand(3, 1, 80, 141, generator=g) * 3).to(cuda)
    labels = torch.randint(1, 29, (3, 7), generator=g).to(cuda)
    ll = torch.tensor([7, 5, 3], dtype=torch.int32, device=cuda)

    def run(model, sink):
        out = model(spec).float()
        il = torch.full((3,), out.shape[1], dtype=torch.int32, device=cuda)
        ops.set_grad_sink(sink)
        try:
            ops.ctc_loss(out, labels, il, ll).backward()
        finally:
            ops.set_grad_sink(None)

    run(m1, None)                                    # autograd allocates and accumulates
    opt = FlatAdam(m2.parameters(), lr=1e-3)
    opt.zero_grad()
    fired = []

    class _DP:   # stands in for BucketedAllReduce: every parameter must be reported exactly once
        def param_ready(self, p):
            fired.append(("sink", p.data_ptr()))
    for p in m2.parameters():
        p.register_post_accumulate_grad_hook(lambda q: fired.append(("hook", q.data_ptr())))
    run(m2, ops.GradSink(_DP()))
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        g1 = p1.grad if p1.grad is not None else torch.zeros_like(p1)
        # forward and activation gradients are bit-reproducible; the two paths differ by the order of
        # f32 additions only (atomics in the weight-gradient kernels, += versus a separate add)
        tol = 2e-3 * float(g1.abs().max()) + 1e-6
        assert float((g1 - p2.grad).abs().max()) <= tol, k
    names = {p.data_ptr(): k for k, p in m2.named_parameters()}
    seen = [names.get(a, "?") for _, a in fired]
    # (autograd may additionally run a parameter's own hook; BucketedAllReduce counts a parameter once)
    missing = sorted(set(names.values()) - set(seen))
    assert not missing and "?" not in seen, (missing, fired[:4])
    by_sink = {names[a] for kind, a in fired if kind == "sink"}
    assert "sru_layers.rnn_lst.1.weight" in by_sink and "cnn.weight" in by_sink and "classifier.0.weight" in by_sink


def test_training_forward_is_bit_reproducible(cuda):
    """Same parameters, same batch, training mode (batch statistics): the logits and the loss of two
    runs are bit-identical -- the BatchNorm reductions use no float atomics."""
    from speech_amd import ops
    outs = []
    for _ in range(2):
        m = _small_model(cuda, torch.bfloat16)
        m.train()
        g = torch.Generator().manual_seed(3)
        spec = (torch.rand(4, 1, 80, 301, generator=g) * 3).to(cuda)
        labels = torch.randint(1, 29, (4, 20), generator=g).to(cuda)
        lens = torch.tensor([20, 11, 15, 8], dtype=torch.int32, device=cuda)
        out = m(spec)
        il = torch.full((4,), out.shape[1], dtype=torch.int32, device=cuda)
        loss = ops.ctc_loss(out.float(), labels, il, lens, blank=0, max_label_len=20)
        outs.append((out.detach().clone(), loss.detach().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_head_matches_classifier_then_ctc(cuda, dtype):
    """SRModel.ctc_loss (classifier + CTC as one node on time-major logits, what train_step uses) against
    ops.ctc_loss(model(x)) (the reference's two steps, train.py:69-77): same loss, same gradients."""
    from speech_amd import ops
    g = torch.Generator().manual_seed(4)
    spec = (torch.rand(3, 1, 80, 161, generator=g) * 3).to(cuda)
    labels = torch.randint(1, 29, (3, 9), generator=g).to(cuda)
    lens = torch.tensor([9, 5, 7], dtype=torch.int32, device=cuda)
    res = []
    for fused in (False, True):
        m = _small_model(cuda, dtype)
        m.train()
        if fused:
            loss = m.ctc_loss(spec, labels, lens, 9)
        else:
            out = m(spec)
            il = torch.full((3,), out.shape[1], dtype=torch.int32, device=cuda)
            loss = ops.ctc_loss(out.float(), labels, il, lens, blank=0, max_label_len=9)
        (loss * 3.0).backward()          # a non-unit incoming gradient
        res.append((loss.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    assert float((res[0][0] - res[1][0]).abs()) <= 1e-6 * float(res[0][0].abs())
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        # f32 stack: last-bit differences of the head; the conv stack keeps bf16 activation gradients, which
        # turn those into rounding flips (5e-3).  bf16: the CTC gradient is rounded before the GEMMs (2e-2)
        conv = k.startswith("cnn") or k.startswith("resnet")
        tol = 2e-2 if dtype == torch.bfloat16 else (5e-3 if conv else 1e-5)
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()) + 1e-7, k


def test_projection_kernels_agree(cuda):
    """Which GEMM kernel runs the projection is decided by the shape alone: a 4608-row layer takes the
    persistent LDS-DMA kernels on the packed, pitch-padded W^T, its two batch halves (2304 rows each) the
    general kernel on W.  Utterances are independent in an SRU layer, so both must give the same
    outputs and input gradients, and the halves' weight gradients must add up to the whole's -- to bf16
    rounding."""
    from speech_amd import ops
    torch.manual_seed(3)
    T, B, d, k = 72, 64, 512, 3
    n_in = 2 * d
    x0 = (torch.randn(T, B, n_in, device=cuda) * 0.7).bfloat16()
    ln_w, ln_b = torch.rand(n_in, device=cuda) + 0.5, torch.randn(n_in, device=cuda) * 0.1
    weight = torch.randn(n_in, 2 * d * k, device=cuda) * (1.0 / n_in) ** 0.5
    wc, bias = torch.randn(4 * d, device=cuda) * 0.3, torch.randn(4 * d, device=cuda) * 0.1
    assert ops._nt_projection(T * B, n_in, 2 * d * k, d, torch.bfloat16)
    assert not ops._nt_projection(T * B // 2, n_in, 2 * d * k, d, torch.bfloat16)
    gh = torch.randn(T, B, 2 * d, device=cuda).bfloat16()

    def run(xs, ghs):
        ps = [p.clone().requires_grad_(True) for p in (ln_w, ln_b, weight, wc, bias)]
        x = xs.clone().requires_grad_(True)
        h = ops.sru_layer(x, *ps, None, d, k)
        h.backward(ghs)
        return h.detach().float(), x.grad.float(), ps[2].grad.float()
    whole = run(x0, gh)
    lo, hi = run(x0[:, :B // 2].contiguous(), gh[:, :B // 2].contiguous()), run(x0[:, B // 2:].contiguous(), gh[:, B // 2:].contiguous())
    halves = (torch.cat([lo[0], hi[0]], 1), torch.cat([lo[1], hi[1]], 1), lo[2] + hi[2])
    for a, b, tol in zip(halves, whole, (2e-2, 2e-2, 5e-3)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())


def test_wav_dataset_to_train_step(cuda, tmp_path):
    """The IO path end to end: 16-bit WAV files + transcripts on disk -> SortedTV -> DataLoader ->
    collate_fn (int16 PCM straight into the fused frontend) -> train_step; the features equal the CPU
    oracle's on the same samples."""
    import pickle
    import wave
    from speech_amd import data
    from speech_amd.train import FlatAdam, train_step
    rs = np.random.RandomState(3)
    entries = []
    for k, n in enumerate([6400, 7000, 8000, 8123]):
        p = tmp_path / f"clip{k}.wav"
        with wave.open(str(p), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((rs.randn(n) * 3000).clip(-32768, 32767).astype("<i2").tobytes())
        (tmp_path / f"clip{k}.txt").write_text("hello world" if k % 2 else "a test")
        entries.append((str(p), n))
    (tmp_path / "m.pkl").write_bytes(pickle.dumps(entries))
    ds = data.SortedTV([str(tmp_path / "m.pkl")], 2, 0, root_map=())
    loader = torch.utils.data.DataLoader(ds, batch_size=None, shuffle=False,
                                         collate_fn=lambda b: data.collate_fn(b, "valid"))
    batches = list(loader)
    assert len(batches) == 2
    spec, labels, ll = batches[1]
    assert spec.is_cuda and tuple(spec.shape) == (2, 1, 80, 1 + 8123 // 160) and ll.tolist() == [6, 11]
    ref = omel.collate([w.float() / 32768.0 for w, _ in ds[1]])
    assert float((spec.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    m = _small_model(cuda, torch.bfloat16)
    m.train()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    l0 = float(train_step(m, opt, spec, labels.to(cuda), ll.to(cuda), 2.0))
    for _ in range(5):
        l1 = float(train_step(m, opt, spec, labels.to(cuda), ll.to(cuda), 2.0))
    assert np.isfinite(l1) and l1 < l0


def test_filter_alignments_workload(cuda, tmp_path, monkeypatch):
    """scripts/tv/filter_alignments.py mirror: manifest -> MetaTV (score filter, > 6 s clips silenced) ->
    batched inference + GPU greedy decode -> first/last-word rule."""
    import pickle
    import wave
    from speech_amd.scripts import filter_alignments as fa
    rs = np.random.RandomState(5)
    entries = []
    for k, (n, score, txt) in enumerate([(8000, 0.1, "hello big world"), (9000, 0.5, "dropped by score"),
                                         (7000, 0.2, ""), (16000 * 7, 0.0, "too long clip"), (6000, 0.3, "one")]):
        p = tmp_path / f"c{k}.wav"
        with wave.open(str(p), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((rs.randn(n) * 2000).astype("<i2").tobytes())
        if k != 2:
            (tmp_path / f"c{k}.txt").write_text(txt)
        entries.append((str(p), score, n / 16000))
    (tmp_path / "manifest.pkl").write_bytes(pickle.dumps(entries))
    ds = fa.MetaTV(str(tmp_path / "manifest.pkl"), 4, root_map=())
    assert [m[1] for m in ds.paths] == [0.3, 0.2, 0.1, 0.0]                 # sorted by duration, score <= 0.3 kept
    clips = ds[0]
    assert clips[3][0].shape[1] == 16000 and float(clips[3][0].abs().max()) == 0     # the 7 s clip is silenced
    assert clips[1][1] is None                                                         # no transcript file
    spec, transcripts, manifests = fa.collate_fn(clips)
    assert spec.is_cuda and spec.shape[0] == 4 and transcripts[2] == "hello big world"
    # the rule, independent of the model
    assert fa.keep_entry("hello small world", "hello big world") and not fa.keep_entry("hello", "hello world")
    assert not fa.keep_entry("hello big word", "hello big world") and not fa.keep_entry("a b", None)
    # end to end, nothing mocked: the HIP decode inside filter_files must agree with the CPU oracle's decode of the
    # model's own logits, string for string, and the kept manifest must be the rule applied to those strings.  The
    # classifier is biased so that the (untrained) model emits several symbols, spaces included
    model = _small_model(cuda, torch.bfloat16).eval()
    with torch.no_grad():
        model.classifier[1].weight.mul_(40.0)
    loader = torch.utils.data.DataLoader(ds, batch_size=None, collate_fn=fa.collate_fn, shuffle=False)
    seen = []
    real_decode = fa.decoder.greedy_decode

    def spy(out, *a, **kw):
        res = real_decode(out, *a, **kw)
        seen.append((out.detach().float().cpu(), res))
        return res
    monkeypatch.setattr(fa.decoder, "greedy_decode", spy)
    kept = fa.filter_files(model, loader)
    assert len(seen) == 1
    logits, hip_strings = seen[0]
    oracle_strings = odec.greedy_decode(logits)
    assert hip_strings == oracle_strings
    assert len(set(oracle_strings)) > 1 or len(oracle_strings[0]) > 0
    expect = [m for p_, t_, m in zip(oracle_strings, transcripts, manifests) if fa.keep_entry(p_, t_)]
    assert kept == expect


def test_train_main_on_disk_corpus(cuda, tmp_path, monkeypatch):
    """train.main wired as train.py:202-304: CombinedTVLibriSpeech (LibriSpeech + TV manifests merged by duration)
    -> DataLoader workers decode, the training process featurises on the GPU -> train steps; evaluation on the
    SortedTV `eval` manifests and the IBM directory in fp32 (train.py:162-168), checkpoint in the reference's
    shape when the IBM CER improves."""
    import pickle
    import wave
    from speech_amd import net, train
    rs = np.random.RandomState(9)

    def wav(path, n):
        with wave.open(str(path), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((rs.randn(n) * 2500).clip(-32768, 32767).astype("<i2").tobytes())
    libri = tmp_path / "LibriSpeech" / "train-clean" / "7" / "11"
    libri.mkdir(parents=True)
    lines, libri_entries = [], []
    for k, n in enumerate([5000, 5600, 6100, 6900]):
        fid = f"7-11-{k:04d}"
        wav(libri / (fid + ".wav"), n)
        lines.append(f"{fid} SOME WORDS HERE")
        libri_entries.append((str(libri / (fid + ".wav")), n))
    (libri / "7-11.trans.txt").write_text("\n".join(lines) + "\n")
    tv = tmp_path / "clean"
    tv.mkdir()
    tv_entries = []
    for k, n in enumerate([5200, 6400, 7000, 7300]):
        wav(tv / f"c{k}.wav", n)
        (tv / f"c{k}.txt").write_text("the tv said so")
        tv_entries.append((str(tv / f"c{k}.wav"), n / 16000))
    ibm = tmp_path / "ibm"
    ibm.mkdir()
    for k in range(3):
        wav(ibm / f"i{k}.wav", 6000 + 300 * k)
        (ibm / f"i{k}.txt").write_text("ibm line")
    mdir = tmp_path / "m"
    (mdir / "datasets" / "librispeech").mkdir(parents=True)
    (mdir / "datasets" / "first").mkdir(parents=True)
    big = libri_entries + [("/nowhere/LibriSpeech/x.wav", 10 ** 9)] * 2000          # survives the [:-2000] cut
    (mdir / "datasets" / "librispeech" / "sorted_train_librispeech.pkl").write_bytes(pickle.dumps(big))
    (mdir / "datasets" / "first" / "sorted_train_cer_0.1.pkl").write_bytes(pickle.dumps(tv_entries))
    (mdir / "datasets" / "first" / "sorted_eval_cer_0.1.pkl").write_bytes(pickle.dumps(tv_entries))
    ckpt = tmp_path / "model.pth"
    hp = train.parse_args(["--one_iter", "--multiplier", "1", "--manifest_root", str(mdir), "--ibm_root", str(ibm),
                           "--num_workers", "0", "--checkpoint", str(ckpt), "--dropout", "0.0"])
    hp.update(batch_size=2, n_rnn_layers=2, rnn_dim=64, tv_manifests=[str(mdir / "datasets" / "first" / "sorted_train_cer_0.1.pkl")])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    cer = train.main(hp)
    assert np.isfinite(cer) and cer <= 2.0
    saved = torch.load(ckpt, weights_only=True)
    # train.py:171-178's keys (+ the scheduler position a resume needs)
    assert set(saved) == {"epoch", "model_state_dict", "optimizer_state_dict", "iterations"} and saved["epoch"] == 1
    m = net.SRModel(2, 64, 28, 80, 0.0).to(cuda)
    m.load_checkpoint(saved)


def test_configs1_full_size_train_step(cuda):
    """BASELINE configs[1] at its real size -- batch 32 x 15 s, 10 x 512 bidirectional SRU, V = 1000, bf16: the
    frontend + forward + fused classifier/CTC is bit-reproducible from run to run (no float atomics feed a bf16
    tensor), the loss is finite, and three optimisation steps on the batch lower it."""
    from speech_amd import net, ops
    from speech_amd.train import FlatAdam, train_step
    B, N, V, L = 32, 240000, 1000, 60
    g = torch.Generator().manual_seed(7)
    wav = (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1).reshape(-1).to(cuda)
    offs = (torch.arange(B + 1, dtype=torch.int64) * N).to(cuda)
    labels = torch.randint(1, V, (B, L), generator=g).to(cuda)
    lens = torch.randint(L // 2, L + 1, (B,), generator=g).to(torch.int32).to(cuda)
    torch.manual_seed(1)
    m = net.SRModel(10, 512, V - 1, 80, 0.1, 0, compute_dtype=torch.bfloat16).to(cuda).train()
    spec = ops.melspec(wav, offs)
    assert tuple(spec.shape) == (B, 1, 80, 1501)
    losses = []
    for _ in range(2):
        torch.manual_seed(11)                     # same dropout masks
        with torch.no_grad():
            losses.append(m.ctc_loss(spec, labels, lens, L).clone())
    assert torch.isfinite(losses[0]) and torch.equal(losses[0], losses[1])
    opt = FlatAdam(m.parameters(), lr=3e-4, weight_decay=1e-3)
    steps = [float(train_step(m, opt, spec, labels, lens, 2.0, None, max_label_len=L)) for _ in range(4)]
    assert all(np.isfinite(steps)) and steps[-1] < steps[0]


@pytest.mark.parametrize("vocab", ["sp", "char"])
def test_configs3_full_size_train_step(cuda, vocab):
    """BASELINE configs[3] at its real size -- long form: batch 64 x 30 s, T = 3001 mel frames, T' = 1501 encoder frames,
    bf16 -- with the sentencepiece-sized vocabulary (V = 1000, up to 120 pieces) and with the reference's characters
    (V = 29, up to 448 labels: S = 897 lattice states, the multi-wave lattice): frontend + forward + fused
    classifier/CTC bit-reproducible from run to run, finite, and the first optimisation step lowers the loss (with 448
    random character labels on 1501 frames the next Adam steps of an untrained model bounce: measured 11.1 -> 8.5 -> 15.3)."""
    from speech_amd import net, ops
    from speech_amd.train import FlatAdam, train_step
    B, N = 64, 480000
    V, L = (1000, 120) if vocab == "sp" else (29, 448)
    g = torch.Generator().manual_seed(7)
    wav = (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1).reshape(-1).to(cuda)
    offs = (torch.arange(B + 1, dtype=torch.int64) * N).to(cuda)
    labels = torch.randint(1, V, (B, L), generator=g).to(cuda)
    lens = torch.randint(L // 2, L + 1, (B,), generator=g).to(torch.int32).to(cuda)
    lens[0] = L
    torch.manual_seed(1)
    m = net.SRModel(10, 512, V - 1, 80, 0.1, 0, compute_dtype=torch.bfloat16).to(cuda).train()
    spec = ops.melspec(wav, offs)
    assert tuple(spec.shape) == (B, 1, 80, 3001)
    losses = []
    for _ in range(2):
        torch.manual_seed(11)                     # same dropout masks
        with torch.no_grad():
            losses.append(m.ctc_loss(spec, labels, lens, L).clone())
    assert torch.isfinite(losses[0]) and torch.equal(losses[0], losses[1])
    opt = FlatAdam(m.parameters(), lr=3e-4, weight_decay=1e-3)
    steps = [float(train_step(m, opt, spec, labels, lens, 2.0, None, max_label_len=L)) for _ in range(3)]
    assert all(np.isfinite(steps)) and steps[1] < steps[0], steps


def test_bench_self_launch_path(cuda):
    """`python bench.py --gpus N` starts its own rank processes (no torch.distributed.run needed).  One card
    allows one rank, so the launcher is driven with N = 1 (BENCH_FORCE_SPAWN): the child reads the launcher's
    environment like any rank and the parent relays rank 0's single JSON line with the contract's fields."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["BENCH_FORCE_SPAWN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--kernel-iters", "2"], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "audio-sec/sec" and d["value"] > 0
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1


def test_adam8_matches_the_blockwise_oracle(cuda):
    """state_bits = 8 (train.py:254 Adam8bit; parity unpinned at the package level): the kernel against the numpy
    restatement -- code tables, block absmax, nearest-code rounding, clip and decoupled decay -- over several steps
    on an arena with a ragged last block; codes may differ by one step of the table where a moment sits on a
    rounding tie, parameters agree to float rounding."""
    from oracle import adam8
    from speech_amd.train import FlatAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(1000, 3), (2048,), (37, 5), (4096 + 12,)]
    params = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(cuda)) for s in shapes]
    opt = FlatAdam(params, lr=1e-2, weight_decay=1e-2, state_bits=8)
    n = opt.total
    ref = adam8.Adam8(n, 1e-2, weight_decay=1e-2)
    p_ref = opt.flat.cpu().numpy().copy()
    for step in range(6):
        grad = (torch.randn(n, generator=g) * (10.0 if step == 2 else 1.0)).to(cuda)
        opt.flat_grad.copy_(grad)
        opt.step(max_norm=2.0, grad_div=2.0)
        p_ref, m_ref, v_ref = ref.step(p_ref, grad.cpu().numpy(), max_norm=2.0, grad_div=2.0)
        got = opt.flat.cpu().numpy()
        np.testing.assert_allclose(got, p_ref, rtol=2e-5, atol=2e-6)
        q1, q2 = opt.exp_avg.cpu().numpy().astype(int), opt.exp_avg_sq.cpu().numpy().astype(int)
        assert np.abs(q1 - ref.q1.astype(int)).max() <= 1 and np.abs(q2 - ref.q2.astype(int)).max() <= 1
        assert (q1 != ref.q1).mean() < 1e-3 and (q2 != ref.q2).mean() < 1e-3
        np.testing.assert_allclose(opt.absmax1.cpu().numpy(), ref.a1, rtol=1e-4)
        np.testing.assert_allclose(opt.absmax2.cpu().numpy(), ref.a2, rtol=1e-4)   # v carries the clip coefficient squared (f32 norm, other summation order)
        # keep the two in lock step: the comparison is per step, not of accumulated rounding ties
        ref.q1, ref.q2 = opt.exp_avg.cpu().numpy().copy(), opt.exp_avg_sq.cpu().numpy().copy()
        ref.a1, ref.a2 = opt.absmax1.cpu().numpy().copy(), opt.absmax2.cpu().numpy().copy()
        p_ref = got.copy()
    sd = opt.state_dict()
    assert sd["state_bits"] == 8 and sd["exp_avg"].dtype == torch.uint8
    opt.load_state_dict(sd)


def test_adam8_trains_like_fp32_adam(cuda):
    """The same small model and batches under fp32 and 8-bit optimizer state: the loss falls alike."""
    from speech_amd.train import FlatAdam, train_step
    losses = {}
    for bits in (32, 8):
        m = _small_model(cuda, dtype=torch.bfloat16, layers=2, dim=64)
        m.train()
        opt = FlatAdam(m.parameters(), lr=2e-3, weight_decay=1e-3, state_bits=bits)
        g = torch.Generator().manual_seed(11)
        spec = (torch.rand(4, 1, 80, 161, generator=g) * 3).to(cuda)
        labels = torch.randint(1, 29, (4, 8), generator=g).to(cuda)
        ll = torch.tensor([8, 6, 5, 3], dtype=torch.int32, device=cuda)
        torch.manual_seed(0)
        losses[bits] = [float(train_step(m, opt, spec, labels, ll, 2.0)) for _ in range(15)]
    assert np.isfinite(losses[8]).all() and losses[8][-1] < 0.7 * losses[8][0]
    assert abs(losses[8][-1] - losses[32][-1]) < 0.15 * losses[32][0], (losses[8][-1], losses[32][-1])


def test_bench_two_ranks_report_roofline_and_exchange(cuda):
    """The N > 1 line of bench.py, rehearsed on the one-GPU lease: two ranks share the card over gloo
    (BENCH_SHARE_GPU / BENCH_BACKEND; RCCL refuses two ranks per device).  Rank 0's single JSON line carries the
    whole-job rate, a roofline object, and the gradient exchange's per-bucket times and overlap fraction."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(BENCH_SHARE_GPU="1", BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--seconds", "4"], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "dp2" and "cpu_baseline" not in d and d["roofline"]["frac"] > 0
    c = d["comm"]
    assert len(c["buckets_mb"]) == len(c["allreduce_ms"]) >= 2 and all(t > 0 for t in c["allreduce_ms"])
    assert abs(sum(c["buckets_mb"]) - 4 * 33.5) < 4 * 3         # the whole f32 gradient arena
    assert c["overlap_frac"] is not None and c["overlap_frac"] <= 1.0 and c["backend"] == "gloo"
    assert c["ms_per_step_without_exchange"] > 0


def test_model_forward_fp16(cuda):
    """compute_dtype = float16 (the reference evaluates with model.half(), scripts/eval_dataset.py): LayerNorm, the
    projections, the recurrence and the classifier run on their f16 instantiations; logits against the CPU
    oracle, and decode parity on them."""
    from speech_amd import decoder
    m = _small_model(cuda, torch.float16).eval()
    g = torch.Generator().manual_seed(2)
    spec = torch.rand(3, 1, 80, 141, generator=g) * 3
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        out = m(spec.to(cuda))
        ref = onet.forward(sd, spec, 2, 64, training=False)
    assert out.dtype == torch.float32 and out.shape == ref.shape
    assert float((out.cpu() - ref).abs().max()) <= 5e-2 * float(ref.abs().max())      # the conv stack keeps bf16 activations
    assert decoder.greedy_decode_ids(out) == odec.greedy_decode_ids(out.cpu())


def test_sru_layer_fp16_big_shape(cuda):
    """One 4608-row SRU layer in fp16 (persistent NT kernel on the packed W^T, generic recurrence kernel) against
    the same layer in fp32."""
    from speech_amd import ops
    torch.manual_seed(6)
    T, B, d, k = 72, 64, 512, 3
    n_in = 2 * d
    x0 = torch.randn(T, B, n_in, device=cuda) * 0.7
    ln_w, ln_b = torch.rand(n_in, device=cuda) + 0.5, torch.randn(n_in, device=cuda) * 0.1
    weight = torch.randn(n_in, 2 * d * k, device=cuda) * (1.0 / n_in) ** 0.5
    wc, bias = torch.randn(4 * d, device=cuda) * 0.3, torch.randn(4 * d, device=cuda) * 0.1
    gh = torch.randn(T, B, 2 * d, device=cuda)
    res = []
    for dt in (torch.float32, torch.float16):
        ps = [p.clone().requires_grad_(True) for p in (ln_w, ln_b, weight, wc, bias)]
        x = x0.clone().to(dt).requires_grad_(True)
        h = ops.sru_layer(x, *ps, None, d, k)
        h.backward(gh.to(dt))
        res.append((h.detach().float(), x.grad.float(), ps[2].grad.float()))
    for a, b, tol in zip(res[1], res[0], (4e-3, 4e-3, 4e-3)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_sru_projection_size(cuda, dtype, tol):
    """projection_size = p > 0 (net.py:65, train.py:319 --projection_size): U = (LN(x) weight_proj) weight.  The
    factorised layer equals the plain layer with the product matrix weight_proj @ weight -- checked against the CPU
    oracle's layer on that product, outputs and all gradients (chain rule through the factors)."""
    from speech_amd import net
    torch.manual_seed(8)
    T, B, d, n_in, p = 19, 3, 64, 128, 32
    cell = net.SRUCell(n_in, d, dropout=0.0, projection_size=p).to(cuda)
    assert tuple(cell.weight_proj.shape) == (n_in, p) and tuple(cell.weight.shape) == (p, 2 * d * 3)
    with torch.no_grad():
        cell.weight.mul_(10.0)
        cell.bias.uniform_(-0.2, 0.2)
    x = torch.randn(T, B, n_in, device=cuda)
    xd = x.clone().to(dtype).requires_grad_(True)
    h = cell(xd)
    gh = torch.randn(T, B, 2 * d, device=cuda)
    h.backward(gh.to(dtype))
    # oracle on W_eff = weight_proj @ weight, with the factors as leaves
    wp = cell.weight_proj.detach().cpu().double().requires_grad_(True)
    w = cell.weight.detach().cpu().double().requires_grad_(True)
    xr = x.detach().cpu().double().requires_grad_(True)
    ln = cell.layer_norm
    href, _ = osru.sru_layer(xr, wp @ w, cell.weight_c.detach().cpu().double(), cell.bias.detach().cpu().double(),
                             ln.weight.detach().cpu().double(), ln.bias.detach().cpu().double(), d)
    href.backward(gh.cpu().double())

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= tol * max(1e-6, float(b.abs().max())), what
    close(h, href, "h"); close(xd.grad, xr.grad, "gx")
    close(cell.weight_proj.grad, wp.grad, "gweight_proj"); close(cell.weight.grad, w.grad, "gweight")
    # and inside the whole model with the reference's constructor argument
    m = net.SRModel(2, 64, 28, 80, 0.0, 32, compute_dtype=dtype).to(cuda)
    assert "sru_layers.rnn_lst.0.weight_proj" in m.state_dict()
    out = m(torch.rand(2, 1, 80, 101, device=cuda))
    assert out.shape == (2, 51, 29) and torch.isfinite(out).all()
