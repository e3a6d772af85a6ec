"""Generate the committed golden vectors under tests/golden/.

Run in the build container (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py

What is produced, and from what:
  text_golden.json    reference text.py (imported as is): CHARS, text_to_int / int_to_text pairs.
  decode_golden.npz   reference decoder.py greedy_decoder / greedy_decode run on seeded scores.
                      decoder.py line 1 imports python-Levenshtein, which is not installed; an
                      empty module of that name is registered first.  The greedy functions never
                      touch it (only cer/wer do), so their outputs are the reference's own.
  ctc_golden.npz      torch.nn.CTCLoss(blank=0) on the CPU, called exactly as train.py:69-77 does
                      (this is the reference's arithmetic: it calls torch's CTC).
  mel_golden.npz      torch.stft-based restatement (oracle.mel; torchaudio is not installed, so
                      these are regression vectors, parity unpinned at the reference level),
                      plus float64 DFT values of the same inputs.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def import_reference():
    sys.path.insert(0, REF)
    sys.modules.setdefault("Levenshtein", types.ModuleType("Levenshtein"))
    import text as ref_text  # noqa
    import decoder as ref_decoder  # noqa
    sys.path.remove(REF)
    return ref_text, ref_decoder


def make_text(ref_text):
    samples = ["hello world", "it's a test", "", "zzz aaa ' _", "the quick brown fox jumps over the lazy dog"]
    out = {
        "CHARS": ref_text.CHARS,
        "pairs": [{"text": s, "ids": ref_text.text_to_int(s)} for s in samples],
        "bad_chars": ["A", "1", "-", "\n"],
    }
    for p in out["pairs"]:
        assert ref_text.int_to_text(p["ids"]) == p["text"]
    with open(os.path.join(HERE, "text_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


def decode_cases():
    g = torch.Generator().manual_seed(7)
    cases = {}
    # random log-probs, char vocabulary, ragged-free (the reference decodes every frame)
    cases["rand_b4_t50_v29"] = torch.log_softmax(torch.randn(4, 50, 29, generator=g) * 3, dim=2)
    cases["rand_b3_t300_v29"] = torch.log_softmax(torch.randn(3, 300, 29, generator=g) * 2, dim=2)
    # known answer: argmax path [0,0,1,1,0,1,2,2] -> [1,1,2]
    path = [0, 0, 1, 1, 0, 1, 2, 2]
    ka = torch.full((1, len(path), 29), -5.0)
    for t, c in enumerate(path):
        ka[0, t, c] = 0.0
    cases["known_answer"] = ka
    # first frame non-blank, leading repeat
    path2 = [3, 3, 3, 0, 3, 4, 4, 0, 0, 5]
    fa = torch.full((1, len(path2), 29), -9.0)
    for t, c in enumerate(path2):
        fa[0, t, c] = -0.5
    cases["first_frame_nonblank"] = fa
    # ties -> lowest index; all-equal rows decode as blank (index 0)
    ties = torch.zeros(2, 6, 29)
    ties[1, :, 5] = 1.0
    ties[1, :, 9] = 1.0  # tie between 5 and 9 -> 5
    cases["ties"] = ties
    # NaN counts as max
    nan = torch.randn(1, 8, 29, generator=g)
    nan[0, 2, 7] = float("nan")
    nan[0, 3, 7] = float("nan")
    nan[0, 5, 1] = float("nan")
    cases["nan"] = nan
    # all blank
    ab = torch.full((2, 10, 29), -3.0)
    ab[:, :, 0] = 0.0
    cases["all_blank"] = ab
    # single frame
    cases["single_frame"] = torch.log_softmax(torch.randn(5, 1, 29, generator=g), dim=2)
    return cases


def make_decode(ref_text, ref_decoder):
    arrays = {}
    meta = {}
    g = torch.Generator().manual_seed(11)
    for name, scores in decode_cases().items():
        B, T, V = scores.shape
        L = 6
        labels = torch.randint(1, 29, (B, L), generator=g)
        label_lengths = torch.randint(1, L + 1, (B,), generator=g).int()
        dec, tgt = ref_decoder.greedy_decoder(scores, labels, label_lengths)
        dec2 = ref_decoder.greedy_decode(scores)
        assert dec == dec2
        arrays[name + "__scores"] = scores.numpy()
        arrays[name + "__labels"] = labels.numpy()
        arrays[name + "__label_lengths"] = label_lengths.numpy()
        meta[name] = {"decodes": dec, "targets": tgt,
                      "ids": [ref_text.text_to_int(s) for s in dec]}
    np.savez_compressed(os.path.join(HERE, "decode_golden.npz"), **arrays)
    with open(os.path.join(HERE, "decode_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)


def ctc_reference(logits_tbv, labels, input_lengths, label_lengths):
    """train.py:69-77 on the CPU."""
    z = logits_tbv.clone().requires_grad_(True)
    out = torch.nn.functional.log_softmax(z, dim=2)
    crit = torch.nn.CTCLoss(blank=0)
    loss = crit(out, labels, input_lengths, label_lengths)
    loss.backward()
    nll = torch.nn.functional.ctc_loss(out.detach(), labels, input_lengths, label_lengths, blank=0,
                                       reduction="none")
    return loss.detach(), nll, z.grad


def make_ctc():
    g = torch.Generator().manual_seed(7)
    arrays = {}

    def add(name, T, B, V, Lmax, lens=None, ilens=None, force_repeat=False, scale=2.0):
        logits = torch.randn(T, B, V, generator=g) * scale
        labels = torch.randint(1, V, (B, max(Lmax, 1)), generator=g)
        if force_repeat and Lmax >= 2:
            labels[:, 1] = labels[:, 0]
        ll = torch.tensor(lens if lens is not None else [Lmax] * B, dtype=torch.int32)
        il = torch.tensor(ilens if ilens is not None else [T] * B, dtype=torch.int32)
        loss, nll, grad = ctc_reference(logits, labels, il, ll)
        arrays[name + "__logits"] = logits.numpy()
        arrays[name + "__labels"] = labels.numpy()
        arrays[name + "__input_lengths"] = il.numpy()
        arrays[name + "__label_lengths"] = ll.numpy()
        arrays[name + "__loss"] = loss.numpy()
        arrays[name + "__nll"] = nll.numpy()
        arrays[name + "__grad"] = grad.numpy()

    add("char_small", 30, 4, 29, 8, lens=[8, 5, 1, 3], force_repeat=True)
    add("char_ragged_T", 40, 3, 29, 10, lens=[10, 7, 4], ilens=[40, 33, 25])
    add("char_mid", 120, 4, 29, 40, lens=[40, 33, 21, 12], force_repeat=True)       # S = 81: one wave, NS=2
    add("char_s200", 200, 2, 29, 100, lens=[100, 90])                                 # S = 201: NS=4
    add("char_s400", 260, 2, 29, 200, lens=[200, 150])                                # S = 401: 2 waves
    add("sp_vocab", 60, 3, 1000, 20, lens=[20, 11, 6], force_repeat=True)             # gathered rows
    add("t1_l1", 1, 2, 29, 1, lens=[1, 1])
    add("l0", 12, 2, 29, 0, lens=[0, 0])
    add("infeasible", 3, 2, 29, 2, lens=[2, 2], force_repeat=True)                    # "aa" needs T >= 3... L+repeats=3 ok for first
    add("infeasible_short", 2, 1, 29, 2, lens=[2], force_repeat=True)                 # needs 3 frames, has 2 -> inf
    np.savez_compressed(os.path.join(HERE, "ctc_golden.npz"), **arrays)


def make_mel():
    from oracle import mel as omel
    g = torch.Generator().manual_seed(7)
    arrays = {}
    sr = 16000

    def add(name, wav):
        wav = wav.float()
        arrays[name + "__wav"] = wav.numpy()
        arrays[name + "__mel"] = omel.mel_spectrogram(wav).numpy()
        arrays[name + "__mel_f64"] = omel.mel_spectrogram_f64(wav.numpy()).astype(np.float64)

    n = 4000
    t = torch.arange(n) / sr
    add("noise", (0.1 * torch.randn(n, generator=g)).clamp(-1, 1))
    add("sine_1k", 0.5 * torch.sin(2 * np.pi * 1000.0 * t))
    imp = torch.zeros(n)
    imp[1234] = 1.0
    add("impulse", imp)
    add("dc", torch.full((n,), 0.25))
    add("short_201", (0.1 * torch.randn(201, generator=g)))
    add("ragged_1777", (0.1 * torch.randn(1777, generator=g)))   # N not a multiple of 160
    arrays["fbank"] = omel.melscale_fbanks_htk().numpy()
    np.savez_compressed(os.path.join(HERE, "mel_golden.npz"), **arrays)


if __name__ == "__main__":
    ref_text, ref_decoder = import_reference()
    make_text(ref_text)
    make_decode(ref_text, ref_decoder)
    make_ctc()
    make_mel()
    print("golden vectors written to", HERE)
