"""CPU tests: the oracle against the committed golden vectors (which come from the reference's
own code where it is importable, see tests/golden/make_golden.py) and against known answers."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import decode as odec
from oracle import mel as omel
from oracle import text as otext


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


# ------------------------------------------------------------------------------- text
def test_text_matches_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "text_golden.json")))
    assert otext.CHARS == g["CHARS"] and len(otext.CHARS) == 29 and otext.CHARS[0] == "_"
    for p in g["pairs"]:
        assert otext.text_to_int(p["text"]) == p["ids"]
        assert otext.int_to_text(p["ids"]) == p["text"]
    for c in g["bad_chars"]:
        with pytest.raises(KeyError):
            otext.text_to_int(c)


# ------------------------------------------------------------------------------- decode
def test_decode_matches_reference(golden_dir):
    arrs = _load(golden_dir, "decode_golden.npz")
    meta = json.load(open(os.path.join(golden_dir, "decode_golden.json")))
    for name, m in meta.items():
        scores = torch.from_numpy(arrs[name + "__scores"])
        labels = torch.from_numpy(arrs[name + "__labels"])
        ll = torch.from_numpy(arrs[name + "__label_lengths"])
        dec, tgt = odec.greedy_decoder(scores, labels, ll)
        assert dec == m["decodes"], name
        assert tgt == m["targets"], name
        assert odec.greedy_decode(scores) == m["decodes"], name
        assert odec.greedy_decode_ids(scores) == m["ids"], name


def test_decode_known_answers():
    assert odec.collapse([0, 0, 1, 1, 0, 1, 2, 2]) == [1, 1, 2]
    assert odec.collapse([3, 3, 0, 3]) == [3, 3]
    assert odec.collapse([]) == []
    assert odec.collapse([0, 0, 0]) == []


def test_cer_wer():
    assert odec.levenshtein("kitten", "sitting") == 3
    assert odec.levenshtein("", "abc") == 3
    assert odec.cer("a b c", "abc") == 0.0
    assert odec.cer("abcd", "abed") == 0.25
    assert odec.wer("the cat sat", "the cat sat") == 0.0
    assert odec.wer("the cat sat", "the dog sat") == pytest.approx(1 / 3)
    assert odec.wer("a b", "a b c d") == 1.0
    with pytest.raises(ZeroDivisionError):
        odec.wer("", "a")
    with pytest.raises(ZeroDivisionError):
        odec.cer("  ", "a")


# ------------------------------------------------------------------------------- ctc
CTC_CASES = ["char_small", "char_ragged_T", "char_mid", "sp_vocab", "t1_l1", "l0"]


@pytest.mark.parametrize("name", CTC_CASES)
def test_ctc_numpy_matches_reference(golden_dir, name):
    g = _load(golden_dir, "ctc_golden.npz")
    logits, labels = g[name + "__logits"], g[name + "__labels"]
    il, ll = g[name + "__input_lengths"], g[name + "__label_lengths"]
    loss, nll, grad = octc.ctc_numpy(logits, labels, il, ll)
    assert np.allclose(loss, g[name + "__loss"], rtol=1e-5, atol=1e-6)
    assert np.allclose(nll, g[name + "__nll"], rtol=1e-5, atol=1e-5)
    gref = g[name + "__grad"]
    assert np.abs(grad - gref).max() <= 2e-4 * np.abs(gref).max() + 1e-7


def test_ctc_torch_reproduces_golden(golden_dir):
    g = _load(golden_dir, "ctc_golden.npz")
    for name in CTC_CASES + ["char_s200", "char_s400"]:
        loss, nll, grad = octc.ctc_torch(g[name + "__logits"], g[name + "__labels"],
                                         g[name + "__input_lengths"], g[name + "__label_lengths"])
        assert torch.allclose(loss, torch.from_numpy(g[name + "__loss"]))
        assert torch.allclose(grad, torch.from_numpy(g[name + "__grad"]))


def test_ctc_known_answers(golden_dir):
    # uniform probabilities, T=1, L=1: p = 1/V -> nll = log V
    V = 29
    z = np.zeros((1, 1, V), dtype=np.float32)
    loss, nll, _ = octc.ctc_numpy(z, np.array([[3]]), [1], [1])
    assert nll[0] == pytest.approx(np.log(V), rel=1e-6)
    # L = 0: nll = -sum_t lp[blank]
    z = np.random.RandomState(0).randn(5, 1, V).astype(np.float32)
    lp = torch.log_softmax(torch.from_numpy(z), 2).numpy()
    _, nll, _ = octc.ctc_numpy(z, np.zeros((1, 1), dtype=np.int64), [5], [0])
    assert nll[0] == pytest.approx(-lp[:, 0, 0].sum(), rel=1e-5)
    # "aa" with T = 2 is infeasible
    g = _load(golden_dir, "ctc_golden.npz")
    assert np.isinf(g["infeasible_short__nll"]).all()
    _, nll, _ = octc.ctc_numpy(g["infeasible_short__logits"], g["infeasible_short__labels"],
                               g["infeasible_short__input_lengths"], g["infeasible_short__label_lengths"])
    assert np.isinf(nll).all()
    # input_lengths < T  =>  zero gradient on the tail
    gr = g["char_ragged_T__grad"]
    assert np.all(gr[33:, 1] == 0) and np.all(gr[25:, 2] == 0)


# ------------------------------------------------------------------------------- mel
def test_fbank_properties():
    fb = omel.melscale_fbanks_htk()
    assert tuple(fb.shape) == (201, 80)
    assert int((fb != 0).sum()) == 392
    assert float(fb.sum()) == pytest.approx(196.151962, rel=1e-5)
    assert int((fb.sum(0) == 0).sum()) == 0       # no empty filter


def test_mel_matches_golden_and_f64(golden_dir):
    g = _load(golden_dir, "mel_golden.npz")
    for name in ["noise", "sine_1k", "impulse", "dc", "short_201", "ragged_1777"]:
        wav = g[name + "__wav"]
        mel = omel.mel_spectrogram(torch.from_numpy(wav)).numpy()
        assert mel.shape == (80, 1 + len(wav) // 160)
        ref = g[name + "__mel"]
        assert np.abs(mel - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-7
        f64 = omel.mel_spectrogram_f64(wav)
        assert np.abs(mel - f64).max() <= 2e-5 * np.abs(f64).max()


def test_mel_known_answers():
    # 1 kHz sine: bin 25 of 201 (1000 / 40 Hz); the energy lands in the mel band around 1 kHz
    sr = 16000
    t = np.arange(8000) / sr
    mel = omel.mel_spectrogram(torch.from_numpy(0.5 * np.sin(2 * np.pi * 1000 * t)).float()).numpy()
    fb = omel.melscale_fbanks_htk().numpy()
    peak_mel = int(np.argmax(fb[25]))
    assert int(np.argmax(mel[:, 20])) == peak_mel
    # masks: width floor(u1*param), start floor(u2*(size - u1*param))
    assert omel.mask_params(0.5, 0.5, 15, 80) == (36, 7)
    assert omel.mask_params(0.0, 0.99, 35, 100) == (99, 0)


def test_collate_contract():
    g = torch.Generator().manual_seed(3)
    wavs = [torch.randn(1, n, generator=g) * 0.1 for n in (1600, 2000, 1000)]
    out = omel.collate(wavs)
    assert tuple(out.shape) == (3, 1, 80, 13)
    assert float(out[2, 0, :, 7:].abs().max()) == 0.0   # zero padded tail (T_2 = 7)
    m = omel.collate(wavs, masks=[(10, 5, 2, 3)] * 3)
    assert float(m[0, 0, 10:15].abs().max()) == 0.0 and float(m[0, 0, :, 2:5].abs().max()) == 0.0


def test_sentencepiece_vocab_round_trip(tmp_path):
    """text.SentencePieceVocab (the subword vocabulary option; scripts/sentence_piece.py:36-38 trains a BPE model
    the reference never consumes): a tiny model trained here with the installed sentencepiece; ids are shifted by
    one so that 0 stays the CTC blank, and encode -> decode returns the text."""
    spm = pytest.importorskip("sentencepiece")
    from speech_amd import text
    rs = np.random.RandomState(0)
    words = ["hello", "world", "speech", "model", "train", "frame", "audio", "signal", "mel", "ctc", "loss", "wave"]
    lines = [" ".join(rs.choice(words, rs.randint(3, 9))) for _ in range(400)]
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join(lines) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "sp"), vocab_size=60, model_type="bpe",
                                   minloglevel=2)
    vocab = text.SentencePieceVocab(str(tmp_path / "sp.model"))
    assert vocab.size == 61
    for line in lines[:20] + ["hello world", "ctc loss"]:
        ids = vocab.text_to_int(line)
        assert ids and min(ids) >= 1 and max(ids) < vocab.size          # 0 is never produced: it is the blank
        assert vocab.int_to_text(ids) == line
