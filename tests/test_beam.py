"""CTC beam search + n-gram LM (scripts/eval_dataset.py:20-27, :43-51): the native host search through the
C ABI against the CPU oracle.  Parity unpinned at the package level (pyctcdecode / kenlm are not installed):
what is checked is (1) an unpruned search reaches the exact optimum of an exhaustive enumeration, with and
without a language model; (2) the pruned search equals the oracle's restatement of the published algorithm
on lattices too large to enumerate; (3) the ARPA back-off scorer equals an independent Python one."""
import math

import numpy as np
import pytest

from oracle import beam as obeam
from speech_amd import _lib, decoder, ops, text

ARPA = """\\data\\
ngram 1=7
ngram 2=8
ngram 3=3

\\1-grams:
-1.2\t<unk>
-99\t<s>\t-0.40
-0.9\t</s>
-0.7\ta\t-0.30
-0.8\tab\t-0.25
-1.0\tba\t-0.20
-1.1\tabba\t-0.10

\\2-grams:
-0.5\t<s> a\t-0.15
-0.6\t<s> ab\t-0.12
-0.4\ta ba\t-0.11
-0.9\ta a
-0.3\tab a\t-0.2
-0.7\tba </s>
-0.5\ta </s>
-0.8\tba ab

\\3-grams:
-0.2\t<s> a ba
-0.35\ta ba </s>
-0.45\tab a </s>

\\end\\
"""

LABELS = ["", " ", "a", "b"]


@pytest.fixture(scope="module")
def arpa_path(tmp_path_factory):
    p = tmp_path_factory.mktemp("lm") / "tiny.arpa"
    p.write_text(ARPA)
    return str(p)


def _host_search(logits, labels, lm=None, beam_width=100, prune=-10.0, min_logp=-5.0, n_best=1):
    """Candidate lists built on the CPU by the oracle (the GPU stage has its own test), searched natively."""
    cands = [obeam.candidates(x, min_logp) for x in logits]
    B, T = len(cands), max(len(c) for c in cands)
    kmax = max(len(f) for c in cands for f in c)
    n = np.zeros((B, T), np.int32)
    ids = np.zeros((B, T, kmax), np.int32)
    lps = np.zeros((B, T, kmax), np.float32)
    for b, c in enumerate(cands):
        for t, f in enumerate(c):
            n[b, t] = len(f)
            for k, (i, p) in enumerate(f):
                ids[b, t, k], lps[b, t, k] = i, p
    cps = [ord(c) if c else 0 for c in labels]
    fl = np.array([len(c) for c in cands], np.int32)
    out_ids, out_len, sc = ops.ctc_beam_decode(n, ids, lps, cps, labels.index(""), labels.index(" ") if " " in labels else -1,
                                               lm.handle if lm else None, beam_width, prune, fl, 4, n_best)
    res = []
    for b in range(B):
        res.append([("".join(labels[i] for i in out_ids[b, k, : out_len[b, k]]), float(sc[b, k, 0]), float(sc[b, k, 1]))
                    for k in range(n_best) if out_len[b, k] >= 0])
    return res


def test_arpa_scorer_matches_python_backoff(arpa_path):
    lm = decoder.NgramLM(arpa_path, 0.3, 1.0)
    ref = obeam.ArpaLM(arpa_path, 0.3, 1.0)
    assert lm.order == 3
    for words in (["a", "ba"], ["ab", "a"], ["ba", "ab", "a", "a"], ["zzz"], ["a", "zzz", "ba"], [], ["abba", "abba"]):
        want, ctx = 0.0, ref.start()
        for w in words:
            s, ctx = ref.raw(ctx, w)
            want += s
        want += ref.raw(ctx, "</s>")[0]
        assert lm.score_sentence(words) == pytest.approx(want, abs=1e-5), words


@pytest.mark.parametrize("with_lm", [False, True])
def test_unpruned_search_reaches_the_exhaustive_optimum(arpa_path, with_lm):
    rng = np.random.default_rng(5)
    lm = decoder.NgramLM(arpa_path, 0.3, 1.0) if with_lm else None
    ref_lm = obeam.ArpaLM(arpa_path, 0.3, 1.0) if with_lm else None
    for case in range(12):
        T = int(rng.integers(1, 8))
        logits = rng.normal(size=(T, 4)) * rng.choice([0.5, 2.0, 4.0])
        table = obeam.exhaustive(logits, LABELS, 0, ref_lm)
        got = _host_search([logits], LABELS, lm, beam_width=100000, prune=-1e30, min_logp=-1e30, n_best=len(table) + 3)[0]
        assert len(got) == len(table)                      # every text reachable, each exactly once
        for txt, logit, total in got:
            assert logit == pytest.approx(table[txt][0], abs=2e-4), (case, txt)
            assert total == pytest.approx(table[txt][1], abs=2e-4), (case, txt)
        best = max(table.items(), key=lambda kv: kv[1][1])
        assert got[0][0] == best[0] or abs(table[got[0][0]][1] - best[1][1]) < 1e-5
        assert all(got[i][2] >= got[i + 1][2] for i in range(len(got) - 1))


@pytest.mark.parametrize("with_lm", [False, True])
def test_pruned_search_equals_the_restated_algorithm(arpa_path, with_lm):
    rng = np.random.default_rng(11)
    lm = decoder.NgramLM(arpa_path, 0.3, 1.0) if with_lm else None
    ref_lm = obeam.ArpaLM(arpa_path, 0.3, 1.0) if with_lm else None
    for case in range(6):
        T = int(rng.integers(20, 60))
        logits = rng.normal(size=(T, 4)) * 3.0
        for bw, prune, mlp in ((8, -10.0, -5.0), (3, -4.0, -2.0), (50, -10.0, -5.0)):
            want = obeam.prefix_beam(logits, LABELS, 0, ref_lm, bw, prune, mlp)
            got = _host_search([logits], LABELS, lm, bw, prune, mlp, n_best=bw)[0]
            assert [g[0] for g in got] == [w[0] for w in want], (case, bw)
            np.testing.assert_allclose([g[1:] for g in got], [w[1:] for w in want], atol=5e-4)


def test_chars_vocabulary_and_batch_of_ragged_utterances(arpa_path):
    """text.CHARS as the reference passes it ('_' is the blank), several utterances on worker threads."""
    labels = list(text.CHARS)
    dec = decoder.BeamSearchDecoderCTC(labels)
    assert dec.labels[dec.blank] == "" and dec.labels[dec.space] == " "
    rng = np.random.default_rng(2)
    utts = [rng.normal(size=(int(rng.integers(5, 40)), len(labels))) * 4.0 for _ in range(7)]
    got = _host_search(utts, dec.labels, None, 16, -10.0, -5.0)
    for x, g in zip(utts, got):
        want = obeam.prefix_beam(x, dec.labels, dec.blank, None, 16, -10.0, -5.0)
        assert g[0][0] == want[0][0]
        assert g[0][1] == pytest.approx(want[0][1], abs=5e-4)


def test_beam_width_one_without_lm_is_not_worse_than_greedy():
    rng = np.random.default_rng(3)
    logits = rng.normal(size=(30, 4)) * 3.0
    lp = obeam.log_softmax_clip(logits)
    greedy = float(lp.max(axis=1).sum())                   # the best single alignment
    got = _host_search([logits], LABELS, None, 1, -10.0, -5.0)[0][0]
    assert got[1] >= greedy - 1e-4


def test_bad_arguments_are_refused(arpa_path, tmp_path):
    with pytest.raises(_lib.SpeechHipError):
        decoder.NgramLM(str(tmp_path / "missing.arpa"))
    bad = tmp_path / "bad.arpa"
    bad.write_text("\\data\\\nngram 1=1\n\n\\1-grams:\n-1.0\ta\n\n\\2-grams:\n-1.0\ta zebra\n\\end\\\n")
    with pytest.raises(_lib.SpeechHipError, match="not a unigram"):
        decoder.NgramLM(str(bad))
    n = np.array([[3]], np.int32)                         # more candidates than the lists hold
    with pytest.raises(_lib.SpeechHipError, match="candidates"):
        ops.ctc_beam_decode(n, np.zeros((1, 1, 2), np.int32), np.zeros((1, 1, 2), np.float32), [0, 32, 97], 0, 1)
    with pytest.raises(ValueError):
        decoder.BeamSearchDecoderCTC(["a", "b"])          # no blank
    lm = decoder.NgramLM(arpa_path)
    with pytest.raises(ValueError):
        decoder.BeamSearchDecoderCTC(["", "a", "b"], lm)  # LM without a space token
