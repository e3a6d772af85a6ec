"""CPU oracle for CTC beam search with an n-gram language model -- TEST INFRASTRUCTURE ONLY.

Restates what the reference's evaluation path asks of pyctcdecode (scripts/eval_dataset.py:20-27:
``build_ctcdecoder(list(text.CHARS), <kenlm scorer>, alpha=0.3, beta=1.0)``; :43-51
``decode_batch(pool, logits)``).  PARITY UNPINNED: pyctcdecode and kenlm are third-party packages that
are neither vendored in the reference nor installed here, and the reference holds no decoded fixture.
Three independent pieces, in plain Python over strings:

* ``ArpaLM``        -- ARPA reader and back-off scoring (what KenLM computes for the same file).
* ``exhaustive``    -- enumerates all V^T alignments of a tiny lattice, collapses them with the CTC rule,
                       normalises white space, sums the probability per text and adds the LM score:
                       the exact answer an unpruned beam search must reach.
* ``prefix_beam``   -- pyctcdecode's published per-frame algorithm (beams keyed by text, open word and
                       last token; merge by log-sum-exp; LM score per finished word, penalty for an open
                       word that starts no unigram; prune by score gap and beam width), for lattices too
                       large to enumerate.
"""
import itertools
import math

import numpy as np

LN10 = math.log(10.0)
LOG_MIN_P = math.log(1e-15)


def log_softmax_clip(logits):
    x = np.asarray(logits, dtype=np.float64)
    x = x - x.max(axis=-1, keepdims=True)
    lp = x - np.log(np.exp(x).sum(axis=-1, keepdims=True))
    return np.clip(lp, LOG_MIN_P, 0.0)


def candidates(logits, token_min_logp):
    """Per frame the (id, log p) pairs a search expands: log p >= threshold, plus the argmax."""
    lp = log_softmax_clip(logits)
    out = []
    for row in lp:
        keep = set(np.nonzero(row >= token_min_logp)[0].tolist()) | {int(row.argmax())}
        out.append([(i, float(row[i])) for i in sorted(keep)])
    return out


class ArpaLM:
    def __init__(self, path, alpha=0.5, beta=1.5, unk_score_offset=-10.0, score_boundary=True):
        self.alpha, self.beta, self.unk_score_offset, self.score_boundary = alpha, beta, unk_score_offset, score_boundary
        self.grams = {}          # tuple of words -> (log10 p, log10 backoff)
        self.order = 0
        n = 0
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line == "\\data\\" or line.startswith("ngram "):
                    continue
                if line == "\\end\\":
                    break
                if line.startswith("\\"):
                    n = int(line[1:].split("-")[0])
                    self.order = max(self.order, n)
                    continue
                tok = line.split()
                words = tuple(tok[1:1 + n])
                bo = float(tok[1 + n]) if len(tok) > 1 + n else 0.0
                self.grams[words] = (float(tok[0]), bo)
        self.unigrams = {w[0] for w in self.grams if len(w) == 1} - {"<unk>", "<s>", "</s>"}

    def start(self):
        return ("<s>",) if self.score_boundary else ()

    def raw(self, ctx, word):
        """(log10 p(word | ctx), next context)."""
        w = word if (word,) in self.grams else "<unk>"
        lp, hist = 0.0, tuple(ctx)
        while True:
            g = hist + (w,)
            if g in self.grams:
                lp += self.grams[g][0]
                break
            if not hist:
                lp = -100.0
                break
            lp += self.grams.get(hist, (0.0, 0.0))[1]
            hist = hist[1:]
        nxt = (tuple(ctx) + (w,))[-(self.order - 1):] if self.order > 1 else ()
        return lp, nxt

    def word(self, ctx, word):
        """Weighted score of one finished word, pyctcdecode's LanguageModel.score without the end."""
        lp, nxt = self.raw(ctx, word)
        if word not in self.unigrams:
            lp += self.unk_score_offset
        return self.alpha * lp * LN10 + self.beta, nxt

    def end(self, ctx):
        return self.alpha * self.raw(ctx, "</s>")[0] * LN10 if self.score_boundary else 0.0

    def text(self, words):
        ctx, total = self.start(), 0.0
        for w in words:
            s, ctx = self.word(ctx, w)
            total += s
        return total + self.end(ctx)

    def partial(self, part):
        pen = 0.0 if any(u.startswith(part) for u in self.unigrams) else self.unk_score_offset
        if len(part) > 6:
            pen = pen * len(part) / 6.0
        return pen


def _logaddexp(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    return max(a, b) + math.log1p(math.exp(-abs(a - b)))


def exhaustive(logits, labels, blank, lm=None):
    """{text: (logit score, combined score)} over ALL alignments; tiny T and V only."""
    lp = log_softmax_clip(logits)
    T, V = lp.shape
    assert V ** T <= 2_000_000, "lattice too large to enumerate"
    acc = {}
    for path in itertools.product(range(V), repeat=T):
        s = sum(lp[t, c] for t, c in enumerate(path))
        chars, prev = [], None
        for c in path:
            if c != blank and c != prev:
                chars.append(labels[c])
            prev = c
        txt = " ".join("".join(chars).split())
        acc[txt] = _logaddexp(acc.get(txt, -math.inf), s)
    return {t: (s, s + (lm.text(t.split()) if lm else 0.0)) for t, s in acc.items()}


def prefix_beam(logits, labels, blank, lm=None, beam_width=100, beam_prune_logp=-10.0, token_min_logp=-5.0):
    """[(text, logit score, combined score)], best first."""
    space = labels.index(" ") if " " in labels else -1
    # beam key: (finished words, open word, last token id); value: logit score
    beams = {((), "", None): 0.0}
    lm_cache = {(): (0.0, lm.start())} if lm else None

    def lm_words(words):
        if lm is None:
            return 0.0
        if words not in lm_cache:
            base = lm_words(words[:-1])
            s, nxt = lm.word(lm_cache[words[:-1]][1], words[-1])
            lm_cache[words] = (base + s, nxt)
        return lm_cache[words][0]

    def trim(scored):
        best = max(s for _, _, s in scored)
        scored = [x for x in scored if x[2] >= best + beam_prune_logp]
        scored.sort(key=lambda x: -x[2])
        return scored[:beam_width]

    for frame in candidates(logits, token_min_logp):
        new = {}
        for c, p in frame:
            for (words, part, last), score in beams.items():
                if c == blank or c == last:
                    key = (words, part, c)
                elif c == space:
                    key = (words + (part,), "", c) if part else (words, "", c)
                else:
                    key = (words, part + labels[c], c)
                new[key] = _logaddexp(new.get(key, -math.inf), score + p)
        scored = [(k, v, v + lm_words(k[0]) + (lm.partial(k[1]) if lm and k[1] else 0.0)) for k, v in new.items()]
        beams = {k: v for k, v, _ in trim(scored)}
    final = {}
    for (words, part, _), score in beams.items():
        w = words + (part,) if part else words
        final[w] = _logaddexp(final.get(w, -math.inf), score)
    scored = []
    for w, v in final.items():
        total = v
        if lm:
            total += lm_words(w) + lm.end(lm_cache[w][1])
        scored.append((w, v, total))
    return [(" ".join(w), v, s) for w, v, s in trim(scored)]
