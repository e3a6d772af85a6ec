"""Greedy CTC decoding and CER/WER -- same module surface as the reference's decoder.py:7-57.

The argmax + blank/repeat collapse runs in the HIP kernel (sp_greedy_decode); only the compacted
id lists come back to the host.  Edit distances run in the library's native host routine.

Beam search with an n-gram language model (scripts/eval_dataset.py:20-27, :43-51) keeps pyctcdecode's
surface -- ``build_ctcdecoder(labels, kenlm_model_path, alpha=, beta=)``, ``decode``, ``decode_beams``,
``decode_batch(pool, logits)`` -- over sp_beam_candidates (GPU) and sp_ctc_beam_decode (native host
threads).  Parity unpinned: pyctcdecode / kenlm are not installed here.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops, text


def _decode_ids(output, blank_label=0, frame_lengths=None):
    if not torch.is_tensor(output):
        output = torch.as_tensor(output)
    if not output.is_cuda:
        output = output.cuda()          # the reference hands over CPU tensors (train.py:125)
    if output.dtype == torch.float64:
        output = output.float()
    ids, lengths = ops.greedy_decode_ids(output, blank=blank_label, frame_lengths=frame_lengths)
    ids, lengths = ids.cpu(), lengths.cpu().tolist()
    return [ids[b, :n].tolist() for b, n in enumerate(lengths)]


def greedy_decode_ids(output, blank_label=0, frame_lengths=None):
    """[B,T,V] scores -> list of id lists (bit-exact with torch.argmax + the reference loop)."""
    return _decode_ids(output, blank_label, frame_lengths)


def greedy_decoder(output, labels, label_lengths, blank_label=0, vocab=text):
    """decoder.py:28-42: (decoded strings, target strings)."""
    decodes = [vocab.int_to_text(ids) for ids in _decode_ids(output, blank_label)]
    labels = torch.as_tensor(labels).cpu()
    label_lengths = torch.as_tensor(label_lengths).cpu().tolist()
    targets = [vocab.int_to_text(labels[i][: label_lengths[i]].tolist()) for i in range(len(decodes))]
    return decodes, targets


def greedy_decode(output, blank_label=0, vocab=text):
    """decoder.py:45-57."""
    return [vocab.int_to_text(ids) for ids in _decode_ids(output, blank_label)]


def wer(s1, s2):
    """decoder.py:7-17: word edit distance / number of reference words (s1 = reference)."""
    w1, w2 = s1.split(), s2.split()
    table = {}
    a = [table.setdefault(w, len(table)) for w in w1]
    b = [table.setdefault(w, len(table)) for w in w2]
    return ops.edit_distance(a, b) / len(w1)


def cer(s1, s2):
    """decoder.py:20-25: spaces removed, character edit distance / reference length."""
    s1, s2 = s1.replace(" ", ""), s2.replace(" ", "")
    return ops.edit_distance([ord(c) for c in s1], [ord(c) for c in s2]) / len(s1)


# ------------------------------------------------------------------------------------------
# Beam search + n-gram language model (scripts/eval_dataset.py:20-27, :43-51)
# ------------------------------------------------------------------------------------------
DEFAULT_ALPHA, DEFAULT_BETA, DEFAULT_UNK_LOGP_OFFSET = 0.5, 1.5, -10.0      # pyctcdecode's constants
DEFAULT_BEAM_WIDTH, DEFAULT_PRUNE_LOGP, DEFAULT_MIN_TOKEN_LOGP = 100, -10.0, -5.0


class NgramLM:
    """An ARPA n-gram model held by the native library (what the reference hands to KenLM)."""

    def __init__(self, arpa_path, alpha=DEFAULT_ALPHA, beta=DEFAULT_BETA, unk_score_offset=DEFAULT_UNK_LOGP_OFFSET,
                 score_boundary=True):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().sp_ngram_load_arpa(str(arpa_path).encode(), float(alpha), float(beta),
                                                 float(unk_score_offset), int(bool(score_boundary)), ctypes.byref(h)),
                   "sp_ngram_load_arpa")
        self.handle = h
        self.alpha, self.beta, self.unk_score_offset, self.score_boundary = alpha, beta, unk_score_offset, score_boundary

    @property
    def order(self):
        return _lib.lib().sp_ngram_order(self.handle)

    def score_sentence(self, words, bos=True, eos=True):
        """log10 p(words) under the model (no alpha / beta)."""
        arr = (ctypes.c_char_p * max(len(words), 1))(*[w.encode() for w in words])
        out = ctypes.c_double()
        _lib.check(_lib.lib().sp_ngram_score_sentence(self.handle, arr, len(words), int(bos), int(eos), ctypes.byref(out)),
                   "sp_ngram_score_sentence")
        return out.value

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _lib.lib().sp_ngram_free(h)
            except Exception:
                pass


class BeamSearchDecoderCTC:
    """pyctcdecode.BeamSearchDecoderCTC for single-character vocabularies such as text.CHARS."""

    def __init__(self, labels, language_model=None):
        labels = list(labels)
        if "" not in labels and "_" in labels:          # pyctcdecode's alphabet normalisation
            labels[labels.index("_")] = ""
        if " " not in labels and "|" in labels:
            labels[labels.index("|")] = " "
        if "" not in labels:
            raise ValueError("the vocabulary needs a blank token ('' or '_')")
        if any(len(c) > 1 for c in labels):
            raise ValueError("only single-character vocabularies are supported (the reference passes text.CHARS)")
        self.labels = labels
        self.blank = labels.index("")
        self.space = labels.index(" ") if " " in labels else -1
        if language_model is not None and self.space < 0:
            raise ValueError("a language model needs a space token in the vocabulary")
        self.codepoints = np.array([ord(c) if c else 0 for c in labels], dtype=np.int32)
        self.lm = language_model

    def _search(self, logits, beam_width, beam_prune_logp, token_min_logp, n_best, frame_lengths=None, n_threads=16):
        if isinstance(logits, (list, tuple)):           # ragged utterances: pad and carry the lengths
            frame_lengths = [len(x) for x in logits]
            T = max(frame_lengths)
            pad = np.zeros((len(logits), T, len(self.labels)), dtype=np.float32)
            for i, x in enumerate(logits):
                pad[i, : len(x)] = np.asarray(x, dtype=np.float32)
            logits = pad
        x = torch.as_tensor(logits)
        if x.dim() == 2:
            x = x[None]
        if x.shape[-1] != len(self.labels):
            raise ValueError(f"logits have {x.shape[-1]} classes, the vocabulary {len(self.labels)}")
        if not x.is_cuda:
            x = x.cuda()
        if x.dtype == torch.float64:
            x = x.float()
        cn, ci, cl = ops.beam_candidates(x, token_min_logp, frame_lengths)
        fl = None if frame_lengths is None else torch.as_tensor(frame_lengths).cpu().numpy()
        ids, lens, sc = ops.ctc_beam_decode(cn.cpu().numpy(), ci.cpu().numpy(), cl.cpu().numpy(), self.codepoints,
                                            self.blank, self.space, self.lm.handle if self.lm else None, beam_width,
                                            beam_prune_logp, fl, n_threads, n_best)
        out = []
        for b in range(ids.shape[0]):
            beams = []
            for k in range(n_best):
                if lens[b, k] < 0:
                    break
                txt = "".join(self.labels[i] for i in ids[b, k, : lens[b, k]])
                beams.append((txt, float(sc[b, k, 0]), float(sc[b, k, 1])))
            out.append(beams)
        return out

    def decode_beams(self, logits, beam_width=DEFAULT_BEAM_WIDTH, beam_prune_logp=DEFAULT_PRUNE_LOGP,
                     token_min_logp=DEFAULT_MIN_TOKEN_LOGP, n_best=None):
        """[T,V] logits -> [(text, logit_score, logit_score + lm_score)], best first."""
        return self._search(logits, beam_width, beam_prune_logp, token_min_logp, n_best or beam_width)[0]

    def decode(self, logits, beam_width=DEFAULT_BEAM_WIDTH, beam_prune_logp=DEFAULT_PRUNE_LOGP,
               token_min_logp=DEFAULT_MIN_TOKEN_LOGP):
        beams = self._search(logits, beam_width, beam_prune_logp, token_min_logp, 1)[0]
        return beams[0][0] if beams else ""         # no hypothesis survives only when the scores are not finite

    def decode_batch(self, pool, logits_list, beam_width=DEFAULT_BEAM_WIDTH, beam_prune_logp=DEFAULT_PRUNE_LOGP,
                     token_min_logp=DEFAULT_MIN_TOKEN_LOGP, frame_lengths=None):
        """[B,T,V] logits (device tensor, CPU tensor, array or list of [T_b,V] arrays) -> list of texts.
        `pool` is accepted for the reference's call shape; the work runs on native threads -- as many as the
        pool has workers when one is given, 16 otherwise (eval_dataset.py:47 uses Pool(16))."""
        n_threads = getattr(pool, "_processes", None) or 16
        res = self._search(logits_list, beam_width, beam_prune_logp, token_min_logp, 1, frame_lengths, n_threads)
        return [beams[0][0] if beams else "" for beams in res]


def build_ctcdecoder(labels, kenlm_model_path=None, unigrams=None, alpha=DEFAULT_ALPHA, beta=DEFAULT_BETA,
                     unk_score_offset=DEFAULT_UNK_LOGP_OFFSET, lm_score_boundary=True):
    """pyctcdecode.build_ctcdecoder.  `kenlm_model_path` must be an ARPA text file (KenLM's binary and the
    Coqui ".scorer" container the reference points at are not read); the unigram list comes from it, as
    pyctcdecode does for .arpa files, so `unigrams` is accepted and ignored."""
    lm = None
    if kenlm_model_path is not None:
        lm = NgramLM(kenlm_model_path, alpha, beta, unk_score_offset, lm_score_boundary)
    return BeamSearchDecoderCTC(labels, lm)
