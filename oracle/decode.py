"""Oracle restatement of greedy CTC decode and CER/WER (reference decoder.py:7-57).

Test infrastructure only.  Pure Python loops on purpose: this is the reference's own
algorithmic shape (argmax on the CPU, then a double loop over B x T).
"""
import numpy as np
import torch

from . import text


def argmax_ids(output):
    """decoder.py:29 / :46 -- torch.argmax(dim=2): ties -> lowest index, NaN counts as max."""
    return torch.argmax(torch.as_tensor(output), dim=2).numpy()


def collapse(args, blank_label=0):
    """decoder.py:35-40 -- drop blanks, drop a frame equal to its predecessor frame."""
    out = []
    for j, index in enumerate(args):
        if index != blank_label:
            if j != 0 and index == args[j - 1]:
                continue
            out.append(int(index))
    return out


def greedy_decode_ids(output, blank_label=0):
    return [collapse(a, blank_label) for a in argmax_ids(output)]


def greedy_decode(output, blank_label=0):
    """decoder.py:45-57."""
    return [text.int_to_text(ids) for ids in greedy_decode_ids(output, blank_label)]


def greedy_decoder(output, labels, label_lengths, blank_label=0):
    """decoder.py:28-42."""
    decodes = greedy_decode(output, blank_label)
    targets = [
        text.int_to_text(np.asarray(labels[i])[: int(label_lengths[i])].tolist())
        for i in range(len(decodes))
    ]
    return decodes, targets


def levenshtein(a, b):
    """Unit-cost edit distance (what Levenshtein.distance returns, decoder.py:17,:25)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def wer(s1, s2):
    """decoder.py:7-17 -- reference first, hypothesis second; ZeroDivisionError on empty ref."""
    w1, w2 = s1.split(), s2.split()
    return levenshtein(w1, w2) / len(w1)


def cer(s1, s2):
    """decoder.py:20-25 -- spaces removed from both strings."""
    s1, s2 = s1.replace(" ", ""), s2.replace(" ", "")
    return levenshtein(s1, s2) / len(s1)
