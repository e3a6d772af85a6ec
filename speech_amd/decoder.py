"""Greedy CTC decoding and CER/WER -- same module surface as the reference's decoder.py:7-57.

The argmax + blank/repeat collapse runs in the HIP kernel (sp_greedy_decode); only the compacted
id lists come back to the host.  Edit distances run in the library's native host routine.
"""
import torch

from . import ops, text


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
