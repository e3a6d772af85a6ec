"""Oracle restatement of the vocabulary (reference text.py:1-12).  Test infrastructure."""

# reference text.py:1 -- 29 symbols, index 0 ('_') is the CTC blank (train.py:266, :343)
CHARS = "_' abcdefghijklmnopqrstuvwxyz"

_C2I = {c: i for i, c in enumerate(CHARS)}
_I2C = {i: c for i, c in enumerate(CHARS)}


def text_to_int(text):
    """reference text.py:7-8 -- KeyError on any character outside CHARS."""
    return [_C2I[c] for c in text]


def int_to_text(labels):
    """reference text.py:11-12."""
    return "".join(_I2C[int(i)] for i in labels)
