"""Character vocabulary of the acoustic model -- same module surface as the reference's
text.py:1-12 (CHARS, text_to_int, int_to_text).  Index 0 ('_') is the CTC blank
(train.py:266 uses blank=0; train.py:343 says n_vocab=28 excludes it).

A SentencePiece vocabulary (scripts/sentence_piece.py trains a 999-piece BPE model that the
reference never consumes) is available through ``SentencePieceVocab`` with the same two
functions; id 0 stays the blank and pieces are shifted by one.
"""

CHARS = "_' abcdefghijklmnopqrstuvwxyz"

char_to_i = {}
i_to_char = {}
for _i, _c in enumerate(CHARS):
    char_to_i[_c] = _i
    i_to_char[_i] = _c


def text_to_int(text):
    """KeyError on any character outside CHARS, like the reference."""
    return [char_to_i[c] for c in text]


def int_to_text(labels):
    return "".join(i_to_char[int(i)] for i in labels)


class SentencePieceVocab:
    """Optional subword vocabulary: V = piece_count + 1 with the blank at id 0."""

    def __init__(self, model_file):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=model_file)
        self.size = self.sp.get_piece_size() + 1

    def text_to_int(self, text):
        return [i + 1 for i in self.sp.encode(text)]

    def int_to_text(self, labels):
        return self.sp.decode([int(i) - 1 for i in labels])
