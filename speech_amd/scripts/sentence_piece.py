"""Subword vocabulary training -- same surface as the reference's scripts/sentence_piece.py:8-38
(``read_transcript``, ``gather_sentences(dataset_paths)``, then a 999-piece BPE model ``sp/tv``).

The reference calls ``torchtext.data.functional.generate_sp_model`` (a thin wrapper over the SentencePiece
trainer); torchtext is not installed here, so the trainer is called directly with the same arguments.  Manifests
go through ``data.load_manifest`` (an unpickler that admits plain containers only); the reference's hard-coded
``/data`` -> ``/hd1`` rewrite is the ``root_map`` argument.  The result is read by ``text.SentencePieceVocab``.

    python -m speech_amd.scripts.sentence_piece [--gather m1.pkl m2.pkl ...] [--sentences sp/sentences.txt]
                                                [--vocab_size 999] [--model_prefix sp/tv]
"""
import argparse
import os

from .. import data


def read_transcript(txt_f):
    with open(txt_f, "r") as f:
        return f.read().strip()


def gather_sentences(dataset_paths, out_path="sp/sentences.txt", root_map=(("/data", "/hd1"),)):
    """scripts/sentence_piece.py:14-27: one transcript (<clip>.txt beside <clip>.wav) per manifest entry."""
    all_paths = []
    for dataset_path in dataset_paths:
        paths = [p for p, _ in data.load_manifest(dataset_path)]
        for src, dst in root_map or ():
            paths = [p.replace(src, dst) for p in paths]
        all_paths.extend(paths)
    sentences = [read_transcript((p[:-4] if p.endswith(".wav") else p) + ".txt") for p in all_paths]
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(sentences))
    return len(sentences)


def generate_sp_model(filename, vocab_size=999, model_type="bpe", model_prefix="sp/tv"):
    """torchtext.data.functional.generate_sp_model(filename, vocab_size, model_type, model_prefix)."""
    import sentencepiece as spm
    os.makedirs(os.path.dirname(model_prefix) or ".", exist_ok=True)
    spm.SentencePieceTrainer.train(input=filename, vocab_size=vocab_size, model_type=model_type,
                                   model_prefix=model_prefix)
    return model_prefix + ".model"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gather", nargs="*", default=None, help="manifests to collect transcripts from first")
    ap.add_argument("--sentences", default="sp/sentences.txt")
    ap.add_argument("--vocab_size", type=int, default=999)
    ap.add_argument("--model_prefix", default="sp/tv")
    ap.add_argument("--root_map", default="/data=/hd1")
    a = ap.parse_args()
    if a.gather:
        gather_sentences(a.gather, a.sentences, tuple(tuple(kv.split("=", 1)) for kv in a.root_map.split(",") if kv))
    generate_sp_model(a.sentences, a.vocab_size, "bpe", a.model_prefix)
