"""Batched inference and CER/WER -- same surface as the reference's scripts/eval_dataset.py:19-129
(``test(dataset_type, batch_size, model, test_loader, beam_decode)``).  With ``beam_decode`` the logits go
through ``decoder.build_ctcdecoder(list(text.CHARS), <ARPA file>, alpha=0.3, beta=1.0).decode_batch`` as in
eval_dataset.py:20-27 and :43-51 (candidate lists on the GPU, prefix search on native host threads).

    python -m speech_amd.scripts.eval_dataset False libri model.pth --manifest_root /corpora --root_map /data=/corpora
    python -m speech_amd.scripts.eval_dataset True ibm model.pth --ibm_root /corpora/ibm --lm lm.arpa
    python -m speech_amd.scripts.eval_dataset False synthetic
The three positional arguments are the reference's (`beam_decode dataset_type model_file`, eval_dataset.py:70-73);
the datasets are built as eval_dataset.py:95-117 builds them (`libri` / `tv` / `ibm` / `eval_high_cer`), the model runs in
fp16 like its `model.half()` (:120; --dtype bf16 / fp32 to change), the loader is the reference's (shuffle off, whole
pre-sorted batches from the Sorted* sets, 64 clips per batch for the IBM directory).  What the reference hard-codes and a
user of it edits in place are options here: --manifest_root (directory the `datasets/...` manifests are relative to),
--root_map old=new[,old=new] (path prefixes rewritten in the manifests; the reference's "/data" -> "/hd1"), --ibm_root,
--lm (ARPA file standing in for the Coqui .scorer of :24, which is not in the reference tree).
"""
import argparse
import os
import sys

import torch

from .. import data, decoder, net, text


def test(dataset_type, batch_size, model, test_loader, beam_decode=False, log=print, lm_path=None,
         alpha=0.3, beta=1.0):
    ctc_decoder = None
    if beam_decode:
        # eval_dataset.py:20-27; lm_path None searches without a language model
        ctc_decoder = decoder.build_ctcdecoder(list(text.CHARS), lm_path, alpha=alpha, beta=beta)
    log("\nevaluating...")
    model.eval()
    test_cer, test_wer = [], []
    with torch.no_grad():
        for batch in test_loader:
            spectrograms, labels, label_lengths = batch
            logits = model(spectrograms.cuda())                       # [B, T', V]
            # eval_dataset.py:40 takes log_softmax first; the greedy path is invariant to it
            if ctc_decoder is not None:
                preds = ctc_decoder.decode_batch(None, logits)        # eval_dataset.py:43-51, raw logits
                targets = [text.int_to_text(labels[j][: label_lengths[j]].tolist()) for j in range(len(preds))]
            else:
                preds, targets = decoder.greedy_decoder(logits, labels, label_lengths)
            for tgt, pred in zip(targets, preds):
                test_cer.append(decoder.cer(tgt, pred))
                test_wer.append(decoder.wer(tgt, pred))
    avg_cer = sum(test_cer) / len(test_cer)
    avg_wer = sum(test_wer) / len(test_wer)
    log("Test set: Average CER: {:4f} Average WER: {:.4f}\n".format(avg_cer, avg_wer))
    return avg_cer, avg_wer


# eval_dataset.py:100-105: the evaluation twins of these training manifests
TV_TRAIN_DATASET_PATHS = [
    "datasets/first/sorted_train_cer_0.2.pkl",
    "datasets/second/sorted_train_cer_0.2.pkl",
    "datasets/third/sorted_train_cer_0.2.pkl",
    "datasets/fourth/sorted_train_cer_0.2.pkl",
]
LIBRI_TEST_PATH = "datasets/librispeech/sorted_test_clean_librispeech.pkl"
HIGH_CER_PATH = "eval_paths_0.21cer.pkl"


def build_dataset(dataset_type, batch_size, manifest_root="", root_map=None, ibm_root="/hd1/ibm"):
    """(dataset, loader batch_size) as eval_dataset.py:95-117: the Sorted* sets yield whole batches (batch_size None in
    the loader), the IBM directory single clips batched by the loader."""
    rel = lambda p: os.path.join(manifest_root, p)      # noqa: E731
    if dataset_type == "libri":
        kw = {} if root_map is None else {"root_map": root_map}
        return data.SortedLibriSpeech(rel(LIBRI_TEST_PATH), batch_size, **kw), None
    if dataset_type == "tv":
        evals = [rel(os.path.join(os.path.dirname(p), os.path.basename(p).replace("train", "eval")))
                 for p in TV_TRAIN_DATASET_PATHS]
        return data.SortedTV(evals, batch_size, 0, root_map), None
    if dataset_type == "ibm":
        return data.IBMDataset(ibm_root), batch_size
    if dataset_type == "eval_high_cer":
        return data.SortedTV([rel(HIGH_CER_PATH)], batch_size, 0, root_map), None
    if dataset_type == "synthetic":
        return data.SyntheticSorted(256, batch_size, max_s=8.0, seed=3), None
    raise SystemExit("Unkown dataset " + str(dataset_type))       # eval_dataset.py:116-117 (message as there)


def _clips(batch):
    return batch


def build_loader(dataset, loader_batch_size, num_workers=3):
    """eval_dataset.py:121-128: shuffle off; the workers decode clips, the featuriser runs on the GPU in this process
    (train.FeaturizingLoader), since forked workers must not touch it."""
    from ..train import FeaturizingLoader
    return FeaturizingLoader(torch.utils.data.DataLoader(dataset=dataset, batch_size=loader_batch_size, shuffle=False,
                                                         collate_fn=_clips, num_workers=num_workers), "valid")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("beam_decode", help="True / False (eval_dataset.py:71)")
    p.add_argument("dataset_type", choices=["libri", "tv", "ibm", "eval_high_cer", "synthetic"])
    p.add_argument("model_file", nargs="?", default=None)
    p.add_argument("--dtype", choices=["fp16", "bf16", "fp32"], default="fp16", help="activation type (the reference: model.half())")
    p.add_argument("--manifest_root", default="")
    p.add_argument("--root_map", default="")
    p.add_argument("--ibm_root", default="/hd1/ibm")
    p.add_argument("--lm", default=None, help="ARPA n-gram model for the beam search (alpha 0.3, beta 1.0)")
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--num_workers", type=int, default=3)
    p.add_argument("--n_rnn_layers", type=int, default=10)
    p.add_argument("--rnn_dim", type=int, default=512)
    p.add_argument("--projection_size", type=int, default=0)
    a = p.parse_args(argv)
    if a.beam_decode not in ("True", "False"):
        p.error("beam_decode must be True or False")
    a.beam_decode = a.beam_decode == "True"
    a.root_map = tuple(tuple(kv.split("=", 1)) for kv in a.root_map.split(",") if kv) or None
    return a


def main(argv=None, log=print):
    a = parse_args(argv)
    hparams = {"shuffle": True, "batch_size": a.batch_size, "n_rnn_layers": a.n_rnn_layers, "rnn_dim": a.rnn_dim, "dropout": 0.1,
               "n_vocab": 28, "n_feats": data.N_MELS, "projection_size": a.projection_size}      # eval_dataset.py:74-84
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
    model = net.SRModel(hparams["n_rnn_layers"], hparams["rnn_dim"], hparams["n_vocab"], hparams["n_feats"],
                        hparams["dropout"], hparams["projection_size"], compute_dtype=dtype).cuda()
    dataset, loader_bs = build_dataset(a.dataset_type, hparams["batch_size"], a.manifest_root, a.root_map, a.ibm_root)
    if a.model_file:
        model.load_checkpoint(torch.load(a.model_file, map_location="cpu", weights_only=True))
    elif a.dataset_type != "synthetic":
        raise SystemExit("model_file is required for a real corpus")
    loader = build_loader(dataset, loader_bs, 0 if a.dataset_type == "synthetic" else a.num_workers)
    return test(a.dataset_type, hparams["batch_size"], model, loader, a.beam_decode, log=log, lm_path=a.lm)


if __name__ == "__main__":
    main()
