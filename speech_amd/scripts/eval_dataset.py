"""Batched inference and CER/WER -- same surface as the reference's scripts/eval_dataset.py:19-129
(``test(dataset_type, batch_size, model, test_loader, beam_decode)``).  With ``beam_decode`` the logits go
through ``decoder.build_ctcdecoder(list(text.CHARS), <ARPA file>, alpha=0.3, beta=1.0).decode_batch`` as in
eval_dataset.py:20-27 and :43-51 (candidate lists on the GPU, prefix search on native host threads).

    python -m speech_amd.scripts.eval_dataset False synthetic [model.pth]
    python -m speech_amd.scripts.eval_dataset True synthetic model.pth lm.arpa
"""
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


if __name__ == "__main__":
    beam_decode = sys.argv[1] == "True"
    dataset_type = sys.argv[2]
    model_file = sys.argv[3] if len(sys.argv) > 3 else None
    hparams = {"batch_size": 64, "n_rnn_layers": 10, "rnn_dim": 512, "dropout": 0.1, "n_vocab": 28,
               "n_feats": data.N_MELS, "projection_size": 0}
    model = net.SRModel(hparams["n_rnn_layers"], hparams["rnn_dim"], hparams["n_vocab"], hparams["n_feats"],
                        hparams["dropout"], hparams["projection_size"], compute_dtype=torch.bfloat16).cuda()
    if model_file:
        model.load_checkpoint(torch.load(model_file, map_location="cpu", weights_only=True))
    if dataset_type != "synthetic":
        raise SystemExit("only the synthetic corpus is available here (the reference's manifests point at /hd1)")
    ds = data.SyntheticSorted(256, hparams["batch_size"], max_s=8.0, seed=3)
    loader = [data.collate_fn(ds[i], "valid") for i in range(len(ds))]
    test(dataset_type, hparams["batch_size"], model, loader, beam_decode, lm_path=sys.argv[4] if len(sys.argv) > 4 else None)
