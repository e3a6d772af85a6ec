"""Bulk pseudo-label filtering -- same surface as the reference's scripts/tv/filter_alignments.py
(``load_model``, ``MetaTV``, ``collate_fn``, ``get_loader``, ``filter_files``): a throughput inference
workload (batch 256, clips up to 6 s) that keeps a manifest entry when the greedy transcript's first
and last word agree with the stored one.  The whole batch is featurised by one launch of the fused
frontend (16-bit WAV goes in as raw int16), decoded on the GPU, and only the id lists come back.

    python -m speech_amd.scripts.filter_alignments model.pth dataset_dir [dataset_dir ...]
"""
import os
import pickle
import sys

import torch

from .. import data, decoder, net


def load_model(model_path, compute_dtype=torch.bfloat16):
    """filter_alignments.py:17-38: the 10 x 512 SRModel, eval mode."""
    model = net.SRModel(10, 512, 28, data.N_MELS, 0.1, 0, compute_dtype=compute_dtype).cuda()
    if model_path is not None:
        model.load_checkpoint(torch.load(model_path, map_location="cpu", weights_only=True))
    model.eval()
    return model


class MetaTV(data.SortedDataset):
    """filter_alignments.py:41-65: manifest tuples (path, score, ..., duration) sorted by their last field,
    entries with score > 0.3 dropped; a clip longer than 6 s is replaced by one second of silence."""

    def __init__(self, manifest_path, batch_size, root_map=(("/data", "/hd1"),)):
        super().__init__(batch_size)
        if manifest_path.endswith(".json"):
            import json
            with open(manifest_path) as f:
                entries = [tuple(m) for m in json.load(f)]
        else:
            with open(manifest_path, "rb") as f:
                entries = [tuple(m) for m in data._ManifestUnpickler(f).load()]
        self.paths = [m for m in sorted(entries, key=lambda t: t[-1]) if m[1] <= 0.3]
        self.root_map = tuple(root_map)

    def get_clip(self, i):
        m = self.paths[i]
        audio_path = data._remap(m[0], self.root_map)
        transcript_path = audio_path[:-4] + ".txt" if audio_path.endswith(".wav") else audio_path + ".txt"
        waveform, _ = data.load_audio(audio_path)
        if waveform.size(1) / 16000 > 6.0:
            waveform = torch.zeros(1, 16000, dtype=waveform.dtype)
        transcript = None
        if os.path.exists(transcript_path):
            with open(transcript_path) as f:
                transcript = f.read().strip() or None
        return waveform, transcript, m


def collate_fn(collated):
    """filter_alignments.py:68-79, the spectrograms as one device tensor [B, 1, 80, Tmax]."""
    spectrograms = data.featurize([w for (w, _, _) in collated])
    return spectrograms, [t for (_, t, _) in collated], [m for (_, _, m) in collated]


def get_loader(manifest_path, batch_size=256, root_map=(("/data", "/hd1"),)):
    """filter_alignments.py:82-92.  The featuriser runs on the GPU, so the loader stays in this process
    (no worker processes may touch the device)."""
    dataset = MetaTV(manifest_path, batch_size, root_map)
    return torch.utils.data.DataLoader(dataset=dataset, batch_size=None, collate_fn=collate_fn, shuffle=True)


def keep_entry(pred, transcript):
    """filter_alignments.py:107-118: at least two words on both sides, first and last word equal."""
    if transcript is None:
        return False
    pred_w, target_w = pred.split(" "), transcript.split(" ")
    return len(target_w) >= 2 and len(pred_w) >= 2 and pred_w[0] == target_w[0] and pred_w[-1] == target_w[-1]


def filter_files(model, loader):
    filtered_manifest = []
    with torch.no_grad():
        for spectrograms, transcripts, manifests in loader:
            output = model(spectrograms.cuda())          # [B, T', n_vocab+1]; argmax is invariant to log_softmax
            preds = decoder.greedy_decode(output)
            filtered_manifest.extend(m for p, t, m in zip(preds, transcripts, manifests) if keep_entry(p, t))
    return filtered_manifest


if __name__ == "__main__":
    model = load_model(sys.argv[1])
    for dataset in sys.argv[2:]:
        print("Processing dataset", os.path.basename(dataset.rstrip("/")))
        kept = filter_files(model, get_loader(os.path.join(dataset, "manifest.pkl")))
        with open(os.path.join(dataset, "filtered_manifest.pkl"), "wb") as f:
            pickle.dump(kept, f)
