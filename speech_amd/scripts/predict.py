"""Single-file inference -- same surface as the reference's scripts/predict.py:16-47
(``get_model()``, ``predict(audio_path, model)``), with the model built the way
scripts/eval_dataset.py:85-92 builds it (predict.py's own constructor call is stale).

    python -m speech_amd.scripts.predict model.pth clip.wav
"""
import sys
import wave

import numpy as np
import torch

from .. import data, decoder, net

HPARAMS = {"n_rnn_layers": 10, "rnn_dim": 512, "dropout": 0.1, "n_vocab": 28,
           "n_feats": data.N_MELS, "projection_size": 0}


def get_model(model_file=None, device="cuda", dtype=torch.float32):
    model = net.SRModel(HPARAMS["n_rnn_layers"], HPARAMS["rnn_dim"], HPARAMS["n_vocab"], HPARAMS["n_feats"],
                        HPARAMS["dropout"], HPARAMS["projection_size"], compute_dtype=dtype)
    if model_file:
        # both checkpoint shapes of the reference (train.py:171-178 dict, bare state_dict)
        model.load_checkpoint(torch.load(model_file, map_location="cpu", weights_only=True))
    model.to(device)
    model.eval()
    return model


def load_wav(audio_path):
    """16-bit PCM wav -> float32 [1, N] in [-1, 1] (torchaudio.load(normalize=True) semantics)."""
    with wave.open(audio_path, "rb") as f:
        if f.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM wav is supported")
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
        if f.getnchannels() > 1:
            pcm = pcm.reshape(-1, f.getnchannels())[:, 0]
        if f.getframerate() != data.SAMPLE_RATE:
            raise ValueError(f"expected {data.SAMPLE_RATE} Hz audio, got {f.getframerate()}")
    return torch.from_numpy(pcm.astype(np.float32) / 32768.0).unsqueeze(0)


@torch.no_grad()
def predict(audio_path, model):
    wav = load_wav(audio_path) if isinstance(audio_path, str) else audio_path
    spect = data.spectrogram_transform(wav).unsqueeze(0)     # [1, 1, 80, T]
    output = model(spect).float()
    # the reference applies log_softmax before the argmax (predict.py:45); argmax is invariant to it
    pred = decoder.greedy_decode(output)
    print(pred)
    return pred


if __name__ == "__main__":
    predict(sys.argv[2], get_model(sys.argv[1]))
