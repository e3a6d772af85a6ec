"""Dataset / IO path of speech_amd.data (the reference's data.py:30-157): manifests, transcript lookup,
SortedDataset family -- on temporary files, no GPU.  The audio is 16-bit PCM WAV written with the
standard library; load_audio hands it on as raw int16 (the GPU featuriser scales it, tests/test_gpu_*)."""
import json
import os
import pickle
import wave

import numpy as np
import pytest
import torch


def _write_wav(path, pcm, sr=16000):
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(np.asarray(pcm, dtype="<i2").tobytes())


@pytest.fixture
def corpus(tmp_path):
    rs = np.random.RandomState(0)
    libri = tmp_path / "data" / "LibriSpeech" / "train-clean" / "19" / "198"
    libri.mkdir(parents=True)
    entries, lines = [], []
    for k, n in enumerate([4000, 5000, 6000, 7000]):
        fid = f"19-198-{k:04d}"
        _write_wav(libri / (fid + ".wav"), rs.randint(-3000, 3000, n))
        lines.append(f"{fid} HELLO WORLD {k}")
        entries.append((str(libri / (fid + ".wav")), n))
    (libri / "19-198.trans.txt").write_text("\n".join(lines) + "\n")
    tv = tmp_path / "hd1" / "clean"
    tv.mkdir(parents=True)
    tv_entries = []
    for k, n in enumerate([4500, 6500]):
        p = tv / f"clip{k}.wav"
        _write_wav(p, rs.randint(-3000, 3000, n))
        (tv / f"clip{k}.txt").write_text(f"tv clip {k}\n")
        tv_entries.append((str(p), n))
    return tmp_path, entries, tv_entries


def test_manifest_formats_and_restricted_unpickler(corpus, tmp_path):
    from speech_amd import data
    _, entries, _ = corpus
    pkl, js = tmp_path / "m.pkl", tmp_path / "m.json"
    pkl.write_bytes(pickle.dumps(entries))
    js.write_text(json.dumps(entries))
    assert data.load_manifest(str(pkl)) == data.load_manifest(str(js)) == entries
    assert data.get_librispeech_paths(str(pkl))[0] == (entries[0][0], 4000 / 16000)
    evil = tmp_path / "evil.pkl"
    evil.write_bytes(pickle.dumps([(os.path.join, 1)]))            # anything but plain containers is refused
    with pytest.raises(pickle.UnpicklingError):
        data.load_manifest(str(evil))


def test_clips_and_sorted_datasets(corpus, tmp_path):
    from speech_amd import data
    root, entries, tv_entries = corpus
    w, u = data.get_librispeech_clip(entries[2][0])
    assert w.dtype == torch.int16 and tuple(w.shape) == (1, 6000) and u == "hello world 2"
    w, u = data.get_tv_clip(tv_entries[1][0])
    assert tuple(w.shape) == (1, 6500) and u == "tv clip 1"
    with pytest.raises(KeyError):
        os.rename(entries[0][0], entries[0][0].replace("0000", "0099"))
        data.get_librispeech_clip(entries[0][0].replace("0000", "0099"))
    os.rename(entries[0][0].replace("0000", "0099"), entries[0][0])

    lib_m = tmp_path / "libri_dev.pkl"
    lib_m.write_bytes(pickle.dumps(entries))
    ds = data.SortedLibriSpeech(str(lib_m), 2, root_map=())
    assert len(ds) == 2 and [u for _, u in ds[1]] == ["hello world 2", "hello world 3"]
    # the reference's path rewriting: "/data" -> "/hd1" (here: to where the files really are)
    moved = [(p.replace(str(root / "data"), "/data"), n) for p, n in entries]
    lib_m.write_bytes(pickle.dumps(moved))
    ds = data.SortedLibriSpeech(str(lib_m), 4, root_map=(("/data", str(root / "data")),))
    assert len(ds[0]) == 4
    train_m = tmp_path / "libri_train.pkl"
    train_m.write_bytes(pickle.dumps(entries * 300))
    assert len(data.SortedLibriSpeech(str(train_m), 10).paths) == 200          # the 1000 longest are dropped

    tv_m = tmp_path / "tv.pkl"
    tv_m.write_bytes(pickle.dumps(tv_entries))
    tv = data.SortedTV([str(tv_m)], 2, 0, root_map=())
    assert len(tv) == 1 and tv[0][0][1] == "tv clip 0"
    big = tmp_path / "libri_big.pkl"
    big.write_bytes(pickle.dumps(entries + [(entries[-1][0], 10 ** 9)] * 2000))
    # (the reference divides only the LibriSpeech manifest by 16000: TV manifests that are merged carry seconds)
    tv_s = tmp_path / "tv_seconds.pkl"
    tv_s.write_bytes(pickle.dumps([(p, n / 16000) for p, n in tv_entries]))
    comb = data.CombinedTVLibriSpeech(str(big), [str(tv_s)], 3, 0, root_map=())
    assert len(comb) == 2
    durs = [w.shape[1] for w, _ in comb[0] + comb[1]]
    assert durs == sorted(durs) == [4000, 4500, 5000, 6000, 6500, 7000]        # merged and re-sorted by duration
    with pytest.raises(ValueError):
        data.load_audio  # (8-bit files are refused)
        p8 = tmp_path / "eight.wav"
        with wave.open(str(p8), "wb") as f:
            f.setnchannels(1); f.setsampwidth(1); f.setframerate(16000); f.writeframes(b"\x00" * 100)
        data.load_audio(str(p8))
