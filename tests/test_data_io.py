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


def test_ibm_dataset_and_rank_sampler(tmp_path):
    """IBMDataset (data.py:82-102) reads wav + txt pairs of one directory; RankBatchSampler hands every rank the
    same number of batches (a rank with one more would hang the bucket all-reduces) and all ranks agree on the
    epoch's permutation."""
    from speech_amd import data
    from speech_amd.train import RankBatchSampler
    rs = np.random.RandomState(1)
    for k, n in enumerate([3000, 3500, 4100]):
        _write_wav(tmp_path / f"utt{k}.wav", rs.randint(-2000, 2000, n))
        (tmp_path / f"utt{k}.txt").write_text(f" ibm utterance {k} \n")
    (tmp_path / "notes.md").write_text("ignored")
    ds = data.IBMDataset(str(tmp_path))
    assert len(ds) == 3
    w, u = ds[1]
    assert w.dtype == torch.int16 and tuple(w.shape) == (1, 3500) and u == "ibm utterance 1"
    for world in (1, 2, 3):
        parts = []
        for rank in range(world):
            s = RankBatchSampler(8, rank, world, shuffle=True)
            s.set_epoch(3)
            parts.append(list(s))
        assert len({len(p) for p in parts}) == 1 and len(parts[0]) == 8 // world
        flat = sum(parts, [])
        assert len(set(flat)) == len(flat)                      # disjoint shares of one permutation
    a, b = RankBatchSampler(8, 0, 2), RankBatchSampler(8, 0, 2)
    a.set_epoch(1); b.set_epoch(2)
    assert list(a) != list(b)


def test_build_loaders_mirrors_reference_wiring(corpus, tmp_path):
    """train.build_loaders (train.py:202-239): CombinedTVLibriSpeech for training, SortedTV on the `eval` twins of the
    non-gigaspeech manifests, IBMDataset batched by 32 * multiplier * 2 -- checked down to the clips the raw loaders
    yield (the GPU featuriser is applied by FeaturizingLoader and is not exercised here)."""
    from speech_amd import train
    root, libri_entries, tv_entries = corpus
    mdir = tmp_path / "manifests"
    (mdir / "datasets" / "librispeech").mkdir(parents=True)
    (mdir / "datasets" / "first").mkdir(parents=True)
    # a LibriSpeech manifest long enough to survive the [:-2000] cut of its longest clips
    big = libri_entries + [("/nowhere/LibriSpeech/x.wav", 10 ** 9)] * 2000
    (mdir / "datasets" / "librispeech" / "sorted_train_librispeech.pkl").write_bytes(pickle.dumps(big))
    tv_train = mdir / "datasets" / "first" / "sorted_train_cer_0.1.pkl"
    tv_train.write_bytes(pickle.dumps([(p, n / 16000) for p, n in tv_entries]))
    (mdir / "datasets" / "first" / "sorted_eval_cer_0.1.pkl").write_bytes(pickle.dumps(tv_entries))
    ibm = tmp_path / "ibm"
    ibm.mkdir()
    _write_wav(ibm / "a.wav", np.zeros(3000))
    (ibm / "a.txt").write_text("ibm a")
    hp = train.parse_args(["--multiplier", "1", "--manifest_root", str(mdir), "--ibm_root", str(ibm),
                           "--root_map", "/nonexistent=/elsewhere", "--num_workers", "0"])
    assert hp["root_map"] == (("/nonexistent", "/elsewhere"),)
    hp["batch_size"], hp["tv_manifests"] = 2, [str(tv_train)]
    train_loader, sampler, evals = train.build_loaders(hp, 0, 1)
    assert [n for n, _ in evals] == ["tv", "ibm"]
    assert len(train_loader) == len(sampler) == (len(libri_entries) + len(tv_entries)) // 2
    batches = list(train_loader.loader)
    assert all(len(b) == 2 and b[0][0].dtype == torch.int16 and isinstance(b[0][1], str) for b in batches)
    assert len(list(evals[0][1].loader)) == len(tv_entries) // 2
    assert list(evals[1][1].loader)[0][0][1] == "ibm a"


def test_eval_dataset_cli_builds_the_reference_datasets(corpus, tmp_path):
    """scripts/eval_dataset.py:95-128: `libri` / `tv` / `ibm` / `eval_high_cer` through the dataset classes with the
    reference's manifest names, shuffle off, whole pre-sorted batches (the IBM directory batched by the loader); argument
    parsing as `beam_decode dataset_type model_file`.  The raw loaders are walked here; the featuriser needs the GPU."""
    from speech_amd.scripts import eval_dataset as ev
    root, libri_entries, tv_entries = corpus
    mdir = tmp_path / "manifests"
    (mdir / "datasets" / "librispeech").mkdir(parents=True)
    (mdir / "datasets" / "librispeech" / "sorted_test_clean_librispeech.pkl").write_bytes(pickle.dumps(libri_entries))
    for name in ("first", "second", "third", "fourth"):
        (mdir / "datasets" / name).mkdir(parents=True)
        (mdir / "datasets" / name / "sorted_eval_cer_0.2.pkl").write_bytes(pickle.dumps(tv_entries if name == "first" else []))
    (mdir / "eval_paths_0.21cer.pkl").write_bytes(pickle.dumps(tv_entries))
    ibm = tmp_path / "ibm"
    ibm.mkdir()
    for k in range(3):
        _write_wav(ibm / f"u{k}.wav", np.zeros(3000 + k))
        (ibm / f"u{k}.txt").write_text(f"ibm {k}")
    a = ev.parse_args(["False", "libri", "m.pth", "--manifest_root", str(mdir), "--root_map", "/nonexistent=/x"])
    assert (a.beam_decode, a.dataset_type, a.model_file, a.dtype) == (False, "libri", "m.pth", "fp16")      # model.half()
    assert a.root_map == (("/nonexistent", "/x"),)
    ds, bs = ev.build_dataset("libri", 2, str(mdir), a.root_map)
    assert bs is None and len(ds) == 2              # "test" manifest: nothing cut off its end
    batches = list(ev.build_loader(ds, bs, 0).loader)
    assert [len(b) for b in batches] == [2, 2] and batches[0][0][1] == "hello world 0"
    ds, bs = ev.build_dataset("tv", 2, str(mdir), (("/nonexistent", "/x"),))
    assert bs is None and len(ds) == 1 and list(ev.build_loader(ds, bs, 0).loader)[0][1][1] == "tv clip 1"
    ds, bs = ev.build_dataset("eval_high_cer", 1, str(mdir), (("/nonexistent", "/x"),))
    assert len(ds) == 2
    ds, bs = ev.build_dataset("ibm", 2, ibm_root=str(ibm))
    assert bs == 2 and [len(b) for b in ev.build_loader(ds, bs, 0).loader] == [2, 1]
    with pytest.raises(SystemExit):
        ev.parse_args(["False", "nope", "m.pth"])
    with pytest.raises(SystemExit):
        ev.parse_args(["maybe", "libri", "m.pth"])
    assert ev.parse_args(["True", "ibm", "m.pth", "--lm", "x.arpa"]).lm == "x.arpa"


def test_load_checkpoint_key_tolerance():
    """SRModel.load_checkpoint takes both checkpoint shapes (train.py:171-178 dict, bare state_dict) and tolerates the
    keys whose presence depends on the sru / torch version (scale_x buffers, BatchNorm step counters); anything else
    that is missing or unknown is an error."""
    from speech_amd import net
    torch.manual_seed(0)
    src = net.SRModel(2, 64, 28, 80, 0.1)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    assert "sru_layers.rnn_lst.1.weight_c" in sd and "classifier.1.weight" in sd and "cnn.bias" in sd
    stripped = {k: v for k, v in sd.items() if not k.endswith((".scale_x", ".num_batches_tracked"))}
    assert len(stripped) < len(sd)
    dst = net.SRModel(2, 64, 28, 80, 0.1)
    res = dst.load_checkpoint({"epoch": 3, "model_state_dict": stripped, "optimizer_state_dict": {}})
    assert all(k.endswith((".scale_x", ".num_batches_tracked")) for k in res.missing_keys) and res.missing_keys
    for k, v in src.state_dict().items():
        if k in stripped:
            assert torch.equal(dst.state_dict()[k], v), k
    extra = dict(sd)
    extra["sru_layers.rnn_lst.0.extra.scale_x"] = torch.ones(1)          # an extra buffer of that kind is ignored
    assert dst.load_checkpoint(extra).unexpected_keys == ["sru_layers.rnn_lst.0.extra.scale_x"]
    bad = dict(sd)
    bad["sru_layers.rnn_lst.0.weight_proj"] = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match="weight_proj"):
        dst.load_checkpoint(bad)
    del stripped["sru_layers.rnn_lst.0.bias"]
    with pytest.raises(RuntimeError, match="rnn_lst.0.bias"):
        dst.load_checkpoint(stripped)


def test_sentence_piece_script(tmp_path):
    """scripts/sentence_piece.py mirror: transcripts gathered from a pickled manifest (path rewriting included), a
    small BPE model trained, and text.SentencePieceVocab reading it back."""
    import pickle
    spm = pytest.importorskip("sentencepiece")  # noqa: F841
    from speech_amd import text
    from speech_amd.scripts import sentence_piece as sps
    real = tmp_path / "hd1"
    real.mkdir()
    words = "the quick brown fox jumps over the lazy dog and runs far away from here".split()
    entries = []
    for i in range(40):
        (real / f"c{i}.txt").write_text(" ".join(words[(i + j) % len(words)] for j in range(6)) + "\n")
        entries.append((str(tmp_path / "data" / f"c{i}.wav"), 16000))
    (tmp_path / "m.pkl").write_bytes(pickle.dumps(entries))
    out = tmp_path / "sp" / "sentences.txt"
    n = sps.gather_sentences([str(tmp_path / "m.pkl")], str(out), ((str(tmp_path / "data"), str(real)),))
    assert n == 40 and len(out.read_text().splitlines()) == 40
    model = sps.generate_sp_model(str(out), vocab_size=40, model_type="bpe", model_prefix=str(tmp_path / "sp" / "tv"))
    vocab = text.SentencePieceVocab(model)
    ids = vocab.text_to_int("the lazy dog")
    assert min(ids) >= 1 and vocab.size == 41 and vocab.int_to_text(ids) == "the lazy dog"
