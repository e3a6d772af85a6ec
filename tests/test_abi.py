"""CPU tests of the boundary: the C-ABI library loads and exports every symbol that
include/speech_hip.h declares; host-only entry points behave; the kernel math header replays
correctly on the host.  No GPU compute calls here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "speech_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from speech_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -m speech_amd.build` first"
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(h, s), f"{s} declared in speech_hip.h but not exported"
    # the ctypes table binds exactly the declared symbols
    assert sorted(_lib.SIGNATURES) == syms


def test_version_and_error_string():
    from speech_amd import _lib
    h = _lib.lib()
    assert h.sp_version() >= 100
    # argument validation happens on the host before any launch
    rc = h.sp_greedy_decode(None, 0, 1, 4, 0, 0, None, None, None, None)
    assert rc == -1 and b"bad shape" in h.sp_last_error()
    rc = h.sp_sru_fwd(None, None, None, None, None, 4, 2, 100, 3, 0, None, None, None)
    assert rc == -1 and b"multiple of 64" in h.sp_last_error()
    rc = h.sp_ctc_fwd_bwd(None, 0, 0, None, 0, None, None, 10, 2, 5000, 0, 4, 1.0, None, None, None, 0, None, 0, None)
    assert rc == -1 and b"exceeds" in h.sp_last_error()
    rc = h.sp_ctc_fwd_bwd(None, 0, 0, None, 0, None, None, 10, 2, 29, 0, 4, 1.0, None, None, 1, 7, None, 0, None)
    assert rc == -1 and b"grad_dtype" in h.sp_last_error()
    # the general GEMM: strides, dtypes, slabs
    rc = h.sp_gemm(16, 16, 16, 8, 8, 8, 16, 2, 8, 1, 8, 2, 2, 0, 1, 0, None)
    assert rc == -1 and b"contiguous along k" in h.sp_last_error()
    rc = h.sp_gemm(16, 16, 16, 8, 8, 8, 8, 1, 8, 1, 4, 2, 2, 0, 1, 0, None)
    assert rc == -1 and b"ldc" in h.sp_last_error()
    rc = h.sp_gemm(16, 16, 16, 8, 8, 8, 8, 1, 8, 1, 8, 2, 2, 1, 1, 0, None)
    assert rc == -1 and b"accumulation" in h.sp_last_error()
    rc = h.sp_gemm(16, 16, 16, 8, 8, 8, 8, 1, 8, 1, 8, 2, 0, 0, 4, 8, None)
    assert rc == -1 and b"slabs" in h.sp_last_error()
    rc = h.sp_gemm(16, 16, 16, 8, 8, 8, 8, 1, 8, 1, 8, 5, 0, 0, 1, 0, None)
    assert rc == -1 and b"in_dtype" in h.sp_last_error()
    assert h.sp_gemm(None, None, None, 0, 8, 8, 8, 1, 8, 1, 8, 2, 2, 0, 1, 0, None) == 0
    assert h.sp_gemm_tn_splits(1024, 3072, 24032, 2) == 5 and h.sp_gemm_tn_splits(1024, 1024, 24032, 2) == 16
    assert h.sp_gemm_tn_splits(1024, 3072, 24032, 0) == 8 and h.sp_gemm_tn_splits(100, 100, 100, 2) == 1
    rc = h.sp_sum_slabs_f32(16, 2, 6, 8, 16, 0, None, None)
    assert rc == -1 and b"multiples of 4" in h.sp_last_error()
    rc = h.sp_linear_pack(16, 1000, 1024, 999, None, 2, 16, None, None)
    assert rc == -1 and b"Vp" in h.sp_last_error()                       # the padded row count cannot be below V
    rc = h.sp_linear_pack(16, 1000, 1024, 1024, None, 0, 16, None, None)
    assert rc == -1 and b"dtype" in h.sp_last_error()                    # 16-bit outputs only
    rc = h.sp_sru_pack_wt(16, 96, 64, 3, 1, 2, 16, 96, None, None)
    assert rc == -1 and b"n_in" in h.sp_last_error()                     # bf16 tiles are 64 rows
    rc = h.sp_sru_pack_wt(16, 128, 64, 3, 1, 2, 16, 100, None, None)
    assert rc == -1 and b"ldo" in h.sp_last_error()
    rc = h.sp_layernorm_fwd(16, 16, 16, 4, 64, 1e-5, 0, 16, 60, 16, 16, None)
    assert rc == -1 and b"ldy" in h.sp_last_error()
    assert h.sp_ctc_workspace_bytes(100, 4, 29, 20) > 0
    assert h.sp_melspec_plan_bytes() > 4 * (400 + 400)


def test_edit_distance_native():
    from speech_amd import ops
    from oracle import decode as odec
    rs = np.random.RandomState(0)
    assert ops.edit_distance([], []) == 0
    assert ops.edit_distance([1, 2, 3], []) == 3
    assert ops.edit_distance(list(b"kitten"), list(b"sitting")) == 3
    for _ in range(50):
        a = rs.randint(0, 5, rs.randint(0, 30)).tolist()
        b = rs.randint(0, 5, rs.randint(0, 30)).tolist()
        assert ops.edit_distance(a, b) == odec.levenshtein(a, b)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from speech_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SpeechHipError):
        _lib.lib()


def test_mel_math_host_replay(tmp_path):
    """The kernel's FFT decomposition (mel_math.h) replayed lane-task by lane-task on the host
    reproduces the float64 DFT oracle."""
    exe = tmp_path / "mel_emulate"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe),
                           os.path.join(ROOT, "tests", "host", "mel_emulate.cpp")])
    g = np.load(os.path.join(ROOT, "tests", "golden", "mel_golden.npz"))
    g["fbank"].astype(np.float32).tofile(tmp_path / "fb.f32")
    for name in ["noise", "sine_1k", "impulse", "dc", "short_201", "ragged_1777"]:
        wav = g[name + "__wav"].astype(np.float32)
        wav.tofile(tmp_path / "w.f32")
        subprocess.check_call([str(exe), str(tmp_path / "w.f32"), str(tmp_path / "fb.f32"), str(tmp_path / "o.f32")])
        out = np.fromfile(tmp_path / "o.f32", dtype=np.float32).reshape(80, 1 + len(wav) // 160)
        ref = g[name + "__mel_f64"]
        assert np.abs(out - ref).max() <= 1e-6 * np.abs(ref).max(), name
