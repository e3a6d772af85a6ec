"""CPU oracle for the speech hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of fernand/speech's ASR hot path
(magnitude-mel frontend, SRU encoder, CTC loss/gradient, greedy CTC decode, CER/WER).
It exists to check the HIP kernels in ``speech_amd`` and to time a CPU baseline.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``speech_amd/`` imports it: the product path has no CPU fallback
and raises when ``libspeech_hip.so`` is missing.

Pinning status (what each restatement was checked against in this container):

* ``oracle.text``    -- pinned: compared with the reference's own ``text.py`` (importable),
                        vectors in ``tests/golden/text_golden.json``.
* ``oracle.decode``  -- pinned: ``greedy_decoder``/``greedy_decode`` compared with the
                        reference's ``decoder.py`` functions run here (its ``Levenshtein``
                        import is not installed; the golden script registers an empty module
                        of that name, the greedy functions never touch it).  ``cer``/``wer``
                        use an own edit distance: python-Levenshtein is absent -> the
                        *distance* is the textbook definition, parity unpinned at the
                        package level.
* ``oracle.ctc``     -- pinned: the reference calls ``torch.nn.CTCLoss`` (train.py:266) and
                        torch 2.10 CPU is installed here, so ``ctc_torch`` IS the reference
                        arithmetic; ``ctc_numpy`` is an independent float64 alpha/beta
                        restatement checked against it.
* ``oracle.mel``     -- parity unpinned at the reference level: torchaudio is not installed
                        (data.py:8 cannot be imported).  Restated from torchaudio's documented
                        defaults on top of ``torch.stft`` (which is what torchaudio calls) and
                        cross-checked against an independent float64 DFT.
* ``oracle.sru`` / ``oracle.net`` -- parity unpinned: the ``sru`` package (net.py:3) is not
                        installed and not vendored; restated from the SRU paper / package
                        behaviour as recorded in SURVEY.md section 8 (N4).
"""
