"""The no-alloc CTC entry point (sp_ctc_fwd_bwd_noalloc: the workspace is a buffer built into the library)
against sp_ctc_fwd_bwd with a caller-owned workspace: bit-identical nll / loss / gradient, on two calls in a
row, the second with a smaller input (so stale contents of the shared buffer would show)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(T, B, V, Lmax, seed, grad_dtype):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(T, B, V, generator=g) * 3).cuda()
    lens = torch.randint(max(1, Lmax // 2), Lmax + 1, (B,), generator=g).to(torch.int32)
    lens[0] = Lmax
    labels = torch.randint(1, V, (B, Lmax), generator=g)
    labels[:, 1::5] = labels[:, 0::5][:, : labels[:, 1::5].shape[1]]      # adjacent repeats
    il = torch.full((B,), T, dtype=torch.int32)
    il[-1] = T - 3                                                         # one shorter input
    return logits, labels.cuda(), il.cuda(), lens.cuda(), grad_dtype


def _run(entry, case, T, B, V, Lmax):
    from speech_amd import _lib
    from speech_amd._lib import check, lib, ptr, stream_ptr
    logits, labels, il, lens, gdt = case
    nll = torch.full((B,), -7.0, device="cuda")
    loss = torch.full((1,), -7.0, device="cuda")
    grad = torch.full(logits.shape, 3.0, dtype=gdt, device="cuda")
    L = lib()
    common = (ptr(logits), logits.stride(0), logits.stride(1), ptr(labels), labels.stride(0), ptr(il), ptr(lens),
              T, B, V, 0, Lmax, 1.0, ptr(nll), ptr(loss), ptr(grad), _lib.dtype_code(grad))
    if entry == "noalloc":
        check(L.sp_ctc_fwd_bwd_noalloc(*common, stream_ptr()), "sp_ctc_fwd_bwd_noalloc")
    else:
        ws = torch.empty(L.sp_ctc_workspace_bytes(T, B, V, Lmax), dtype=torch.uint8, device="cuda")
        check(L.sp_ctc_fwd_bwd(*common, ptr(ws), ws.numel(), stream_ptr()), "sp_ctc_fwd_bwd")
    torch.cuda.synchronize()
    return nll, loss, grad


def test_ctc_noalloc_matches_workspace_entry_point():
    from speech_amd._lib import lib
    L = lib()
    assert L.sp_ctc_noalloc_max_bytes() == 64 << 20
    # first call: long lattice (multi-state lanes), large vocabulary, bf16 gradient; second: smaller in
    # every dimension, char vocabulary, f32 gradient -- it must not see anything the first call left
    shapes = [(300, 6, 1000, 40, 1, torch.bfloat16), (75, 3, 29, 12, 2, torch.float32)]
    for T, B, V, Lmax, seed, gdt in shapes:
        case = _case(T, B, V, Lmax, seed, gdt)
        a = _run("noalloc", case, T, B, V, Lmax)
        b = _run("workspace", case, T, B, V, Lmax)
        for x, y, name in zip(a, b, ("nll", "loss", "grad")):
            assert torch.isfinite(x.float()).all(), name
            assert torch.equal(x, y), f"{name} differs at T={T} B={B} V={V}"


def test_ctc_noalloc_rejects_oversize_and_allocates_nothing():
    from speech_amd._lib import lib
    L = lib()
    # 1501 frames x 64 utterances x 450 labels: 0.9 GB of lattice
    assert L.sp_ctc_workspace_bytes(1501, 64, 29, 450) > L.sp_ctc_noalloc_max_bytes()
    rc = L.sp_ctc_fwd_bwd_noalloc(None, 0, 0, None, 450, None, None, 1501, 64, 29, 0, 450, 1.0, None, None, None, 0, None)
    assert rc == -2 and b"scratch" in L.sp_last_error()
    # no device allocation happens inside the call: torch's allocator statistics and the driver's free
    # memory are unchanged by it
    case = _case(75, 3, 29, 12, 5, torch.float32)
    _run("noalloc", case, 75, 3, 29, 12)           # set-up call
    free0, _ = torch.cuda.mem_get_info()
    from speech_amd import _lib
    from speech_amd._lib import check, ptr, stream_ptr
    logits, labels, il, lens, _ = case
    nll = torch.empty(3, device="cuda"); loss = torch.empty(1, device="cuda"); grad = torch.empty_like(logits)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    check(L.sp_ctc_fwd_bwd_noalloc(ptr(logits), logits.stride(0), logits.stride(1), ptr(labels), labels.stride(0),
                                   ptr(il), ptr(lens), 75, 3, 29, 0, 12, 1.0, ptr(nll), ptr(loss), ptr(grad), 0,
                                   stream_ptr()), "noalloc")
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0
