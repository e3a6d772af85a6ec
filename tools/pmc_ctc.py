"""sp_ctc_fwd_bwd three times at each of the four bench shapes, for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
passes (separate runs, counters only; MI355X_MICROARCH.md "HBM").  `--sum <dir>`: per (kernel, grid) mean of the
counter over the last two launches, in KiB, from the pass's counter_collection.csv."""
import collections
import csv
import glob
import os
import sys

SHAPES = ((751, 32, 1000, 60), (751, 32, 29, 225), (1501, 64, 1000, 120), (1501, 64, 29, 448))


def summarise(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "ctc" in r["Kernel_Name"]:
            acc.setdefault((r["Counter_Name"], r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    for (cn, k, g), v in acc.items():
        print(f"{cn:11s} {k:60s} grid {g:>9s} n={len(v)} {sum(v[-2:]) / len(v[-2:]):12.0f} KiB")


def main():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from speech_amd import ops
    dev = torch.device("cuda")
    for T, B, V, L in SHAPES:
        g = torch.Generator().manual_seed(1)
        logits = (torch.randn(T, B, V, generator=g) * 3).to(dev)
        labels = torch.randint(1, V, (B, L), generator=g).to(dev)
        ll = (torch.rand(B, generator=g) * 0.5 + 0.5).mul(L).long().clamp(min=1).int().to(dev)
        ll[0] = L
        il = torch.full((B,), T, dtype=torch.int32, device=dev)
        for _ in range(3):
            ops.ctc_fwd_bwd(logits, labels, il, ll, time_major=True, max_label_len=L)
        torch.cuda.synchronize()
        print(T, B, V, L, "sum(2L+1) =", int((2 * ll + 1).sum()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--sum":
        summarise(sys.argv[2])
    else:
        main()
