"""Runs the SRU projection GEMMs at the bench's sizes (M = 751 x 32 rows, 1024 <-> 3072) through sp_gemm, a few
launches each with an HBM-bound launch in between, so that a PMC pass can read MFMA utilisation, wait cycles and
L2 hit rates per kernel (profiles/r02_gemm_pmc.md):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_g -- python3 tools/pmc_gemm.py
    python3 tools/pmc_gemm.py --sum gpurun_out/pmc_g        # per-kernel averages of a finished pass
"""
import collections
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--sum":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "gemm" in k:
            print(k)
            for c, v in d.items():
                print("    %-28s %16.0f per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import ops          # noqa: E402
dev = torch.device("cuda")
M = 24032
xn = torch.empty(M, 1088, device=dev, dtype=torch.bfloat16)[:, :1024]      # padded row pitch, as in the model
xn.copy_(torch.nn.functional.layer_norm(torch.randn(M, 1024, device=dev), (1024,)))
wk = (torch.randn(1024, 3072, device=dev) * 0.03).bfloat16()
wkT = torch.empty(3072, 1088, device=dev, dtype=torch.bfloat16)[:, :1024]
wkT.copy_(wk.t())
gU = (torch.randn(M, 3072, device=dev) * 1e-3).bfloat16()
oU = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16); oX = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
P = torch.empty(5, 1024, 3072, device=dev)
spacer = torch.empty(32 << 20, device=dev)
for _ in range(3):
    ops.gemm(xn, wkT, out=oU); spacer.mul_(1.0)
    ops.gemm(gU, wk, out=oX); spacer.mul_(1.0)
    ops.gemm(xn.t(), gU.t(), out=P, out_dtype=torch.float32, split_k=5); spacer.mul_(1.0)
torch.cuda.synchronize()
