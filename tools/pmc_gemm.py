"""Runs the SRU projection GEMMs at the bench's sizes (M = 751 x 32 rows, 1024 <-> 3072) on the library
(torch.mm -> hipBLASLt) and on sp_gemm_nt_bf16, a few launches each, so that a PMC pass can compare MFMA
utilisation (profiles/r01_mfma_util.md):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_g -- python3 tools/pmc_gemm.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import ops
dev = torch.device("cuda")
M = 24032
xn = torch.empty(M, 1088, device=dev, dtype=torch.bfloat16)[:, :1024]      # padded row pitch, as in the model
xn.copy_(torch.nn.functional.layer_norm(torch.randn(M, 1024, device=dev), (1024,)))
wk = (torch.randn(1024, 3072, device=dev) * 0.03).bfloat16()
wkT = torch.empty(3072, 1088, device=dev, dtype=torch.bfloat16)[:, :1024]
wkT.copy_(wk.t())
gU = (torch.randn(M, 3072, device=dev) * 1e-3).bfloat16()
oU = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16); oX = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
fwd_only = len(sys.argv) > 1 and sys.argv[1] == "fwd"     # only the own forward GEMM (bench.py's traffic figure)
for _ in range(3):
    if fwd_only:
        ops.gemm(xn, wkT, out=oU)
        continue
    torch.mm(xn, wk, out=oU); ops.gemm(xn, wkT, out=oU); torch.mm(gU, wk.t(), out=oX); ops.gemm(gU, wk, out=oX)
torch.cuda.synchronize()
