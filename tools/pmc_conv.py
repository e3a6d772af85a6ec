"""Runs the 3x5 convolutions of the residual blocks (forward conv1 / conv2, the two backward-data forms, the two
weight-gradient forms) a few times at the bench's sizes (32 x 40 x 751 positions, 32 channels) so that PMC passes
can read their instruction mix and stall reasons (profiles/r02_conv_pmc.md):
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
        SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_c1 -- python3 tools/pmc_conv.py
    python3 tools/pmc_conv.py --sum gpurun_out/pmc_c1 [name filter]
"""
import collections
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--sum":
    flt = sys.argv[3] if len(sys.argv) > 3 else "conv"
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:64]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in sorted(acc.items()):
        if flt in k:
            print(k)
            for c, v in d.items():
                print("    %-28s %16.0f per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import net          # noqa: E402
dev = torch.device("cuda")
torch.manual_seed(0)
m = net.SRModel(1, 64, 29, 80, 0.1, 0, compute_dtype=torch.bfloat16).to(dev)
m.train()
spec = (torch.rand(32, 1, 80, 1501, device=dev) * 3)
from speech_amd import convstack  # noqa: E402
for _ in range(3):
    for p in m.parameters():
        p.grad = None
    x = convstack.conv_stack(spec, m.cnn, list(m.resnet_layers), True)
    x.float().square().mean().backward()
torch.cuda.synchronize()
