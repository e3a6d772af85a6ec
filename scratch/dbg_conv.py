import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_kernels import _conv_ref_and_hip
cuda = torch.device('cuda:0')
_, (cnn, blocks, out), (cnn_r, blocks_r, ref) = _conv_ref_and_hip(cuda, 2, 24, 150, 9)
(0.5*ref.float()**2).sum().backward(); (0.5*out.float()**2).sum().backward()
def rel(a, b): return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)
for i in (1, 0):
    a, b = blocks[i], blocks_r[i]
    for n in ['bn2.bias', 'bn2.weight', 'conv2.weight', 'bn1.bias', 'bn1.weight', 'conv1.weight']:
        m, p = n.split('.')
        ga, gb = getattr(getattr(a, m), p).grad, getattr(getattr(b, m), p).grad
        print(i, n, rel(ga, gb), float(gb.abs().max()))
print('stem w', rel(cnn.weight.grad, cnn_r.weight.grad), 'b', rel(cnn.bias.grad, cnn_r.bias.grad))
print(cnn.weight.grad.flatten()[:9], cnn_r.weight.grad.flatten()[:9])
print(cnn.bias.grad[:8], cnn_r.bias.grad[:8])
