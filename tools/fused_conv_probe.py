import torch, time
import torch.nn.functional as F
dev = "cuda"
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for cl in (False, True):
    x = torch.randn(16, 64, 96, 320, device=dev)
    w1 = torch.randn(64, 64, 1, 1, device=dev) * 0.05; b1 = torch.randn(64, device=dev)
    w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    w3 = torch.randn(256, 64, 1, 1, device=dev) * 0.05; b3 = torch.randn(256, device=dev)
    z = torch.randn(16, 256, 96, 320, device=dev)
    if cl:
        x, z = x.contiguous(memory_format=torch.channels_last), z.contiguous(memory_format=torch.channels_last)
        w1, w2, w3 = (w.contiguous(memory_format=torch.channels_last) for w in (w1, w2, w3))
    with torch.no_grad():
        ref = F.relu(F.conv2d(x, w2, b1, 1, 1))
        try:
            got = torch.ops.aten.miopen_convolution_relu(x, w2, b1, [1, 1], [1, 1], [1, 1], 1)
            print("cl", cl, "conv3x3+relu: max diff", (ref - got).abs().max().item(),
                  "separate ms %.3f" % t(lambda: F.relu(F.conv2d(x, w2, b1, 1, 1))),
                  "fused ms %.3f" % t(lambda: torch.ops.aten.miopen_convolution_relu(x, w2, b1, [1, 1], [1, 1], [1, 1], 1)))
        except Exception as e:
            print("miopen_convolution_relu failed:", repr(e)[:200])
        ref = F.relu(F.conv2d(x, w3, b3) + z)
        try:
            got = torch.ops.aten.miopen_convolution_add_relu(x, w3, z, 1.0, b3, [1, 1], [0, 0], [1, 1], 1)
            print("cl", cl, "conv1x1+add+relu: max diff", (ref - got).abs().max().item(),
                  "separate ms %.3f" % t(lambda: F.relu(F.conv2d(x, w3, b3) + z)),
                  "fused ms %.3f" % t(lambda: torch.ops.aten.miopen_convolution_add_relu(x, w3, z, 1.0, b3, [1, 1], [0, 0], [1, 1], 1)))
        except Exception as e:
            print("miopen_convolution_add_relu failed:", repr(e)[:200])
