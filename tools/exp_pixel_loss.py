"""nm_pixel_loss alone at the BASELINE image sizes (HIP events).    python tools/exp_pixel_loss.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
for (h, w) in ((256, 256), (800, 800), (1080, 1920)):
    img, gt = torch.rand(3, h, w, device=dev), torch.rand(3, h, w, device=dev)
    loss, g = torch.zeros((), device=dev), torch.empty(3, h, w, device=dev)
    for _ in range(5):
        lib.nm_pixel_loss(1, 1.0, h, w, 0, 0, L.ptr(img), L.ptr(gt), L.ptr(loss), L.ptr(g), L.stream_ptr(dev))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        lib.nm_pixel_loss(1, 1.0, h, w, 0, 0, L.ptr(img), L.ptr(gt), L.ptr(loss), L.ptr(g), L.stream_ptr(dev))
    b.record(); torch.cuda.synchronize()
    print(f"{w}x{h}: {1e3 * a.elapsed_time(b) / 50:.1f} us per call ({48.0 * h * w / (a.elapsed_time(b) / 50 * 1e-3) / 1e9:.0f} GB/s of 48 B per pixel)")
