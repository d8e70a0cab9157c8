"""Encoder forward (HIP graph, 384 frames) alone and beside a few resident spinning wavefronts on another stream."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "bsc-nav_amd", "tunableop_gfx950.csv")
import shutil; shutil.copy(src, "/tmp/evh_0.csv")
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME="/tmp/evh_.csv", PYTORCH_TUNABLEOP_VERBOSE="0")
import torch
from bsc_nav_amd import encoder
hog = C.CDLL(os.path.join(ROOT, "scripts", "microbench", "libhog.so"))
vit = encoder.RandomViT("vit_b16").cuda()
B = 384
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
enc = encoder.GraphedEncoder(vit, B, 480, 640, 4, True)
sink = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()
for n_hog, thr in ((0, 64), (1, 64), (16, 64), (512, 64)):
    torch.cuda.synchronize()
    if n_hog:
        hog.hog_launch(C.c_void_p(side.cuda_stream), n_hog, thr, C.c_double(120000.0), C.c_void_p(sink.data_ptr()))
        time.sleep(0.003)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        enc(rgb)
    e1.record(); e1.synchronize()
    print(f"hog {n_hog} x {thr}: encoder {e0.elapsed_time(e1) / 5:.2f} ms per 384 frames")
    torch.cuda.synchronize()
