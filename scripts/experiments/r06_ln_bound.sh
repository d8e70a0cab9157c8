# upper bound of what producer-written LayerNorm pieces could buy: the f32 forward fused (default), with LayerNorm passes, and with
# the passes skipped (stale pieces: WRONG results, timing only).  Alternating, one box; 12 forwards each after warm-up.
cat > /tmp/enc_t.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bsc_nav_amd import encoder
B = 768
vit = encoder.RandomViT("vit_b16", image_size=224, seed=0, dtype=torch.float32).cuda()
rgb = torch.randint(0, 255, (B, 480, 640, 4), dtype=torch.uint8, device="cuda")
g = encoder.GraphedEncoder(vit, B, 480, 640, 4, False)
for _ in range(6): g(rgb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(12): g(rgb)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 12 * 1e3:.2f} ms per forward")
PY
for i in 1 2; do
  for v in "X=1" "BSC_ENC_LN_FUSED=0" "BSC_ENC_LN_FUSED=0 BSC_ENC_LN_SKIP=1"; do echo -n "[$v] "; env $v python /tmp/enc_t.py 2>&1 | tail -1; done
done
