"""two identical C5-width steps (pair mode active: 8192 rows, d = 1024) must give bitwise identical gradients, 10 times"""
import os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests/golden')
import paramgen
from pasero_amd import config as C, rng
from pasero_amd.transformer import Transformer
cfg = C.NLLB1B3Config(encoder_layers=2, decoder_layers=2, dropout=0.1)
torch.manual_seed(0)
model = Transformer(cfg, C.DistributedConfig(), C.SyntheticTask(4000)).to(torch.bfloat16).cuda().train()
batch = {k: torch.from_numpy(v).cuda() for k, v in paramgen.make_text_batch(3, 64, 128, 128, 4000).items()}
ref = None
side = torch.cuda.Stream(); src = torch.randn(32 << 20, device='cuda'); dst = torch.empty_like(src)
for it in range(10):
    if it % 2:
        with torch.cuda.stream(side):
            dst.copy_(src)
    rng.manual_seed(7)
    for p in model.parameters(): p.grad = None
    loss, _ = model(**batch); loss.backward()
    g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    if ref is None: ref = (loss.item(), g)
    else:
        assert loss.item() == ref[0], (it, loss.item(), ref[0])
        bad = [n for n in g if not torch.equal(g[n], ref[1][n])]
        assert not bad, (it, bad[:5])
torch.cuda.synchronize()
print('C5-width step: 10 runs bitwise identical; loss', ref[0])
