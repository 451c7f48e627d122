"""long race screen of the in-kernel two-slab reduction: N launches over three data sets, bitwise against the reduction launch"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('', 'tests', os.path.join('tests', 'golden')):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch
import test_wgrad_group_gpu as T
from pasero_amd import functional as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sets = [T._c5_layer_entries(100 + 10 * k, decoder=bool(k & 1)) for k in range(3)]
prev = T._pair_mode(0)
refs = [[(dw.clone(), None if db is None else db.clone()) for dw, db in F.wgrad_group(s)] for s in sets]
T._pair_mode(1)
side = torch.cuda.Stream()
src = torch.randn(64 << 20, device='cuda'); dst = torch.empty_like(src)
bad = 0; t0 = time.time()
for it in range(n):
    if it % 3 != 2:
        with torch.cuda.stream(side):
            dst.copy_(src)
    k = (it * 7) % 3
    out = F.wgrad_group(sets[k])
    same = torch.stack([torch.equal(dw, rw) and ((db is None and rb is None) or torch.equal(db, rb)) and torch.tensor(True) for (dw, db), (rw, rb) in zip(out, refs[k])])
    if not bool(same.all()):
        bad += 1
        print('MISMATCH at', it, k, flush=True)
    if it % 500 == 0:
        print(it, 'launches', round(time.time() - t0, 1), 's, mismatches', bad, flush=True)
torch.cuda.synchronize()
T._pair_mode(prev)
print('done:', n, 'launches, mismatches', bad)
