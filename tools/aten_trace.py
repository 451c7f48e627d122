"""which aten ops with device work a step of a workload calls, with sizes and the Python frames that called them"""
import os, sys, torch, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (tools/ -> repo root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from pasero_amd import functional as PF
from torch.utils._python_dispatch import TorchDispatchMode
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4_whisper'
cfg, model, batch, wav = bench.build_workload(wl, torch.bfloat16, 'cuda')
def step():
    for p in model.parameters(): p.grad = None
    if wav is not None: batch['encoder_input'] = PF.log_mel(wav).to(torch.bfloat16)
    loss, logs = model(**batch); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
agg = collections.Counter()
SKIP = ('view', 'reshape', 'detach', 'alias', 'empty', 'as_strided', 'slice', 'select', 'transpose', 'unsqueeze', 'squeeze', 't.', 'expand', 'permute', 'split', 'unbind', 'narrow', '_unsafe_view', 'size', 'stride', 'is_', 'item', '_local_scalar')
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            fr = [f for f in traceback.extract_stack() if 'pasero_amd' in f.filename or 'bench.py' in f.filename][-2:]
            agg[(name, str(shapes), ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (name, shapes, where), n in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f'{n:4d} x {name:30s} {shapes:45s} {where}')
