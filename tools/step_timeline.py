"""how far the GPU runs behind the host along one training step: events recorded at every layer boundary (forward hooks,
gradient hooks on the layer outputs); lag = GPU time of the event - host time of its recording, both from a common start
after a synchronisation.  A lag near zero = the GPU is waiting for the host there."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import bench
from pasero_amd import functional as PF
from pasero_amd import transformer as T
w = sys.argv[1] if len(sys.argv) > 1 else 'c4_iwslt'
cfg, model, batch, wav = bench.build_workload(w, torch.bfloat16, torch.device('cuda:0'))
marks = []
on = [False]
def mark(name):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, time.perf_counter(), e))
def fwd_hook(name):
    def h(mod, inp, out):
        mark('fwd ' + name)
        o = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(o) and o.requires_grad:
            o.register_hook(lambda g: mark('bwd>' + name))
    return h
for n, m in model.named_modules():
    if isinstance(m, (T.TransformerEncoderLayer, T.TransformerDecoderLayer)):
        m.register_forward_hook(fwd_hook(n))
def step():
    for p in model.parameters():
        p.grad = None
    mark('grads cleared')
    if wav is not None:
        batch['encoder_input'] = PF.log_mel(wav).to(torch.bfloat16)
        mark('log-mel')
    loss, logs = model(**batch)
    mark('loss')
    loss.backward()
    mark('backward enqueued')
    n = logs['num_tokens']
    mark('logs read')
    return n
for _ in range(4):
    step()
torch.cuda.synchronize()
on[0] = True
for s in range(3):
    mark('step %d' % s)
    step()
mark('end')
torch.cuda.synchronize()
n0, h0, e0 = marks[0]
print('%-44s %10s %10s %9s' % ('mark', 'host ms', 'gpu ms', 'lag ms'))
for name, h, e in marks:
    g = e0.elapsed_time(e)
    print('%-44s %10.2f %10.2f %9.2f' % (name, 1e3 * (h - h0), g, g - 1e3 * (h - h0)))
