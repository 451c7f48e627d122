"""which calls of a training step synchronise the host with the GPU, and how long the host takes to enqueue one step"""
import sys, time, traceback, warnings
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import bench
from pasero_amd import functional as PF
w = sys.argv[1] if len(sys.argv) > 1 else 'c4_whisper'
cfg, model, batch, wav = bench.build_workload(w, torch.bfloat16, torch.device('cuda:0'))
def step(read=True):
    for p in model.parameters():
        p.grad = None
    if wav is not None:
        batch['encoder_input'] = PF.log_mel(wav).to(torch.bfloat16)
    loss, logs = model(**batch)
    loss.backward()
    return logs['num_tokens'] if read else 0
for _ in range(4):
    step()
torch.cuda.synchronize()
seen = {}
def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack()[:-2] if 'pasero_amd' in f.filename or 'sync_probe' in f.filename or 'bench.py' in f.filename]
    key = tuple((f.filename.split('/')[-1], f.lineno) for f in st[-3:])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = show
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode(1)
step()
torch.cuda.set_sync_debug_mode(0)
for k, v in seen.items():
    print('SYNC x%d at %s' % (v, ' <- '.join('%s:%d' % kk for kk in reversed(k))))
torch.cuda.synchronize()
# host time to enqueue a step (GPU far behind: after a sync the queue is empty, so the host is never blocked on queue space)
for read in (True, False):
    host = []
    t_all = time.perf_counter()
    for _ in range(20):
        t0 = time.perf_counter(); step(read); host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    host.sort()
    print(f'{w} read_logs={read}: host per step median {1e3*host[10]:.2f} ms (min {1e3*host[0]:.2f}, max {1e3*host[-1]:.2f}); 20 steps enqueued in {1e3*t_enq/20:.2f} ms/step, done in {1e3*t_tot/20:.2f} ms/step')
