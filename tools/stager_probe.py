"""Where the host-staged input path spends its time (tools, not product): gather variants into pinned memory, H2D copy,
and the launching thread's issue rate with and without a busy gather thread beside it.
    python tools/stager_probe.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device("cuda", 0)
B, L, rows = 256, 50176, 2048
c = torch.rand(rows, L)
j = torch.randperm(rows)[:B]
pin = torch.empty((B, L), pin_memory=True)
dv = torch.empty((B, L), device=dev)
print("threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)))

def t(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

cn, pn, jn = c.numpy(), pin.numpy(), j.numpy()
print("index_select -> pinned  %.2f ms" % t(lambda: torch.index_select(c, 0, j, out=pin)))
print("np.take clip -> pinned  %.2f ms" % t(lambda: np.take(cn, jn, axis=0, out=pn, mode="clip")))
print("H2D 51 MB pinned        %.2f ms" % t(lambda: dv.copy_(pin, non_blocking=True)))
s2 = torch.cuda.Stream(device=dev)
def h2d_side():
    with torch.cuda.stream(s2):
        dv.copy_(pin, non_blocking=True)
print("H2D on a side stream    %.2f ms" % t(h2d_side))

# launching thread: a stand-in loop of tiny launches, alone and beside a gather thread
x = torch.zeros(1024, device=dev)
def issue(n=3000):
    t0 = time.perf_counter()
    for _ in range(n): x.add_(1.0)
    return (time.perf_counter() - t0) / n * 1e6
torch.cuda.synchronize()
print("issue alone             %.2f us per launch" % issue())
for name, g in (("index_select", lambda: torch.index_select(c, 0, j, out=pin)), ("np.take", lambda: np.take(cn, jn, axis=0, out=pn, mode="clip"))):
    stop = [False]
    cnt = [0]
    def work():
        while not stop[0]:
            g(); cnt[0] += 1
            time.sleep(0.02)
    th = threading.Thread(target=work); th.start()
    time.sleep(0.2)
    r = issue()
    stop[0] = True; th.join()
    torch.cuda.synchronize()
    print("issue beside %-12s %.2f us per launch (%d gathers meanwhile)" % (name, r, cnt[0]))
