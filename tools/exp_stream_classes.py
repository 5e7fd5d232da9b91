"""Which HIP streams of this process can run side by side?  A spin kernel on stream A, a tiny kernel on stream B behind a start event:
B finishes early <=> A and B sit on different hardware queues.  Prints the classes of the first 12 torch pool streams (+ the null
stream), then evaluate_pairs throughput for representatives of the classes.  python tools/exp_stream_classes.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(64, device=dev)
null = torch.cuda.current_stream(dev)
ss = [null]
for i in range(12):
    s_ = torch.cuda.Stream(dev); s_.cuda_stream; ss.append(s_)
# calibrate the spin
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cyc = 1_000_000
for _ in range(2):
    e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
cyc = int(cyc * 0.5 / max(ms, 1e-3))          # ~0.5 ms
def beside(a, b):
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        ea.record()
        torch.cuda._sleep(cyc)
    with torch.cuda.stream(b):
        b.wait_event(ea) if False else None
        y = x + 1
        eb.record()
    torch.cuda.synchronize()
    return ea.elapsed_time(eb)
print(f"spin {cyc} cycles = {0.5:.2f} ms nominal")
M = np.zeros((13, 13))
for i in range(13):
    for j in range(13):
        if i != j:
            M[i, j] = beside(ss[i], ss[j])
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print("ms until a tiny kernel on stream j (column) is done while stream i (row) spins 0.5 ms; index 0 = the null stream")
print(M)
same = M > 0.25
print("same-queue (serialised) pairs:", [(i, j) for i in range(13) for j in range(i + 1, 13) if same[i, j] or same[j, i]])
