"""Which hardware queue do the step's streams land on?  Run under rocprofv3 --kernel-trace and read queue ids from the trace
(tools/queue_report.py).  Launches a tagged elementwise kernel on: the null stream, pool streams of both priorities."""
import torch
dev = torch.device("cuda", 0)
x = [torch.zeros(1 << 20, device=dev) for _ in range(8)]
streams = [("null", torch.cuda.current_stream(dev))] + [(f"p0_{i}", torch.cuda.Stream(dev)) for i in range(3)] + \
          [(f"hi_{i}", torch.cuda.Stream(dev, priority=-1)) for i in range(3)]
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "", flush=True)
for k, (name, s) in enumerate(streams):
    with torch.cuda.stream(s):
        for _ in range(k + 1):      # k+1 launches of add_ identify the stream in the trace
            x[k].add_(1.0)
    torch.cuda.synchronize()
    print(name, s.cuda_stream, flush=True)
