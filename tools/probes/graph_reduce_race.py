"""Development aid: is torch's column sum (the bias gradient of a stock nn.Linear: a two-pass reduction whose semaphores are cleared by a
memset node) reproducible inside a replayed HIP graph with a second branch running beside it?"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
dy = torch.randn(640, 768, device=dev).bfloat16()
a = torch.randn(4096, 4096, device=dev).bfloat16()
for two in (False, True):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out = torch.empty(768, device=dev, dtype=torch.bfloat16)
    junk = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)

    def work():
        cur = torch.cuda.current_stream()
        if two:
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                t = a
                for _ in range(6):
                    t = (t @ a) * 1e-2
                junk.copy_(t)
        for _ in range(8):
            out.copy_(dy.sum(0))
        if two:
            cur.wait_stream(s2)

    with torch.cuda.stream(s1):
        work()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        work()
    first, bad = None, 0
    for it in range(200):
        g.replay()
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
        elif not torch.equal(first, out):
            bad += 1
    ref = dy.float().sum(0)
    print(f"second branch {two}: {bad} of 199 replays differ from the first; first vs fp32 reference max |diff| {(first.float() - ref).abs().max().item():.3e}", flush=True)
