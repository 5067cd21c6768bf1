"""Round 6 probe: do Swin's weight-gradient launches (whose target-step results nobody reads) co-run with the text encoder's small vendor GEMMs, i.e. would a DEFERRED batch of them fill the
~10 ms per step in which text-encoder kernels run alone on half the chip?  One HIP graph with ONE fork: stream A = a dependent chain of text-backward-like GEMMs (2048 tokens), stream B = a
batch of weight-gradient launches; against the two run one after the other."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from facialmmt_amd import ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
# text-like chain: per "layer" four GEMMs of a RoBERTa-large backward step (dX only), dependent
W = [rnd(1024, 1024) * 0.03 for _ in range(2)] + [rnd(1024, 4096) * 0.03, rnd(4096, 1024) * 0.03]
def text_chain(x, layers=24):
    for _ in range(layers):
        x = x @ W[0]
        x = x @ W[1]
        h = x @ W[2]
        x = h @ W[3]
    return x
xt = rnd(2048, 1024)
# weight-gradient batch: stage-2 (DMA kernel, 512-thread workgroups, one per CU) and stage-0 (register-staged, 256-thread) shapes
shapes2 = [(125440, 1536, 384), (125440, 384, 1536), (125440, 1152, 384), (125440, 384, 384)]
shapes0 = [(2007040, 384, 96), (2007040, 96, 384)]
ops2 = [(rnd(M, N), rnd(M, K)) for (M, N, K) in shapes2]
ops0 = [(rnd(M, N), rnd(M, K)) for (M, N, K) in shapes0]
def wg_batch(which, reps):
    for _ in range(reps):
        for dy, x in which:
            ops.wgrad_raw(dy, x, True)
side = torch.cuda.Stream()
def run_case(name, which, reps):
    def serial():
        text_chain(xt); wg_batch(which, reps)
    def conc():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wg_batch(which, reps)
        text_chain(xt)
        cur.wait_stream(side)
    res = {}
    for label, fn in (("text alone", lambda: text_chain(xt)), ("wgrad alone", lambda: wg_batch(which, reps)), ("serial", serial), ("concurrent", conc)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(); fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                fn()
            gr.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): gr.replay()
            torch.cuda.synchronize()
            res[label] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: text chain alone {res['text alone']:.2f} ms | weight gradients alone {res['wgrad alone']:.2f} ms | one after the other {res['serial']:.2f} ms | two branches of one graph {res['concurrent']:.2f} ms", flush=True)
run_case("stage-2 weight gradients (DMA-staged, 512-thread workgroups)", ops2, 6)
run_case("stage-0 weight gradients (register-staged, 256-thread workgroups)", ops0, 3)
