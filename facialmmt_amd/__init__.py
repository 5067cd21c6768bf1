import os as _os

# ROCm 7.0's HIP-graph "packet capture" path (taken from the third launch of a graph on) drops the effect of some
# nodes: gradients coming out of replayed graphs are wrong from the third replay (measured: bias gradients of nn.Linear
# zero or garbage; with the flag off, graph replay is bit-identical to eager launches and no slower).  The runtime reads
# the flag when it initialises, so it has to be in the environment before the first HIP call.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
