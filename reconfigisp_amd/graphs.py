"""hipGraph capture of a fixed-shape inference forward.

A 256x256 x 64 batch of the element-wise pipeline is ~37 us of GPU work - less than what the
Python interpreter needs to walk the stage list and allocate the outputs - so serving loops
(``test_split.py`` tiles, the benchmark) replay a captured graph instead: one host call per forward,
the same kernels, the same buffers.  Capture goes through ``torch.cuda.CUDAGraph`` (hipGraph on
ROCm); the library's launches land on the capturing stream because every C-ABI call takes
``torch.cuda.current_stream()``.
"""
import torch


class GraphedQueue:
    """Serving form: ONE graph replay runs the network over a queue of ``len(examples)`` resident batches
    (distinct static input slots, distinct output buffers), so the per-replay launch latency (~8 us on this
    stack, more than 10 % of a 55 us forward) is paid once per queue, not once per batch.

    ``q = GraphedQueue(net, [b0, b1, b2, b3]); outs = q()`` -> list of network outputs, ``q.stage_outputs[k]``
    the per-stage outputs of slot k.  ``q.load(k, x)`` copies new data into slot k."""

    def __init__(self, net, examples, warmup=2):
        self.net = net
        self.slots = [e.detach().clone() for e in examples]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):
                net(self.slots[0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.outputs, self.stage_outputs = [], []
        with torch.no_grad(), torch.cuda.graph(self.graph):
            for slot in self.slots:
                self.outputs.append(net(slot))
                self.stage_outputs.append(list(getattr(net, 'intermediate_results', [])))

    def load(self, k, x):
        self.slots[k].copy_(x, non_blocking=True)

    def __call__(self):
        self.graph.replay()
        return self.outputs


class GraphedForward:
    """``g = GraphedForward(net, example); y = g(x)`` - x must keep the example's shape.

    ``g.outputs`` is the network output, ``g.intermediate_results`` the per-stage outputs (static
    buffers, overwritten by the next replay)."""

    def __init__(self, net, example, warmup=2):
        if not example.is_cuda:
            raise RuntimeError('GraphedForward needs a CUDA/HIP example tensor')
        self.net = net
        self.static_in = example.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                       # builds weight packs / parameter blocks
                net(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = net(self.static_in)
            self.intermediate_results = list(getattr(net, 'intermediate_results', []))

    def __call__(self, x=None):
        if x is not None and x.data_ptr() != self.static_in.data_ptr():
            if x.shape != self.static_in.shape:
                raise ValueError('captured for shape %s, got %s' % (tuple(self.static_in.shape), tuple(x.shape)))
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.outputs
