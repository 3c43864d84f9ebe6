"""Device-resident chaining of message-passing layers.

A thin mirror of the part of network_type that the msgpass examples use: `add(layer, input_list=,
operator=)` (athena_network_sub.f90:764-900), a forward that feeds every layer the 'concatenate' (||)
or 'add' (+) merge of its parents' outputs, the matching reverse pass, and `update` (:2816-2929).
Activations between layers stay in HBM: the only host transfers are the caller's input and whatever
the caller reads back.  Parents are merged in ascending order of their position in the network
(the reference packs the adjacency column in index order), 0 being the network input -- so
`input_list=[0, -1]` of example/msgpass_euler/src/main.f90:192-255 puts the input features first.
"""
import torch

from . import ops, optim


class network_type:
    def __init__(self):
        self.layers = []      # layer k (1-based id) = self.layers[k-1]
        self.parents = []     # per layer: sorted parent ids, 0 = network input
        self.operator = []    # per layer: 1 concatenate, 2 add
        self.optimiser = None

    def add(self, layer, input_list=None, operator="concatenate"):
        k = len(self.layers) + 1
        op = {"||": 1, "concat": 1, "concatenate": 1, "append": 1, 1: 1, "+": 2, "add": 2, 2: 2}.get(
            operator.lower() if isinstance(operator, str) else operator)
        if op is None:
            raise ValueError("invalid operator")                            # stop_program("invalid operator") :824-827
        if input_list is None:
            parents = [k - 1]
        else:
            parents = []
            for i in input_list:
                if i <= -k or i >= k:
                    raise ValueError(f"input vertex index {i} out of range ({-k + 1}:{k - 1})")
                parents.append(k + i if i < 0 else i)
        self.layers.append(layer)
        self.parents.append(sorted(set(parents)))
        self.operator.append(op)
        return self

    def compile(self, optimiser):
        self.optimiser = optimiser
        return self

    def set_graph(self, graph):
        for l in self.layers:
            l.set_graph(graph)

    def get_num_params(self):
        return sum(l.get_num_params() for l in self.layers)

    @property
    def num_layers(self):
        """the reference counts the input layer it inserts in front of the first added layer
        (test_msgpass_network.f90:55: one msgpass layer -> num_layers = 2)"""
        return len(self.layers) + 1 if self.layers else 0

    def get_params(self):
        """flat parameter vector, layers in order (network%get_params)"""
        import numpy as np
        parts = [l.get_params() for l in self.layers if l.get_num_params()]
        return np.concatenate(parts).astype(np.float32) if parts else np.zeros(0, np.float32)

    def set_params(self, params):
        import numpy as np
        params = np.asarray(params, np.float32)
        if params.size != self.get_num_params():
            raise ValueError("set_params: wrong number of parameters")
        o = 0
        for l in self.layers:
            n = l.get_num_params()
            if n:
                l.set_params(params[o:o + n])
                o += n

    def reset(self):
        """network%reset: back to an empty network (test_msgpass_network.f90:207-211)"""
        self.layers, self.parents, self.operator = [], [], []
        self.optimiser = None
        self._out = None
        self._captured = None

    # -------------------------------------------------------------------------------------------
    def forward(self, x, edge_features=None):
        """edge_features: the input graphs' edge features, handed to every layer that takes them (Duvenaud / GNO;
        the reference forwards input(2,s) alongside the vertex features, athena_msgpass_layer_sub.f90:184-198)"""
        x = self.layers[0]._cat(x)
        e = self.layers[0]._cat(edge_features) if edge_features is not None else None
        self._out = [x]
        self._widths = []
        for k, layer in enumerate(self.layers, start=1):
            srcs = [self._out[p] for p in self.parents[k - 1]]
            inp = srcs[0]
            if self.operator[k - 1] == 1:
                for s in srcs[1:]:
                    inp = ops.concat_features(inp, s)
            elif len(srcs) > 1:
                inp = inp.clone()
                for s in srcs[1:]:
                    ops.axpy(1.0, s, inp)
            self._widths.append([s.shape[1] for s in srcs])
            self._out.append(layer.forward(inp, e) if layer._needs_edges else layer.forward(inp))
        return self._out[-1]

    def backward(self, upstream, need_input_grad=False):
        grads = [None] * (len(self.layers) + 1)
        grads[-1] = self.layers[-1]._t(upstream)

        def give(p, g):
            if grads[p] is None:
                grads[p] = g
            else:
                ops.axpy(1.0, g, grads[p])

        for k in range(len(self.layers), 0, -1):
            if grads[k] is None:
                continue
            parents = self.parents[k - 1]
            need = [p > 0 or need_input_grad for p in parents]
            dinp = self.layers[k - 1].backward(grads[k], need_input_grad=any(need))
            grads[k] = None
            if dinp is None:
                continue
            if self.operator[k - 1] == 2 or len(parents) == 1:
                for p, n in zip(parents, need):
                    if n:
                        give(p, dinp if len(parents) == 1 else dinp.clone())
                continue
            # split a chained concatenation back to its parents, last parent first
            widths = self._widths[k - 1]
            rest = dinp
            for j in range(len(parents) - 1, 0, -1):
                left = sum(widths[:j])
                rest, piece = ops.concat_features_bwd(rest, left, need_a=True, need_b=need[j])
                if need[j]:
                    give(parents[j], piece)
            if need[0]:
                give(parents[0], rest)
        return grads[0]

    # -------------------------------------------------------------------------------------------
    def capture_step(self, x, target, loss, edge_features=None):
        """Record forward -> loss -> reverse pass of this network once into a HIP graph and return
        (replay, x_buf, target_buf, loss_buf).  Small graphs (the msgpass_euler mesh, molecule mini-batches)
        are launch bound: a step is ~100 short kernels, and replaying the captured graph removes the per-launch
        and per-op host cost.  New samples are fed by copying into x_buf / target_buf; gradients land in the
        layers' `.grads` buffers as usual, so `update()` (whose Adam bias correction changes every step) is
        called eagerly after replay().  The graph handles must be set before capturing.  With edge_features
        (Duvenaud / GNO networks) the captured edge buffer is `self.captured_edge_buf`."""
        dev = self.layers[0].device
        x_buf = self.layers[0]._cat(x).clone()
        t_buf = self.layers[0]._t(target).clone()
        e_buf = self.layers[0]._cat(edge_features).clone() if edge_features is not None else None
        self.captured_edge_buf = e_buf

        def run():
            out = self.forward(x_buf, e_buf)
            l, d = loss.compute(out, t_buf)
            self.backward(d)
            return l

        run()                                   # eager warm-up: grows the library's workspaces outside the capture
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                loss_buf = run()
        torch.cuda.current_stream(dev).wait_stream(side)
        grads = [[g for g in l.grads] for l in self.layers]     # the captured step always writes these buffers

        def replay():
            graph.replay()
            for l, gs in zip(self.layers, grads):
                l.grads = list(gs)
            return loss_buf

        self._captured = graph
        return replay, x_buf, t_buf, loss_buf

    def update(self, epoch=None):
        if self.optimiser is None:
            raise RuntimeError("No optimiser is defined for the network")   # :1713-1716
        return optim.update(self.layers, self.optimiser, epoch)
