"""Host-side mirrors of athena's three message-passing layer types, driving the HIP path.

Same names, constructor arguments, parameter layout and accessor semantics as the reference:
  kipf_msgpass_layer_type       athena_kipf_msgpass_layer.f90:78-98   (update_message :915-959)
  duvenaud_msgpass_layer_type   athena_duvenaud_msgpass_layer.f90:86-121 (:755-859)
  graph_nop_layer_type          athena_graph_nop_layer.f90:78-104     (:690-788)
  get_params / set_params / get_gradients / set_gradients   athena_base_layer_sub.f90:545-691
  set_graph                     athena_msgpass_layer_sub.f90:144-174

Differences that are deliberate: a batch of graphs is ONE block-diagonal device CSR
(the reference loops over samples on the host, athena_kipf_msgpass_layer.f90:940); `set_graph`
caches the device handle instead of copying the CSR every forward; and the
reverse pass is an explicit `backward(upstream)` that runs exactly the `get_partial_*_val`
callbacks diffstruc's `grad_reverse` would invoke for this layer, writing
`params(i)%grad` (here `self.grads[i]`).

No CPU fallback: every op goes through libathena_mp.so.
"""
import os

import numpy as np
import torch

from . import ops
from .graph import DeviceGraph, graph_type


def _fusable(activation):
    """can the activation ride in a GEMM epilogue (plain none / relu / sigmoid / tanh)?"""
    return isinstance(activation, str) and activation in ops.ACT


def _identity(activation):
    return isinstance(activation, str) and activation in ("none", "linear")


def _expand(nf, T, what):
    nf = list(np.atleast_1d(nf))
    if len(nf) == 1:
        return [int(nf[0])] * (T + 1)
    if len(nf) == T + 1:
        return [int(v) for v in nf]
    # athena_kipf_msgpass_layer.f90:271-274
    raise ValueError(f"Error: {what} must be a scalar or a vector of length num_time_steps + 1")


def _init_weights(rng, n, fan_in, fan_out, activation, initialiser=None):
    """default initialiser: he_normal for relu-family else glorot_uniform
    (docs/source/layers/msgpass/kipf_msgpass_layer.rst:61-65); by name: athena_initialiser_{glorot,he,lecun,
    zeros,ones}.f90 (limits / sigmas as at _glorot.f90:120,178, _he.f90:167,229, _lecun.f90:117,172).  The
    random stream is numpy's, not the reference's -- distributions match, draws do not."""
    if n == 0:                      # zero-width tensors (the placeholder layers the reference's readers build)
        return np.zeros(0, np.float32)
    name = initialiser
    if name is None:
        name = "he_normal" if getattr(activation, "name", activation) in ("relu", "leaky_relu", "swish", "selu") else "glorot_uniform"
    uniform = {"glorot_uniform": 6.0 / (fan_in + fan_out), "he_uniform": 6.0 / fan_in, "lecun_uniform": 3.0 / fan_in}
    normal = {"glorot_normal": 2.0 / (fan_in + fan_out), "he_normal": 2.0 / fan_in, "lecun_normal": 1.0 / fan_in}
    if name in uniform:
        lim = np.sqrt(uniform[name])
        return rng.uniform(-lim, lim, n).astype(np.float32)
    if name in normal:
        return (rng.standard_normal(n) * np.sqrt(normal[name])).astype(np.float32)
    if name in ("zeros", "ones"):
        return np.full(n, 0.0 if name == "zeros" else 1.0, np.float32)
    raise ValueError(f"unknown initialiser '{name}'")


class _batched_graph:
    """block-diagonal concatenation of a batch of graph_type (vertex / edge-id offsets)"""

    def __init__(self, graphs, device_index, keep_edges):
        graphs = list(graphs)
        nv = [g.num_vertices for g in graphs]
        ne = [max(g.num_edges, int(g.adj_ja[1].max()) if g.nnz else 0) for g in graphs]
        self.vertex_offsets = np.concatenate([[0], np.cumsum(nv)]).astype(np.int32)
        self.edge_offsets = np.concatenate([[0], np.cumsum(ne)]).astype(np.int32)
        ia = [np.ones(1, np.int64)]
        ja = []
        base = 0
        for s, g in enumerate(graphs):
            ia.append(g.adj_ia[1:].astype(np.int64) + base)
            j = g.adj_ja.astype(np.int64).copy()
            j[0] += self.vertex_offsets[s]
            j[1] = np.where(j[1] > 0, j[1] + self.edge_offsets[s], 0)
            ja.append(j)
            base += g.nnz
        self.adj_ia = np.concatenate(ia).astype(np.int32)
        self.adj_ja = np.asfortranarray(np.concatenate(ja, axis=1) if ja else np.zeros((2, 0)), dtype=np.int32)
        self.num_vertices = int(self.vertex_offsets[-1])
        self.num_edges = int(self.edge_offsets[-1])
        self.batch = len(graphs)
        # shared: the layers of a network are all handed the same batch (athena_network_sub.f90:2727-2730) -- one set of
        # device arrays behind all of them (athena_mp_graph_acquire keys on the content of the assembled CSR)
        self.device = DeviceGraph(self.adj_ia, self.adj_ja, n_edge_cols=self.num_edges if keep_edges else 0,
                                  device=device_index, shared=True)


class msgpass_layer_type:
    """abstract base: athena_msgpass_layer.f90:19-76"""

    name = "msgp"
    _needs_edges = False

    def __init__(self, device="cuda:0", seed=0):
        self.device = torch.device(device)
        self._rng = np.random.default_rng(seed)
        self.params = []   # flat float32 device tensors == params(i)%val(:,1)
        self.grads = []    # params(i)%grad%val(:,1) or None
        self.graph = None
        self.output = None
        self.inference = False   # base_layer_type%inference (athena_base_layer.f90:48; network%set_training / inference,
                                 # athena_network_sub.f90:1882-1915): a forward pass that no reverse pass follows

    # -- graph ---------------------------------------------------------------------------------
    def _ensure_graph(self, graphs):
        """the block-diagonal device graph of `graphs`, built once and cached; True when it was (re)built"""
        # the cache is keyed on CONTENT (object, version, sizes, checksum of adj_ia / adj_ja), and the layer holds the
        # graph objects so that an id cannot be recycled for a new graph while the key is alive
        key = tuple(g.topology_key() for g in graphs)
        old = getattr(self, "_graph_key", None)
        if old == key:
            return False
        if (self.graph is not None and old is not None and len(old) == len(key)
                and all(a[0] == b[0] and a[2:] == b[2:] for a, b in zip(old, key))):
            # same objects, same sizes, same content key, newer version: graph_type.touch() (or a re-assignment)
            # says the arrays changed although the key does not -- the cached handle must not be found again
            self.graph.device.evict()
        self.graph = _batched_graph(graphs, self.device.index or 0, self._needs_edges)
        self._graph_key = key
        self._graph_refs = list(graphs)
        self._own_offsets = self.graph.vertex_offsets
        self._cut = None
        return True

    def _set_cut(self, vo, batched):
        """where each sample's vertices start: the list's own offsets (set_graph) or the caller's (set_graph_batched).  Part of
        the layer's graph state beside the handle's key: the same cached graph can be cut either way by consecutive calls."""
        cut = getattr(self, "_cut", None)
        if cut is not None and cut[0] == batched and (cut[1] is vo or np.array_equal(cut[1], vo)):
            return
        self.graph.vertex_offsets = vo
        self.graph.batch = int(vo.size - 1)
        self._seg = torch.from_numpy(vo).to(self.device)
        self._cut = (batched, vo)

    def set_graph(self, graphs):
        """athena_msgpass_layer_sub.f90:144-174; here the device handle is built once and cached"""
        if isinstance(graphs, graph_type):
            graphs = [graphs]
        self._ensure_graph(graphs)
        self._set_cut(self._own_offsets, False)
        return self

    def set_graph_batched(self, graph, vertex_offsets):
        """A batch that is ALREADY one block-diagonal graph (what set_graph assembles from a list): `graph` holds every sample's
        vertices and edge columns back to back, vertex_offsets [batch + 1] (0-based) says where each sample's vertices start.
        Same layer, same kernels; the per-sample graph objects never exist (130 000 of them at BASELINE configs[2])."""
        vo = np.ascontiguousarray(vertex_offsets, dtype=np.int32)
        if not (vo.ndim == 1 and vo.size >= 2 and vo[0] == 0 and vo[-1] == graph.num_vertices and np.all(np.diff(vo) >= 0)):
            raise ValueError("set_graph_batched: vertex_offsets must run from 0 to graph.num_vertices, ascending")
        self._ensure_graph([graph])
        self._set_cut(vo.copy(), True)
        return self

    # -- text card of the network file (print_base / read) -------------------------------------------
    def print(self, file=None):
        """the layer's card as athena writes it (athena_base_layer_sub_io.f90:14-65); appended to `file` if given"""
        from . import io
        card = io.layer_card(self)
        if file is not None:
            with open(file, "a") as f:
                f.write(card)
        return card

    # -- learnable accessors (athena_base_layer_sub.f90:545-691) ----------------------------------
    def get_num_params(self):
        return int(sum(p.numel() for p in self.params))

    @property
    def num_params(self):
        return self.get_num_params()

    def get_params(self):
        return np.concatenate([p.detach().cpu().numpy() for p in self.params]).astype(np.float32)

    def set_params(self, params):
        params = np.asarray(params, np.float32)
        if params.size != self.get_num_params():
            raise ValueError("set_params: wrong number of parameters")
        o = 0
        for p in self.params:
            p.copy_(torch.from_numpy(params[o:o + p.numel()]).to(self.device))
            o += p.numel()

    def get_gradients(self):
        out = []
        for p, g in zip(self.params, self.grads):
            out.append(np.zeros(p.numel(), np.float32) if g is None else g.detach().cpu().numpy())  # :627-631
        return np.concatenate(out).astype(np.float32)

    def set_gradients(self, gradients):
        if np.ndim(gradients) == 0:                                   # rank(0) branch :667-673
            self.grads = [torch.full_like(p, float(gradients)) for p in self.params]
            return
        gradients = np.asarray(gradients, np.float32)
        o = 0
        for i, p in enumerate(self.params):
            self.grads[i] = torch.from_numpy(gradients[o:o + p.numel()].copy()).to(self.device)
            o += p.numel()

    # -- merging replicas (reduce_learnable / add_learnable, athena_base_layer_sub.f90:468-540) -------------
    def reduce(self, other):
        """this = this + other: parameters summed, gradients summed where both layers hold one -- what the reference
        uses to merge per-thread replicas of a layer; on the device, one axpy per tensor"""
        if len(self.params) != len(other.params):
            raise ValueError("reduce_learnable: incompatible parameter sizes")
        for i in range(len(self.params)):
            if self.params[i].numel() != other.params[i].numel():
                raise ValueError("reduce_learnable: incompatible parameter sizes")
            ops.axpy(1.0, other.params[i].to(self.device), self.params[i])
            if self.grads[i] is not None and other.grads[i] is not None:
                ops.axpy(1.0, other.grads[i].to(self.device), self.grads[i])
        return self

    def __add__(self, other):
        """add_learnable: a copy of this layer (same hyperparameters, graph handle shared) reduced with other"""
        import copy
        out = copy.copy(self)
        out.params = [p.clone() for p in self.params]
        out.grads = [None if g is None else g.clone() for g in self.grads]
        return out.reduce(other)

    def _t(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)

    def _cat(self, a):
        return self._t(np.concatenate([np.asarray(v, np.float32) for v in a], axis=0)) if isinstance(a, (list, tuple)) else self._t(a)

    def forward(self, vertex_features, edge_features=None):
        """forward_msgpass = update_message + update_readout (athena_msgpass_layer_sub.f90:184-198)"""
        if self.graph is None:
            raise RuntimeError("set_graph must be called before forward")
        x = self._cat(vertex_features)
        e = self._cat(edge_features) if edge_features is not None else None
        self.update_message(x, e)
        self.update_readout()
        return self.output


# ==================================================================================================
class kipf_msgpass_layer_type(msgpass_layer_type):
    name = "kipf"

    def __init__(self, num_vertex_features, num_time_steps, activation="none", kernel_initialiser=None,
                 verbose=0, device="cuda:0", seed=0, order="auto"):
        """order: which of the two associations of W.(A X) a time step evaluates --
        "aggregate_first" (the reference's: P = A X, then the dense step), "transform_first" (Y = X W^T, then
        A Y: the aggregation runs on F_out-wide rows) or "auto" (transform first when 4 F_out <= 3 F_in).
        Same result up to fp32 summation order."""
        super().__init__(device, seed)
        if not (order in ("auto", "aggregate_first", "transform_first")):
            raise ValueError('expected: order in ("auto", "aggregate_first", "transform_first")')
        self.order = order
        self.num_time_steps = int(num_time_steps)
        self.num_vertex_features = _expand(num_vertex_features, self.num_time_steps, "num_vertex_features")
        self.num_edge_features = [0] * (self.num_time_steps + 1)
        self.activation = ops.resolve_activation(activation or "none")   # a name or an ops.actv_type
        self.use_graph_output = True
        # init_kipf :345-380 -- params(t): W(F_t, F_{t-1}) flat column-major
        for t in range(1, self.num_time_steps + 1):
            fi, fo = self.num_vertex_features[t - 1], self.num_vertex_features[t]
            self.params.append(self._t(_init_weights(self._rng, fo * fi, fi, fo, self.activation, kernel_initialiser)))
        self.grads = [None] * len(self.params)
        if verbose:
            print(f"KIPF activation function: {self.activation}")

    def update_message(self, x, e=None):
        """athena_kipf_msgpass_layer.f90:915-959: X_t = act(W_t . kipf_propagate(X_{t-1}))"""
        g = self.graph.device
        if not (x.shape == (g.n_cols, self.num_vertex_features[0])):
            raise ValueError("vertex feature shape mismatch")
        self._tape = []
        cur = x
        for t in range(1, self.num_time_steps + 1):
            if self._transform_first(t):
                # dense step first: the gather then moves F_t-wide rows instead of F_{t-1}-wide ones
                y = ops.matmul(self.params[t - 1], cur, self.num_vertex_features[t])
                if _fusable(self.activation):     # plain activation: in the aggregation's store
                    z = None
                    nxt = ops.kipf_propagate_act(g, y, act=self.activation)
                else:
                    z = ops.kipf_propagate(g, y)
                    nxt = ops.activation(self.activation, z)
                self._tape.append((cur, nxt, z if ops.needs_input(self.activation) else None))
                cur = nxt
                continue
            if _fusable(self.activation):
                # aggregation + dense step (+ activation) in one launch where the fused kernel exists
                p, nxt = ops.kipf_layer_fwd(g, cur, self.params[t - 1], self.num_vertex_features[t], act=self.activation)
                z = None
            else:
                # 'softmax' / 'swish' (msgpass_euler) and the activations with attributes run as their own launch;
                # the pre-activation is kept for the ones that differentiate at their input
                p, z = ops.kipf_layer_fwd(g, cur, self.params[t - 1], self.num_vertex_features[t], act="none")
                nxt = ops.activation(self.activation, z)
                if not ops.needs_input(self.activation):
                    z = None
            self._tape.append((p, nxt, z))
            cur = nxt
        self.output = cur

    def _transform_first(self, t):
        fi, fo = self.num_vertex_features[t - 1], self.num_vertex_features[t]
        g = self.graph.device
        if g.n_rows != g.n_cols:      # a row shard gathers halo rows: the layer step of athena_amd.dist decides there
            return False
        return self.order == "transform_first" or (self.order == "auto" and 4 * fo <= 3 * fi)

    def update_readout(self):
        pass  # node-level output (update_readout_kipf :964-971)

    def backward(self, upstream, need_input_grad=True, exact=False):
        """reverse pass: activation -> matmul (dW, dP) -> get_partial_kipf_propagate_left_val.
        exact=False reproduces the reference's coefficient-free scatter."""
        g = self.graph.device
        gcur = self._t(upstream)
        for t in range(self.num_time_steps, 0, -1):
            p, out, z = self._tape[t - 1]
            dz = ops.activation_bwd(self.activation, out, gcur, z=z) if not _identity(self.activation) else gcur
            if self._transform_first(t):
                # Z = A (X W^T):  dW = (A^T dZ)^T X with the coefficient, dX = (scatter of dZ) W in the reference's
                # coefficient-free form -- both sums from one gather of the dZ rows (p holds the step's input X)
                if t == 1 and not need_input_grad:
                    self.grads[t - 1] = ops.matmul_dw(p, ops.kipf_propagate_bwd(g, dz, exact=True))
                    return None
                if exact:
                    q_plain = q_coef = ops.kipf_propagate_bwd(g, dz, exact=True)
                else:
                    q_plain, q_coef = ops.kipf_propagate_bwd_dual(g, dz)
                self.grads[t - 1] = ops.matmul_dw(p, q_coef)
                gcur = ops.matmul_dx(self.params[t - 1], q_plain, self.num_vertex_features[t - 1])
                continue
            dw = ops.matmul_dw(p, dz)
            self.grads[t - 1] = dw
            if t == 1 and not need_input_grad:   # input layer output has requires_grad = .false.
                return None
            gcur = ops.kipf_layer_bwd_x(g, dz, self.params[t - 1], self.num_vertex_features[t - 1], exact=exact)
        return gcur


# ==================================================================================================
class duvenaud_msgpass_layer_type(msgpass_layer_type):
    name = "duvenaud"
    _needs_edges = True

    def __init__(self, num_vertex_features, num_edge_features, num_time_steps, max_vertex_degree, num_outputs,
                 min_vertex_degree=1, message_activation="sigmoid", readout_activation="softmax",
                 kernel_initialiser=None, verbose=0, device="cuda:0", seed=0):
        super().__init__(device, seed)
        self.num_time_steps = T = int(num_time_steps)
        self.num_vertex_features = _expand(num_vertex_features, T, "num_vertex_features")
        self.num_edge_features = _expand(num_edge_features, T, "num_edge_features")
        self.min_vertex_degree, self.max_vertex_degree = int(min_vertex_degree), int(max_vertex_degree)
        self.num_outputs = int(num_outputs)
        self.activation = ops.resolve_activation(message_activation or "sigmoid")       # :123
        self.activation_readout = ops.resolve_activation(readout_activation or "softmax")  # :124
        self.use_graph_output = False
        D = self.max_vertex_degree - self.min_vertex_degree + 1
        Fe = self.num_edge_features[0]
        # init_duvenaud :547-589 -- T message tensors W(F_t, F_{t-1}+Fe, D), then T readout R(num_outputs, F_t)
        for t in range(1, T + 1):
            fi, fo = self.num_vertex_features[t - 1] + Fe, self.num_vertex_features[t]
            self.params.append(self._t(_init_weights(self._rng, fo * fi * D, fi, fo, self.activation, kernel_initialiser)))
        for t in range(1, T + 1):
            fv = self.num_vertex_features[t]
            self.params.append(self._t(_init_weights(self._rng, self.num_outputs * fv, sum(self.num_vertex_features), self.num_outputs,
                                                     self.activation, kernel_initialiser)))
        self.grads = [None] * len(self.params)

    def _split_a(self):
        """a = [a_x | a_e] kept split for the whole layer: every time step in the widths of the split launches (F_v = 64 throughout,
        F_e <= 16, at most 16 outputs, fused message activation, softmax readout).  The edge features do not change from time step
        to time step (update_message_duvenaud, :755-817, passes the same edge_features to every duvenaud_propagate), so their
        neighbour sums a_e are gathered ONCE per forward pass; each time step gathers the vertex part alone into 256-byte rows."""
        return (_fusable(self.activation) and self.activation_readout == "softmax" and self.num_outputs <= 16
                and all(v == 64 for v in self.num_vertex_features) and 0 < self.num_edge_features[0] <= 16)

    def update_message(self, x, e):
        """athena_duvenaud_msgpass_layer.f90:755-817"""
        g = self.graph.device
        T = self.num_time_steps
        if not (e is not None and e.shape[0] == g.n_edge_cols):
            raise ValueError("edge feature shape mismatch")
        self._e = e
        self._a, self.z, self._c = [], [], []
        self._p_early = [None] * T            # the readout's per-vertex p where it rode in the update launch
        self._a_e = ops.duvenaud_propagate_edges(g, e) if self._split_a() else None
        cur = x
        for t in range(1, T + 1):
            if self._a_e is not None:
                a = ops.neighbour_sum(g, cur)
                zt, self._p_early[t - 1] = ops.duvenaud_update_act_readout_split(
                    g, a, self._a_e, self.params[t - 1], self.min_vertex_degree, self.max_vertex_degree, 64,
                    self.params[T + t - 1], self.num_outputs, act=self.activation)
                self._a.append(a)
                self.z.append(zt)
                self._c.append(None)
                cur = zt
                continue
            a = ops.duvenaud_propagate(g, cur, e)
            if _fusable(self.activation) and self.activation_readout == "softmax" and self.num_outputs <= 16:
                # the bucket contraction with the activation AND the readout's softmax(R z) in its epilogue: the readout of
                # this time step then only sums p per graph (update_readout below)
                zt, self._p_early[t - 1] = ops.duvenaud_update_act_readout(
                    g, a, self.params[t - 1], self.min_vertex_degree, self.max_vertex_degree, self.num_vertex_features[t],
                    self.params[T + t - 1], self.num_outputs, act=self.activation)
                c = None
            elif _fusable(self.activation):   # the bucket contraction with the activation in its epilogue
                zt = ops.duvenaud_update_act(g, a, self.params[t - 1], self.min_vertex_degree, self.max_vertex_degree,
                                             self.num_vertex_features[t], act=self.activation)
                c = None
            else:                             # shaped / attributed activations: their own launch
                c = ops.duvenaud_update_act(g, a, self.params[t - 1], self.min_vertex_degree, self.max_vertex_degree,
                                            self.num_vertex_features[t], act="none")
                zt = ops.activation(self.activation, c)
                if not ops.needs_input(self.activation):
                    c = None
            self._a.append(a)
            self.z.append(zt)
            self._c.append(c)
            cur = zt

    def update_readout(self):
        """athena_duvenaud_msgpass_layer.f90:822-859: out[:,s] = sum_t sum_{v in s} act_readout(R_t z_t[:,v])"""
        T = self.num_time_steps
        out = None
        self._p = []
        for t in range(1, T + 1):
            if self.activation_readout == "softmax" and getattr(self, "_p_early", [None] * T)[t - 1] is not None:
                p = self._p_early[t - 1]                  # computed beside z in update_message: only the per-graph sums are left
                out = ops.segment_sum(p, self._seg, out=out)
            elif self.activation_readout == "softmax":    # one launch incl. the logits contraction
                p, out = ops.duvenaud_readout(self.params[T + t - 1], self.z[t - 1], self._seg, self.num_outputs, out=out)
            else:                                         # any other readout activation: op by op
                logits = ops.matmul(self.params[T + t - 1], self.z[t - 1], self.num_outputs)
                p = ops.activation(self.activation_readout, logits) if not _identity(self.activation_readout) else logits
                out = ops.segment_sum(p, self._seg, out=out)
                if ops.needs_input(self.activation_readout):
                    p = (p, logits)
            self._p.append(p)
        self.output = out   # [batch, num_outputs]

    def backward(self, upstream, need_input_grad=True, need_edge_grad=False):
        g = self.graph.device
        T = self.num_time_steps
        gout = self._t(upstream)
        if not (gout.shape == (self.graph.batch, self.num_outputs)):
            raise ValueError('expected: gout.shape == (self.graph.batch, self.num_outputs)')
        dz_next = None
        de = None
        dx = None
        da_e_sum = None
        for t in range(T, 0, -1):
            # readout branch of step t + the gradient arriving from step t+1, through the message activation
            fused_act = _fusable(self.activation)
            Fv = self.num_vertex_features[t - 1]
            split_a = getattr(self, "_a_e", None) is not None
            if split_a or (self.activation_readout == "softmax" and fused_act and Fv == 64 and self.num_vertex_features[t] == 64
                           and self.num_edge_features[0] > 0 and (t > 1 or need_input_grad or need_edge_grad)):
                # the readout's reverse and the update's reverse of this time step in ONE call: dc [n, 64] never reaches HBM where
                # the library's fused launch covers the shape (profiles/r05_c3_readout_update_fused_ab.txt: 0.825 -> 0.66 ms)
                # ... and the edge part of da is SUMMED over these time steps (the scatter to the edge features is linear in it):
                # one duvenaud_propagate_bwd_e after the loop instead of one + an axpy per time step
                da_x, da_e_sum, self.grads[t - 1], self.grads[T + t - 1] = ops.duvenaud_readout_update_bwd(
                    g, self.params[T + t - 1], self.z[t - 1], self._p[t - 1], self._seg, gout, self._a[t - 1], self.params[t - 1],
                    self.min_vertex_degree, self.max_vertex_degree, Fv, act=self.activation, dz_next=dz_next,
                    da_e=da_e_sum if need_edge_grad else None, a_e=self._a_e if split_a else None)
                if t > 1 or need_input_grad:
                    dz_next = ops.duvenaud_propagate_bwd_x(g, da_x, Fv)
                if t == 1:
                    dx = dz_next
                continue
            if self.activation_readout == "softmax":
                dc, self.grads[T + t - 1] = ops.duvenaud_readout_bwd(self.params[T + t - 1], self.z[t - 1], self._p[t - 1],
                                                                     self._seg, gout, dz_next=dz_next,
                                                                     act=self.activation if fused_act else "none")
                if not fused_act:
                    dc = ops.activation_bwd(self.activation, self.z[t - 1], dc, z=self._c[t - 1])
            else:
                p = self._p[t - 1]
                p, logits = p if isinstance(p, tuple) else (p, None)
                dl = ops.segment_sum_bwd(gout, self._seg, p.shape[0])
                if not _identity(self.activation_readout):
                    dl = ops.activation_bwd(self.activation_readout, p, dl, z=logits)
                self.grads[T + t - 1] = ops.matmul_dw(self.z[t - 1], dl)
                dz = ops.matmul_dx(self.params[T + t - 1], dl, self.num_vertex_features[t])
                if dz_next is not None:
                    ops.axpy(1.0, dz_next, dz)
                dc = ops.activation_bwd(self.activation, self.z[t - 1], dz, z=self._c[t - 1]) if not _identity(self.activation) else dz
            # message branch
            if t == 1 and not (need_input_grad or need_edge_grad):
                self.grads[t - 1] = ops.duvenaud_update_bwd_w(g, dc, self._a[t - 1], self.min_vertex_degree, self.max_vertex_degree)
                break
            # both reverse products of the update from one pass over dc (one launch where the fused kernel covers the widths)
            if Fv == 64 and self.num_edge_features[0] > 0:
                # da split where it is written (256-byte vertex rows + a dense edge part): the two propagate partials then
                # gather whole cache lines -- bit-identical dx / de (profiles/r05_c3_split_da_ab.txt)
                da_x, da_e, self.grads[t - 1] = ops.duvenaud_update_bwd_split(g, dc, self._a[t - 1], self.params[t - 1],
                                                                              self.min_vertex_degree, self.max_vertex_degree, Fv)
                fv_e = 0
            else:
                da, self.grads[t - 1] = ops.duvenaud_update_bwd(g, dc, self._a[t - 1], self.params[t - 1], self.min_vertex_degree,
                                                                self.max_vertex_degree)
                da_x = da_e = da
                fv_e = Fv
            if need_edge_grad:
                d = ops.duvenaud_propagate_bwd_e(g, da_e, fv_e)
                de = d if de is None else ops.axpy(1.0, d, de)
            if t > 1 or need_input_grad:
                dz_next = ops.duvenaud_propagate_bwd_x(g, da_x, Fv)
            if t == 1:
                dx = dz_next
        if need_edge_grad and da_e_sum is not None:
            d = ops.duvenaud_propagate_bwd_e(g, da_e_sum, 0)
            de = d if de is None else ops.axpy(1.0, d, de)
        return (dx, de) if need_edge_grad else dx


# ==================================================================================================
class graph_nop_layer_type(msgpass_layer_type):
    name = "graph_nop"
    _needs_edges = True

    def __init__(self, num_outputs, coord_dim, kernel_hidden=None, num_inputs=None, use_bias=True,
                 activation="none", kernel_initialiser=None, bias_initialiser=None, verbose=0,
                 device="cuda:0", seed=0, keep_s=None):
        super().__init__(device, seed)
        # keep_s: the forward pass keeps S = sum_e [h_e;1] x_j^T per vertex for the reverse pass (the reference keeps every
        # forward intermediate on its tape).  None: whenever the shape takes the kernels that do and S is at most
        # ATHENA_MP_GNO_KEEP_S_MAX_GB (default 64: BASELINE configs[3] needs 33 of the 288 GB); True / False: always / never.
        self.keep_s = keep_s
        self._s_save = None       # the buffer, reused from step to step
        self._s_valid = False     # it holds the S of the forward pass the next backward pass belongs to
        self.num_outputs = int(num_outputs)
        self.coord_dim = int(coord_dim)
        self.kernel_hidden = int(kernel_hidden) if kernel_hidden else 16
        self.use_bias = bool(use_bias)
        self.activation = ops.resolve_activation(activation or "none")
        self.use_graph_output = True
        self.num_time_steps = 1
        if num_inputs is not None:
            self.init([int(num_inputs), 0])

    def init(self, input_shape):
        """init_gno :357-458 -- params(1) = [U(H,d) | b_u(H) | V(F,H) | b_v(F)], params(2) = W(F_out,F_in),
        params(3) = b(F_out) if use_bias"""
        Fi, Fo, d, H = int(input_shape[0]), self.num_outputs, self.coord_dim, self.kernel_hidden
        self.num_vertex_features = [Fi, Fo]
        F = Fo * Fi
        r = self._rng
        theta = np.concatenate([
            _init_weights(r, H * d, d, H, self.activation), np.zeros(H, np.float32),
            _init_weights(r, F * H, H, F, self.activation), np.zeros(F, np.float32)])
        self.params = [self._t(theta), self._t(_init_weights(r, Fo * Fi, Fi + int(self.use_bias), Fo, self.activation))]
        if self.use_bias:
            self.params.append(self._t(np.zeros(Fo, np.float32)))
        self.grads = [None] * len(self.params)

    def update_message(self, x, coords):
        """athena_graph_nop_layer.f90:690-788"""
        g = self.graph.device
        if not self.params:                 # num_inputs left to the network: taken from the upstream width
            self.init([int(x.shape[1]), 0])
        Fi, Fo = self.num_vertex_features
        if coords is None:
            raise RuntimeError("graph_nop layer expects vertex and edge feature inputs")   # :725-728
        if not (x.shape == (g.n_cols, Fi) and coords.shape == (g.n_edge_cols, self.coord_dim)):
            raise ValueError('expected: x.shape == (g.n_cols, Fi) and coords.shape == (g.n_edge_cols, self.coord_dim)')
        self._x, self._coords = x, coords
        self._s_valid = False
        m = None
        if self._keeps_s(g, Fi, Fo):
            try:
                m, self._s_save = ops.gno_aggregate_save(g, self.params[0], coords, x, self.coord_dim, self.kernel_hidden, Fo,
                                                         s_save=self._s_save)                            # steps 1+2, S kept
                self._s_valid = True
            except torch.OutOfMemoryError:
                if self.keep_s is True:
                    raise
                self.keep_s, self._s_save = False, None        # no room for S beside the model: rebuild it from now on
        if m is None:
            m = ops.gno_aggregate(g, self.params[0], coords, x, self.coord_dim, self.kernel_hidden, Fo)   # steps 1+2
        z = ops.matmul(self.params[1], x, Fo, bias=self.params[2] if self.use_bias else None)          # steps 3+5
        ops.axpy(1.0, m, z)                                                                            # step 4
        out = ops.activation(self.activation, z) if not _identity(self.activation) else z               # step 6
        self._z = z if ops.needs_input(self.activation) else None
        self.output = out
        self.output_edge = coords   # output(2,s): edge geometry forwarded without gradient (:781-785)

    def update_readout(self):
        pass

    def _keeps_s(self, g, Fi, Fo):
        if self.keep_s is False or self.inference:      # inference mode: no reverse pass will ask for S
            return False
        nbytes = ops.gno_saved_bytes(g, self.coord_dim, self.kernel_hidden, Fi, Fo)
        if nbytes == 0:
            if self.keep_s is True:
                raise ValueError("keep_s=True, but this shape does not take the kernels that keep S (H = F_in = F_out = 64, d <= 3)")
            return False
        return self.keep_s is True or nbytes <= float(os.environ.get("ATHENA_MP_GNO_KEEP_S_MAX_GB", "64")) * 1e9

    def backward(self, upstream, need_input_grad=True, need_coord_grad=False):
        g = self.graph.device
        Fi, Fo = self.num_vertex_features
        d, H = self.coord_dim, self.kernel_hidden
        gup = self._t(upstream)
        dz = ops.activation_bwd(self.activation, self.output, gup, z=self._z) if not _identity(self.activation) else gup
        if self.use_bias:
            ones = torch.ones((dz.shape[0], 1), device=self.device)
            self.grads[2] = ops.matmul_dw(ones, dz)          # db[o] = sum_v dz[v,o]
        self.grads[1] = ops.matmul_dw(self._x, dz)
        # the whole reverse pass of gno_aggregate from ONE G = dz . Vmat^T (athena_mp_gno_aggregate_bwd: the kernel that
        # holds a piece of G_i in LDS for the kernel MLP's gradient also emits every entry's partial of dx)
        dxa, self.grads[0], dc, _ = ops.gno_aggregate_bwd(g, self.params[0], self._coords, self._x, dz, d, H,
                                                          s_save=self._s_save if self._s_valid else None,
                                                          need_dx=need_input_grad, need_dcoords=need_coord_grad)
        dx = None
        if need_input_grad:
            dx = ops.matmul_dx(self.params[1], dz, Fi)
            ops.axpy(1.0, dxa, dx)
        return (dx, dc) if need_coord_grad else dx


# ==================================================================================================
class full_layer_type(msgpass_layer_type):
    """athena_full_layer.f90 -- the dense head the msgpass examples put after a graph-level readout
    (example/msgpass_chemical/src/main.f90:139-157): output = act(W . input + b) on [batch, num_inputs] rows
    (forward_full :835-875: matmul(params(1), input) + params(2), then the activation).  params(1) = W(num_outputs,
    num_inputs) flat column-major, params(2) = b (init_full :369-413).  One GEMM launch with the bias (and a plain
    activation) in its epilogue; activations with attributes run as their own launch on the kept pre-activation."""

    name = "full"

    def __init__(self, num_outputs, num_inputs=None, use_bias=True, activation="none", kernel_initialiser=None,
                 bias_initialiser=None, verbose=0, device="cuda:0", seed=0):
        super().__init__(device, seed)
        self.num_outputs, self.num_inputs = int(num_outputs), None
        self.use_bias = bool(use_bias)
        self.activation = ops.resolve_activation(activation or "none")
        self._kernel_initialiser, self._bias_initialiser = kernel_initialiser, bias_initialiser or "zeros"
        if num_inputs is not None:
            self._init(int(num_inputs))

    def _init(self, num_inputs):
        """init_full: the input width may come from the previous layer when the network is assembled"""
        self.num_inputs = num_inputs
        fi, fo = num_inputs, self.num_outputs
        self.params = [self._t(_init_weights(self._rng, fo * fi, fi, fo, self.activation, self._kernel_initialiser))]
        if self.use_bias:
            self.params.append(self._t(_init_weights(self._rng, fo, fi, fo, self.activation, self._bias_initialiser)))
        self.grads = [None] * len(self.params)

    def get_num_params(self):
        if self.num_inputs is None:
            return 0
        return (self.num_inputs + int(self.use_bias)) * self.num_outputs       # get_num_params_full :122-138

    def set_graph(self, graphs):
        return self          # no graph input

    def forward(self, x, edge_features=None):
        x = self._cat(x)
        if self.num_inputs is None:
            self._init(int(x.shape[1]))
        if not (x.dim() == 2 and x.shape[1] == self.num_inputs):
            raise ValueError("full layer: input width mismatch")
        self._x = x
        b = self.params[1] if self.use_bias else None
        if _fusable(self.activation):
            self._z = None
            self.output = ops.matmul(self.params[0], x, self.num_outputs, bias=b, act=self.activation)
        else:
            z = ops.matmul(self.params[0], x, self.num_outputs, bias=b)
            self.output = ops.activation(self.activation, z)
            self._z = z if ops.needs_input(self.activation) else None
        return self.output

    def backward(self, upstream, need_input_grad=True):
        gup = self._t(upstream)
        dz = ops.activation_bwd(self.activation, self.output, gup, z=self._z) if not _identity(self.activation) else gup
        self.grads[0] = ops.matmul_dw(self._x, dz)
        if self.use_bias:
            ones = torch.ones((dz.shape[0], 1), device=self.device)
            self.grads[1] = ops.matmul_dw(ones, dz)               # db[o] = sum over the batch
        if not need_input_grad:
            return None
        return ops.matmul_dx(self.params[0], dz, self.num_inputs)


def read_layer(text, device="cuda:0"):
    """layer from its text card (read_kipf_msgpass_layer / read_graph_nop_layer); weights are set as read"""
    from . import io
    name, hp, weights = io.parse_layer_card(text)
    act = io.activation_from_card(hp)
    if name == "duvenaud":
        # read_duvenaud_msgpass_layer (athena_duvenaud_msgpass_layer.f90:719-745) builds this placeholder and calls
        # read_duvenaud, which is an empty stub (:703-714): the card's contents are NOT loaded.  Mirrored as is --
        # the card carries neither the degree range nor num_outputs, so it could not rebuild the layer anyway.
        import warnings
        warnings.warn("DUVENAUD card: the reference's read_duvenaud is an empty stub; returning its placeholder layer")
        return duvenaud_msgpass_layer_type(num_vertex_features=[0], num_edge_features=[0], num_time_steps=1,
                                           max_vertex_degree=1, num_outputs=1, device=device)
    if name == "full":
        layer = full_layer_type(num_outputs=int(hp["NUM_OUTPUTS"]), num_inputs=int(hp["NUM_INPUTS"]),
                                use_bias=hp.get("USE_BIAS", "T").strip().upper().startswith("T"), activation=act,
                                device=device)
    elif name == "kipf":
        nvf = [int(v) for v in hp["NUM_VERTEX_FEATURES"].split()]
        T = int(hp.get("NUM_TIME_STEPS", len(nvf) - 1))
        if T != len(nvf) - 1:
            raise ValueError(f"NUM_TIME_STEPS = {T} does not match length of NUM_VERTEX_FEATURES = {len(nvf) - 1}")
        layer = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=T, activation=act, device=device)
    else:
        layer = graph_nop_layer_type(num_outputs=int(hp["NUM_OUTPUTS"]), coord_dim=int(hp["COORD_DIM"]),
                                     kernel_hidden=int(hp["KERNEL_HIDDEN"]), num_inputs=int(hp["NUM_INPUTS"]),
                                     use_bias=hp.get("USE_BIAS", "T").strip().upper().startswith("T"), activation=act,
                                     device=device)
    if weights is None:
        import warnings
        warnings.warn(f"WEIGHTS card in {name.upper()} not found")
    else:
        layer.set_params(weights)
    return layer
