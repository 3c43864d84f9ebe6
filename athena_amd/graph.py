"""Host-side mirror of graphstruc's `graph_type` as athena's msgpass layers see it, plus the
device graph handle.

The reference reads only `adj_ia`, `adj_ja`, vertex/edge feature arrays and the counts from
`graph_type` (call sites athena_kipf_msgpass_layer.f90:943-946,
athena_msgpass_layer_sub.f90:144-174).  Conventions are kept Fortran-like so tests read like the
reference's:
  adj_ia : int32 [num_vertices+1], 1-based row pointers
  adj_ja : int32 [2, nnz]; adj_ja[0, w] = neighbour (1-based), adj_ja[1, w] = undirected edge id
           (1-based; 0 for a self-loop entry, as in test/test_diffstruc_extd_kipf.f90:29-31)
  vertex_features : float32 [num_vertex_features, num_vertices] (Fortran shape) -- stored here
           transposed as C-contiguous [num_vertices, F], which is the same memory.
"""
import ctypes as C

import numpy as np

from . import _capi


class graph_type:
    """Undirected sparse graph in athena's CSR convention."""

    def __init__(self):
        self._version = 0
        self.num_vertices = 0
        self.num_edges = 0
        self.num_vertex_features = 0
        self.num_edge_features = 0
        self.adj_ia = np.ones(1, np.int32)
        self.adj_ja = np.zeros((2, 0), np.int32, order="F")
        self.vertex_features = None  # [num_vertices, Fv]
        self.edge_features = None    # [num_edges, Fe]
        self.is_sparse = True

    # adjacency arrays: assigning either one (generate_adjacency, add_self_loops, from_csr, user code) bumps a version
    # counter that the layers' device-handle cache keys on
    @property
    def adj_ia(self):
        return self._adj_ia

    @adj_ia.setter
    def adj_ia(self, a):
        self._adj_ia = a
        self._version += 1

    @property
    def adj_ja(self):
        return self._adj_ja

    @adj_ja.setter
    def adj_ja(self, a):
        self._adj_ja = a
        self._version += 1

    def topology_key(self):
        """what a cached device handle of this graph is valid for: the object, its version (bumped by every assignment
        of adj_ia / adj_ja), the sizes and the C ABI's content key of the arrays (athena_mp_graph_key: EVERY word of
        both arrays, hashed by a few host threads -- an in-place edit is always seen, as in the reference, which
        re-copies the CSR on every set_graph.  Only under ATHENA_MP_GRAPH_KEY_SAMPLED=1 does a graph of 2^18 entries
        or more get the cheap sampled key; then an in-place edit needs touch(), which also evicts the stale handle
        from the library's cache)."""
        import ctypes as C

        from . import _capi

        # A FROZEN graph (freeze(): both arrays read-only, so an in-place edit raises instead of going unseen) keeps its key
        # for as long as the arrays and the version stay what they were: set_graph before every forward then costs a tuple
        # compare, not a pass over every word of the CSR (1.5 ms per 84 MB, per layer, per step).
        memo = getattr(self, "_key_memo", None)
        if memo is not None:
            ia0, ja0 = self._adj_ia, self._adj_ja
            if (memo[0] == (self._version, id(ia0), id(ja0), ia0.ctypes.data, ja0.ctypes.data, ia0.size, ja0.size)
                    and not ia0.flags.writeable and not ja0.flags.writeable):
                return memo[1]
        ia = np.ascontiguousarray(self._adj_ia, dtype=np.int32)
        ja = np.asfortranarray(self._adj_ja, dtype=np.int32)
        key = C.c_uint64(0)
        _capi.call("athena_mp_graph_key", int(ia.size - 1), int(ja.shape[1]), ia.ctypes.data, ja.ctypes.data, C.byref(key))
        out = (id(self), self._version, int(self.num_vertices), int(ja.shape[1]), int(key.value))
        # memoised ONLY for the arrays freeze() made: views of immutable bytes objects, which numpy refuses to make writeable
        # again -- a read-only array that owns its data could be un-frozen, edited and re-frozen behind the memo
        ia0, ja0 = self._adj_ia, self._adj_ja
        if getattr(self, "_frozen", None) == (id(ia0), id(ja0)) and not ia0.flags.writeable and not ja0.flags.writeable:
            self._key_memo = ((self._version, id(ia0), id(ja0), ia0.ctypes.data, ja0.ctypes.data, ia0.size, ja0.size), out)
        else:
            self._key_memo = None
        return out

    def freeze(self):
        """Make adj_ia / adj_ja IMMUTABLE arrays (int32, contiguous / Fortran order, views of bytes objects: an in-place edit
        raises, and so does setting flags.writeable back to True), so topology_key() may memoise the content key -- nothing can
        change under it without an assignment (which bumps the version and drops the memo).  Returns self."""
        ia_src = np.ascontiguousarray(self._adj_ia, dtype=np.int32)
        ja_src = np.asfortranarray(self._adj_ja, dtype=np.int32)
        ia = np.frombuffer(ia_src.tobytes(order="C"), dtype=np.int32)
        ja = np.frombuffer(ja_src.tobytes(order="F"), dtype=np.int32).reshape(ja_src.shape, order="F")
        self._adj_ia, self._adj_ja = ia, ja
        self._version += 1
        self._key_memo = None
        self._frozen = (id(ia), id(ja))
        return self

    def touch(self):
        """the adjacency arrays were edited in place: cached device handles of this graph are stale"""
        self._version += 1

    # -- graphstruc API used by the reference's tests (test_kipf_msgpass_layer.f90:71-100) -------
    def set_num_vertices(self, n, num_features=0):
        self.num_vertices = int(n)
        self.num_vertex_features = int(num_features)
        self.vertex_features = np.zeros((self.num_vertices, self.num_vertex_features), np.float32)

    def set_num_edges(self, n, num_features=0):
        self.num_edges = int(n)
        self.num_edge_features = int(num_features)
        self.edge_features = np.zeros((self.num_edges, self.num_edge_features), np.float32)

    def generate_adjacency(self, index_list):
        """index_list [2, num_edges], 1-based vertex pairs; edge e = column e (1-based id).

        Each undirected edge contributes two CSR entries sharing the same edge id.  Within a row,
        entries are ordered by edge id (graphstruc's internal order is not visible from the
        reference and only affects fp32 summation order)."""
        idx = np.asarray(index_list, dtype=np.int64)
        if not (idx.ndim == 2 and idx.shape[0] == 2):
            raise ValueError('expected: idx.ndim == 2 and idx.shape[0] == 2')
        E = idx.shape[1]
        if self.num_edges == 0:
            self.num_edges = E
        src = np.concatenate([idx[0], idx[1]])
        dst = np.concatenate([idx[1], idx[0]])
        eid = np.concatenate([np.arange(1, E + 1), np.arange(1, E + 1)])
        loops = idx[0] == idx[1]
        if loops.any():  # a self edge is a single CSR entry
            keep = np.concatenate([np.ones(E, bool), ~loops])
            src, dst, eid = src[keep], dst[keep], eid[keep]
        self._from_entries(src, dst, eid)

    def generate_adjacency_device(self, index_list, add_self_loops=False):
        """generate_adjacency (+ add_self_loops) on the GPU: same arrays as the host methods, built by one radix sort
        (athena_mp_csr_from_edges); for edge lists of millions of pairs"""
        idx = np.asfortranarray(np.asarray(index_list, dtype=np.int32))
        if not (idx.ndim == 2 and idx.shape[0] == 2):
            raise ValueError('expected: idx.ndim == 2 and idx.shape[0] == 2')
        E = idx.shape[1]
        if self.num_edges == 0:
            self.num_edges = E
        _capi.init(0)
        ia = np.empty(self.num_vertices + 1, np.int32)
        nnz = C.c_int64()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        _capi.call("athena_mp_csr_from_edges", self.num_vertices, E, vp(idx), int(add_self_loops), vp(ia), None, 0, C.byref(nnz))
        ja = np.empty((2, nnz.value), np.int32, order="F")
        _capi.call("athena_mp_csr_from_edges", self.num_vertices, E, vp(idx), int(add_self_loops), vp(ia), vp(ja), nnz.value,
                   C.byref(nnz))
        self.adj_ia, self.adj_ja = ia, ja

    def add_self_loops(self):
        """A~ = A + I: one entry (v, v) with edge id 0 per vertex that has none."""
        rows = np.repeat(np.arange(1, self.num_vertices + 1), np.diff(self.adj_ia))
        has = np.zeros(self.num_vertices + 1, bool)
        has[rows[self.adj_ja[0] == rows]] = True
        add = np.nonzero(~has[1:])[0] + 1
        src = np.concatenate([rows, add])
        dst = np.concatenate([self.adj_ja[0], add])
        eid = np.concatenate([self.adj_ja[1], np.zeros(add.size, np.int64)])
        self._from_entries(src, dst, eid, self_loops_first=True)

    def _from_entries(self, src, dst, eid, self_loops_first=False):
        n = self.num_vertices
        key = eid.astype(np.int64)
        if self_loops_first:
            key = np.where(eid == 0, -1, key)
        order = np.lexsort((key, src))
        src, dst, eid = src[order], dst[order], eid[order]
        counts = np.bincount(src - 1, minlength=n)
        self.adj_ia = np.concatenate([[1], 1 + np.cumsum(counts)]).astype(np.int32)
        ja = np.empty((2, src.size), np.int32, order="F")
        ja[0] = dst
        ja[1] = eid
        self.adj_ja = ja

    @property
    def nnz(self):
        return int(self.adj_ja.shape[1])

    @classmethod
    def from_csr(cls, adj_ia, adj_ja, num_edges=None):
        g = cls()
        g.adj_ia = np.ascontiguousarray(adj_ia, np.int32)
        g.adj_ja = np.asfortranarray(adj_ja, np.int32)
        g.num_vertices = g.adj_ia.size - 1
        g.num_edges = int(g.adj_ja[1].max()) if (num_edges is None and g.adj_ja.shape[1]) else int(num_edges or 0)
        return g


class DeviceGraph:
    """Owns an `athena_mp_graph*` (device CSR + transposed CSR + coefficients).

    Built once and reused: the reference re-copies the CSR on every forward
    (athena_network_sub.f90:2727-2730); the handle is what `set_graph` caches."""

    def __init__(self, adj_ia, adj_ja, n_cols=None, n_edge_cols=None, row_deg=None, col_deg=None, device=0, shared=False):
        """shared=True (square graph, degrees = row lengths): the handle comes from the library's content-keyed cache
        (athena_mp_graph_acquire) -- every layer of a network that is given the same graph gets the same device arrays,
        and close() releases a reference instead of freeing them"""
        _capi.init(device)
        ia = np.ascontiguousarray(adj_ia, np.int32)
        ja = np.asfortranarray(adj_ja, np.int32)
        if not (ja.ndim == 2 and ja.shape[0] == 2):
            raise ValueError("adj_ja must be [2, nnz]")
        self.n_rows = ia.size - 1
        self.n_cols = self.n_rows if n_cols is None else int(n_cols)
        self.nnz = int(ja.shape[1])
        if n_edge_cols is None:
            n_edge_cols = int(ja[1].max()) if self.nnz else 0
        elif n_edge_cols == 0 and self.nnz and ja[1].any():
            ja = ja.copy(order="F")  # edge ids not needed (Kipf): drop them, skip the edge index
            ja[1] = 0
        self.n_edge_cols = int(n_edge_cols)
        rd = cd = None
        if row_deg is not None:
            rd = np.ascontiguousarray(row_deg, np.int32)
            cd = np.ascontiguousarray(col_deg, np.int32)
            if not (rd.size == self.n_rows and cd.size == self.n_cols):
                raise ValueError('expected: rd.size == self.n_rows and cd.size == self.n_cols')
        h = C.c_void_p()
        if shared and rd is None and self.n_cols == self.n_rows:
            _capi.call("athena_mp_graph_acquire", self.n_rows, self.nnz, ia.ctypes.data_as(C.c_void_p),
                       ja.ctypes.data_as(C.c_void_p), self.n_edge_cols, C.byref(h))
        else:
            _capi.call(
                "athena_mp_graph_create", self.n_rows, self.n_cols, self.nnz,
                ia.ctypes.data_as(C.c_void_p), ja.ctypes.data_as(C.c_void_p), self.n_edge_cols,
                rd.ctypes.data_as(C.c_void_p) if rd is not None else None,
                cd.ctypes.data_as(C.c_void_p) if cd is not None else None, C.byref(h))
        self.handle = h

    _ARRAYS = {"rowptr": (0, np.int32), "col": (1, np.int32), "eid": (2, np.int32), "coef": (3, np.float32),
               "t_rowptr": (4, np.int32), "t_src": (5, np.int32), "t_eid": (6, np.int32), "t_coef": (7, np.float32),
               "e_rowptr": (8, np.int32), "e_row": (9, np.int32), "e_entry": (10, np.int32),
               "deg_row": (11, np.int32), "deg_col": (12, np.int32)}

    def export(self, name):
        """one array of the device handle as numpy (0-based indices)"""
        which, dt = self._ARRAYS[name]
        n = C.c_int64()
        _capi.call("athena_mp_graph_export", self.handle, which, None, 0, C.byref(n))
        out = np.empty(n.value, dt)
        _capi.call("athena_mp_graph_export", self.handle, which, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        return out

    @classmethod
    def from_edges(cls, num_vertices, index_list, add_self_loops=False, with_edge_ids=True, want_adjacency=False, device=0):
        """edge list -> device handle in one call (athena_mp_graph_create_from_edges): generate_adjacency
        [+ add_self_loops] and set_graph with the CSR entries staying in HBM in between.  Returns the handle, or
        (handle, adj_ia, adj_ja) with want_adjacency (adj_ia is produced either way)."""
        _capi.init(device)
        idx = np.asfortranarray(np.asarray(index_list, dtype=np.int32))
        if not (idx.ndim == 2 and idx.shape[0] == 2):
            raise ValueError('expected: idx.ndim == 2 and idx.shape[0] == 2')
        n, E = int(num_vertices), idx.shape[1]
        self = cls.__new__(cls)
        ia = np.empty(n + 1, np.int32)
        nnz = C.c_int64()
        h = C.c_void_p()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        ja = None
        if want_adjacency:
            cap = 2 * E + n
            ja = np.empty((2, cap), np.int32, order="F")
        _capi.call("athena_mp_graph_create_from_edges", n, E, vp(idx), int(bool(add_self_loops)), int(bool(with_edge_ids)),
                   vp(ia), vp(ja) if ja is not None else None, ja.shape[1] if ja is not None else 0, C.byref(nnz), C.byref(h))
        self.handle = h
        self.n_rows = self.n_cols = n
        self.nnz = int(nnz.value)
        self.n_edge_cols = E if with_edge_ids else 0
        if want_adjacency:
            return self, ia, np.asfortranarray(ja[:, :self.nnz])
        return self

    @classmethod
    def borrow(cls, handle, owner=None):
        """wrap a handle somebody else owns (the row blocks of an athena_mp_shard): never destroyed from here"""
        self = cls.__new__(cls)
        self.handle, self._borrowed, self._owner = handle, True, owner
        nr, nc, nnz, ne = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
        _capi.call("athena_mp_graph_dims", handle, C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(ne))
        self.n_rows, self.n_cols, self.nnz, self.n_edge_cols = nr.value, nc.value, nnz.value, ne.value
        return self

    @classmethod
    def from_graph(cls, g, device=0):
        return cls(g.adj_ia, g.adj_ja, n_edge_cols=max(g.num_edges, int(g.adj_ja[1].max()) if g.nnz else 0),
                   device=device)

    def evict(self):
        """close(), and the library's handle cache forgets this handle (athena_mp_graph_evict): the next acquire of the
        same key builds from the arrays.  What a layer does with its old handle after graph_type.touch()."""
        if getattr(self, "handle", None):
            if not getattr(self, "_borrowed", False):
                _capi.call("athena_mp_graph_evict", self.handle)
            self.handle = None

    def close(self):
        if getattr(self, "handle", None):
            if not getattr(self, "_borrowed", False):
                _capi.load().athena_mp_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
