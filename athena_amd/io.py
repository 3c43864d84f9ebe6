"""Graph interchange in the reference's `all_graphs.txt` text format and block-diagonal batching.

The format is what example/msgpass_chemical/src/main.f90:353-396 (`export_all_graphs`) writes and
example/msgpass_chemical/pytorch_network.py:372- reads:

    <num_graphs>
    per graph:
      <num_vertices> <num_edges> <num_csr_entries> <num_vertex_features> <num_edge_features>
      num_vertices rows of vertex features   (ES16.8E2, one row per vertex)
      num_edges    rows of edge features
      adj_ia       (num_vertices + 1 integers, 1-based, one line)
      num_csr_entries rows "adj_ja(1,j) adj_ja(2,j)"
      <label>      (ES16.8E2)

`batch_graphs` is the ingest step in front of the Duvenaud path: the graphs of a mini-batch become one
block-diagonal CSR (vertex and edge ids offset per graph) plus the per-graph vertex offsets the readout
needs -- one device graph handle and one launch per op instead of a host loop over ~130 k tiny graphs
(SURVEY.md 8a row a10).
"""
import numpy as np

from .graph import graph_type


def _es(v):
    """Fortran ES16.8E2: 16 wide, 8 decimals, 2-digit exponent, upper-case E"""
    s = f"{float(v):.8E}"
    mant, ex = s.split("E")
    return f"{mant}E{int(ex):+03d}".rjust(16)


def write_all_graphs(path, graphs, labels):
    with open(path, "w") as f:
        f.write(f"{len(graphs)}\n")
        for g, y in zip(graphs, labels):
            f.write(f"{g.num_vertices} {g.num_edges} {g.nnz} {g.num_vertex_features} {g.num_edge_features}\n")
            for row in np.asarray(g.vertex_features, np.float32).reshape(g.num_vertices, -1):
                f.write(" ".join(_es(v) for v in row) + " \n")
            for row in np.asarray(g.edge_features, np.float32).reshape(g.num_edges, -1):
                f.write(" ".join(_es(v) for v in row) + " \n")
            f.write(" ".join(str(int(v)) for v in g.adj_ia) + " \n")
            for j in range(g.nnz):
                f.write(f"{int(g.adj_ja[0, j])} {int(g.adj_ja[1, j])}\n")
            f.write(_es(y) + "\n")


def read_all_graphs(path):
    """returns (list of graph_type with features attached, float32 labels)"""
    with open(path) as f:
        tok = f.read().split()
    pos = 0

    def take(n, dtype):
        nonlocal pos
        out = np.array(tok[pos:pos + n], dtype=dtype)
        if out.size != n:
            raise ValueError("all_graphs file ended early")
        pos += n
        return out

    n_graphs = int(take(1, np.int64)[0])
    graphs, labels = [], np.empty(n_graphs, np.float32)
    for s in range(n_graphs):
        nv, ne, nnz, fv, fe = (int(v) for v in take(5, np.int64))
        g = graph_type()
        g.set_num_vertices(nv, fv)
        g.set_num_edges(ne, fe)
        g.vertex_features = take(nv * fv, np.float64).astype(np.float32).reshape(nv, fv)
        g.edge_features = take(ne * fe, np.float64).astype(np.float32).reshape(ne, fe)
        g.adj_ia = take(nv + 1, np.int64).astype(np.int32)
        g.adj_ja = np.asfortranarray(take(2 * nnz, np.int64).astype(np.int32).reshape(nnz, 2).T)
        if g.adj_ia[0] != 1 or g.adj_ia[-1] != nnz + 1:
            raise ValueError(f"graph {s + 1}: adj_ia does not span the {nnz} CSR entries")
        labels[s] = np.float32(take(1, np.float64)[0])
        graphs.append(g)
    if pos != len(tok):
        raise ValueError("trailing data in all_graphs file")
    return graphs, labels


def batch_graphs(graphs):
    """block-diagonal batch: returns (adj_ia, adj_ja, vertex_offsets[S+1] (0-based), vertex_features, edge_features)"""
    S = len(graphs)
    voff = np.zeros(S + 1, np.int64); eoff = np.zeros(S + 1, np.int64); woff = np.zeros(S + 1, np.int64)
    for s, g in enumerate(graphs):
        voff[s + 1] = voff[s] + g.num_vertices
        eoff[s + 1] = eoff[s] + g.num_edges
        woff[s + 1] = woff[s] + g.nnz
    ia = np.empty(voff[-1] + 1, np.int32)
    ja = np.empty((2, woff[-1]), np.int32, order="F")
    ia[0] = 1
    for s, g in enumerate(graphs):
        ia[voff[s] + 1: voff[s + 1] + 1] = g.adj_ia[1:] + woff[s]
        w = slice(woff[s], woff[s + 1])
        ja[0, w] = g.adj_ja[0] + voff[s]
        ja[1, w] = np.where(g.adj_ja[1] > 0, g.adj_ja[1] + eoff[s], 0)     # id 0 (self loop) stays 0
    x = np.concatenate([np.asarray(g.vertex_features, np.float32).reshape(g.num_vertices, -1) for g in graphs]) if S else None
    e = np.concatenate([np.asarray(g.edge_features, np.float32).reshape(g.num_edges, -1) for g in graphs]) if S else None
    return ia, ja, voff.astype(np.int32), x, e


# ---- layer cards of athena's network file ("print" / "read" of the three message-passing layers) ------------
# print_base (athena_base_layer_sub_io.f90:14-65) writes   <NAME> / body / END <NAME>;  the bodies are
# print_to_unit_kipf (athena_kipf_msgpass_layer.f90:398-436), print_to_unit_duvenaud
# (athena_duvenaud_msgpass_layer.f90:~640-700) and print_to_unit_gno (athena_graph_nop_layer.f90); activations
# print an ACTIVATION block of "name = value" attributes (athena_misc_types_sub.f90:43-75).  Weights go out as
# '(5(E16.8E2))' records, one WRITE per parameter tensor, and are read back list-directed.

def _e16(v):
    """Fortran E16.8E2: 0.dddddddd mantissa, two-digit exponent, right-justified in 16 columns"""
    v = float(v)
    if v == 0.0 or v != v or v in (float("inf"), float("-inf")):
        body = "0.00000000E+00" if v == 0.0 else ("NaN" if v != v else ("Infinity" if v > 0 else "-Infinity"))
        return body.rjust(16)
    mant, ex = f"{abs(v):.7E}".split("E")            # d.ddddddd E ee  ->  0.dddddddd E (ee+1)
    digits = mant.replace(".", "")
    return (("-" if v < 0 else "") + "0." + digits + f"E{int(ex) + 1:+03d}").rjust(16)


def _weights_block(tensors):
    lines = ["WEIGHTS"]
    for t in tensors:                                   # one WRITE statement per tensor: records of five values
        flat = np.asarray(t, np.float32).reshape(-1)
        for i in range(0, flat.size, 5):
            lines.append("".join(_e16(v) for v in flat[i:i + 5]))
    lines.append("END WEIGHTS")
    return lines


def _activation_block(actv, identifier=None):
    """print_to_unit_actv (athena_misc_types_sub.f90:43-75) over export_attributes_* : name, scale, then the
    activation's own attributes, floats as adjustl('(F10.6)')"""
    from . import ops
    a = actv if isinstance(actv, ops.actv_type) else ops.actv_type(actv)
    out = ["   ACTIVATION" + (f": {identifier}" if identifier else "")]
    out.append(f"      name = {a.name}")
    out.append(f"      scale = {a.scale:.6f}")
    if a.name == "swish":
        out.append(f"      beta = {a.beta:.6f}")
    for k, spec in enumerate(ops.ACTP.get(a.name, (0, None, None))[1:]):
        if spec is not None:
            out.append(f"      {spec[0].rstrip('_')} = {a.p[k]:.6f}")
    out.append("   END ACTIVATION")
    return out


def activation_from_card(hp, prefix="activation_"):
    """the ACTIVATION block of a parsed card -> a name or an ops.actv_type (read_activation + apply_attributes_*)"""
    from . import ops
    name = hp.get(prefix + "name", "none")
    attrs = {k[len(prefix):]: float(v) for k, v in hp.items() if k.startswith(prefix) and k != prefix + "name"}
    return ops.resolve_activation(ops.actv_type(name, **attrs))


def layer_card(layer):
    """the text card of a layer mirror (kipf / duvenaud / graph_nop), as print_base would write it"""
    name = layer.name.upper()
    lines = [name]
    params = [p.detach().cpu().numpy() for p in layer.params]
    if layer.name == "kipf":
        lines.append(f"   NUM_TIME_STEPS = {layer.num_time_steps}")
        lines.append("   NUM_VERTEX_FEATURES =" + "".join(f" {v}" for v in layer.num_vertex_features))
        if layer.activation != "none":
            lines += _activation_block(layer.activation)
    elif layer.name == "duvenaud":
        lines.append(f"   NUM_TIME_STEPS = {layer.num_time_steps}")
        lines.append("   NUM_VERTEX_FEATURES =" + "".join(f" {v}" for v in layer.num_vertex_features))
        lines.append("   NUM_EDGE_FEATURES =" + "".join(f" {v}" for v in layer.num_edge_features))
        if layer.activation != "none":
            lines += _activation_block(layer.activation, identifier="MESSAGE")
        lines += _activation_block(layer.activation_readout, identifier="READOUT")
    elif layer.name == "full":             # print_to_unit_full, athena_full_layer.f90:434-468
        lines.append(f"   NUM_INPUTS = {layer.num_inputs}")
        lines.append(f"   NUM_OUTPUTS = {layer.num_outputs}")
        lines.append(f"   USE_BIAS = {'T' if layer.use_bias else 'F'}")
        if layer.activation != "none":
            lines += _activation_block(layer.activation)
    elif layer.name == "graph_nop":
        lines.append(f"   NUM_INPUTS = {layer.num_vertex_features[0]}")
        lines.append(f"   NUM_OUTPUTS = {layer.num_outputs}")
        lines.append(f"   COORD_DIM = {layer.coord_dim}")
        lines.append(f"   KERNEL_HIDDEN = {layer.kernel_hidden}")
        lines.append(f"   USE_BIAS = {'T' if layer.use_bias else 'F'}")
        if layer.activation != "none":
            lines += _activation_block(layer.activation)
    else:
        raise ValueError(f"no card format for layer '{layer.name}'")
    lines += _weights_block(params)
    lines.append("END " + name)
    return "\n".join(lines) + "\n"


def parse_layer_card(text):
    """Reads one KIPF or GRAPH_NOP card (read_kipf :440-600, read_gno; the reference's read_duvenaud is an empty
    stub, :703-714, so there is nothing to mirror for it).  Returns (name, hyperparameters, flat weight vector)."""
    lines = [l.rstrip() for l in text.splitlines() if l.strip()]
    name = lines[0].strip().lower()
    if name not in ("kipf", "graph_nop", "full", "duvenaud"):
        raise ValueError(f"unsupported layer card '{lines[0].strip()}'")
    if lines[-1].strip() != "END " + name.upper():
        raise ValueError(f"END {name.upper()} not where expected")                  # the reference's message
    hp, weights, i = {}, None, 1
    while i < len(lines) - 1:
        l = lines[i].strip()
        if l.startswith("ACTIVATION"):
            j = i + 1
            while lines[j].strip() != "END ACTIVATION":
                k, v = (s.strip() for s in lines[j].split("=", 1))
                hp["activation_" + k] = v
                j += 1
            i = j + 1
        elif l == "WEIGHTS":
            j = i + 1
            vals = []
            while lines[j].strip() != "END WEIGHTS":
                vals += [float(f) for f in lines[j].split()]   # every E16.8E2 field starts with a blank: list-directed read
                j += 1
                if j >= len(lines):
                    raise ValueError("END WEIGHTS not where expected")
            weights = np.array(vals, np.float32)
            i = j + 1
        elif "=" in l:
            k, v = (s.strip() for s in l.split("=", 1))
            hp[k] = v
            i += 1
        else:
            raise ValueError(f"Unrecognised line in input file: {l}")            # read_kipf :~545
    return name, hp, weights
