"""Writes tests/golden/*.json.

Provenance of every vector (nothing here is computed by this repo's own code):

reference_kat_kipf_identity.json
    Transcribed from the reference's only numeric known-answer test on this path,
    /root/reference/test/test_diffstruc_extd_kipf.f90: identity graph adj_ia=[1,2,3],
    adj_ja=[[1,0],[2,0]] (:29-31), X=[[1,2],[3,4]] (:25-26): forward == X (:33-37), gradient of
    sum(output) == 1 (:39-45), partials of upstream [[5,6],[7,8]] / [[2,1],[4,3]] equal the
    upstreams (:62-91).

survey_recorded_path_graph.json
    Outputs of the reference's unmodified arithmetic on a 3-vertex path graph as recorded at
    survey time in SURVEY.md section 8(c) ("How" row), 6 significant digits as printed there.
    They cannot be regenerated in this round (the reference needs the un-vendored diffstruc /
    coreutils modules, fpm.toml:18-21), so they are kept as transcribed data with tolerance 5e-6
    relative (print precision).

reference_test_topologies.json
    Graph topologies and feature patterns the reference's layer tests use (inputs only; those
    tests assert shapes/properties, not numbers): test_kipf_msgpass_layer.f90:83-90 (6v/8e),
    test_msgpass_network.f90:249-276 (5v/6e with feature values), test_onnx_gnn.f90:113-131 (4v/5e).
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

kat = {
    "source": "test/test_diffstruc_extd_kipf.f90:22-91",
    "adj_ia": [1, 2, 3],
    "adj_ja": [[1, 2], [0, 0]],
    "x": [[1.0, 2.0], [3.0, 4.0]],
    "forward": [[1.0, 2.0], [3.0, 4.0]],
    "grad_of_sum": [[1.0, 1.0], [1.0, 1.0]],
    "upstream": [[5.0, 6.0], [7.0, 8.0]],
    "reverse_partial": [[5.0, 6.0], [7.0, 8.0]],
    "second_upstream": [[2.0, 1.0], [4.0, 3.0]],
    "forward_partial": [[2.0, 1.0], [4.0, 3.0]],
    "tol_abs": 1e-6,
}

survey = {
    "source": "SURVEY.md section 8(c): survey-time run of the reference's unmodified hot-path files",
    "graph": {"adj_ia": [1, 2, 4, 5], "adj_ja": [[2, 1, 3, 2], [1, 1, 2, 2]], "comment": "path 1-2-3, edge 1=(1,2), edge 2=(2,3), no self loops"},
    "x": [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]],
    "kipf_fwd": [[2.12132, 2.82843], [4.24264, 5.65685], [2.12132, 2.82843]],
    "kipf_bwd_upstream": [[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]],
    "kipf_bwd": [[2.0, 20.0], [4.0, 40.0], [2.0, 20.0]],
    "edge_features": [[0.5], [0.25]],
    "duvenaud_propagate": [[3.0, 4.0, 0.5], [6.0, 8.0, 0.75], [3.0, 4.0, 0.25]],
    "duvenaud_update": {
        "weight": [round(0.1 * k, 1) for k in range(1, 13)],
        "min_degree": 1, "max_degree": 2, "num_outputs": 2,
        "out": [[1.75, 2.5], [6.1125, 6.85], [1.625, 2.35]],
    },
    "tol_rel": 5e-6,
}

topo = {
    "kipf_layer_6v8e": {"source": "test/test_kipf_msgpass_layer.f90:83-90", "num_vertices": 6,
                        "index_list": [[1, 1, 2, 2, 3, 4, 4, 5], [2, 3, 3, 4, 5, 5, 6, 6]]},
    "network_5v6e": {"source": "test/test_msgpass_network.f90:249-276", "num_vertices": 5,
                     "index_list": [[1, 1, 2, 2, 3, 4], [2, 3, 3, 4, 5, 5]],
                     "vertex_features_rows": [[1.0, 2.0, 3.0, 4.0, 5.0], [0.1, 0.2, 0.3, 0.4, 0.5],
                                              [1.1, 1.2, 1.3, 1.4, 1.5], [0.5, 0.4, 0.3, 0.2, 0.1],
                                              [2.0, 1.8, 1.6, 1.4, 1.2], [0.0, 0.1, 0.2, 0.3, 0.4],
                                              [1.0, 0.9, 0.8, 0.7, 0.6], [0.2, 0.4, 0.6, 0.8, 1.0]],
                     "edge_features_rows": [[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], [1.1, 1.2, 1.3, 1.4, 1.5, 1.6]]},
    "onnx_gnn_4v5e": {"source": "test/test_onnx_gnn.f90:113-131", "num_vertices": 4,
                      "index_list": [[1, 2, 3, 4, 1], [2, 3, 4, 1, 3]],
                      "vertex_features_cols": [[1.0, 2.0], [3.0, 4.0], [0.5, 1.5], [2.5, 3.5]]},
}

for name, obj in (("reference_kat_kipf_identity.json", kat), ("survey_recorded_path_graph.json", survey),
                  ("reference_test_topologies.json", topo)):
    with open(os.path.join(HERE, name), "w") as fh:
        json.dump(obj, fh, indent=1)
print("wrote golden fixtures")
